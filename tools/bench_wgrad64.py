#!/usr/bin/env python3
"""GPU box: A/B of the 64 -> 64 3x3 weight gradient at 1080p in ONE process, interleaved rounds (cdna_hip_programming.md rule 24):
ZT_WGRAD_DMA=1 (LDS-DMA double-buffered, swizzled image: wgrad64_dma_bf16_kernel) vs 0 (register-staged wgrad_mfma_bf16_kernel<3,3,4,4,8>).
Times the partial kernel alone (slabs appended, no reduce).  Run under `rocprofv3 --kernel-trace --stats` for per-kernel averages."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops_mod = importlib.import_module("zero-tig_amd.ops")
lib_mod = importlib.import_module("zero-tig_amd.lib")
ops = ops_mod.Ops(lib_mod.get_lib())
CV = ops_mod.CV
dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
CH = int(os.environ.get("ZT_BENCH_CH", "64"))          # 64 (Enhancer conv.0) or 48 (Denoise_1/2 conv2)
x = (torch.randn(1, H, W, CH, device=dev) * 0.5).bfloat16()
dz = (torch.randn(1, H, W, CH, device=dev) * 0.5).bfloat16()
per = ops.wgrad_slab_floats(CH, CH, 3)
slab = torch.empty(600 * per, dtype=torch.float32, device=dev)
res = {"1": [], "0": []}
for rnd in range(12):
    for v in ("1", "0"):
        os.environ["ZT_WGRAD_DMA"] = v
        ops.wgrad_partial_bf16(CV(x), CV(dz), CH, 3, slab, 0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.wgrad_partial_bf16(CV(x), CV(dz), CH, 3, slab, 0)
        e1.record()
        torch.cuda.synchronize()
        if rnd >= 2:
            res[v].append(e0.elapsed_time(e1) / 10 * 1e3)
alg = H * W * 2 * CH * 2
for v, name in (("1", "LDS-DMA double-buffered"), ("0", "register-staged")):
    t = sorted(res[v])
    med = t[len(t) // 2]
    print("wgrad %d->%d 3x3 %dx%d %-24s: median %.1f us (min %.1f) = %.2f TB/s algorithmic (%.0f %% of 8 TB/s), %.0f TFLOP/s" %
          (CH, CH, H, W, name, med, t[0], alg / med / 1e6, 100 * alg / med / 1e6 / 8.0, 2.0 * 9 * CH * CH * H * W / med / 1e6), flush=True)
