#!/usr/bin/env python3
"""Tuning tool (GPU box): time the bf16 convolution variants on the 1080p Enhancer layer, with ablations."""
import importlib
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops_mod = importlib.import_module("zero-tig_amd.ops")
lib_mod = importlib.import_module("zero-tig_amd.lib")
ops = ops_mod.Ops(lib_mod.get_lib())
dev = torch.device("cuda:0")
H, W = 1080, 1920


def run(cin, cout, k, variant, epi=0, iters=10):
    x = (torch.randn(1, H, W, (cin + 7) // 8 * 8, device=dev) * 0.5).bfloat16()
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wd = ops.repack_weight_bf16(w)
    aux = torch.randn(1, H, W, (cout + 7) // 8 * 8, device=dev).bfloat16() if epi else None
    out = torch.empty(1, H, W, (cout + 7) // 8 * 8, device=dev, dtype=torch.bfloat16)
    for _ in range(2):
        ops.conv2d_bf16(ops_mod.CV(x, 0, cin), wd, None, cout, k, k, (k // 2, k // 2), "relu", out=out, aux=aux, epi=epi, variant=variant)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d_bf16(ops_mod.CV(x, 0, cin), wd, None, cout, k, k, (k // 2, k // 2), "relu", out=out, aux=aux, epi=epi, variant=variant)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


AB_MODE = os.environ.get("ZT_BENCH_AB")
names = {3: "register-stationary", 33: "rs no-mfma", 34: "rs no-store", 36: "rs no-halo", 38: "rs mfma only", 39: "rs barriers only", 1: "ws full", 2: "tiled"}
ONLY = os.environ.get("ZT_BENCH_ONLY")
if ONLY:          # one variant of one layer, for counter collection: ZT_BENCH_ONLY="64,64,3[,epi]"
    t = [int(v) for v in ONLY.split(",")]
    print("c%d->%d variant %d epi %d: %7.1f us" % (t[0], t[1], t[2], t[3] if len(t) > 3 else 0, run(t[0], t[1], 3, t[2], epi=t[3] if len(t) > 3 else 0, iters=30)), flush=True)
    sys.exit(0)
ABL = os.environ.get("ZT_BENCH_ABL")
if ABL:           # phase ablations of the register-stationary kernel on one layer: ZT_BENCH_ABL="9,64"
    cin, cout = (int(t) for t in ABL.split(","))
    for v in (3, 33, 34, 36, 38, 39):
        print("c%d->%d  %-22s %7.1f us" % (cin, cout, names[v], run(cin, cout, 3, v, iters=30)), flush=True)
    sys.exit(0)
if AB_MODE:      # A/B of two variants, interleaved, many iterations: ZT_BENCH_AB="3,40"
    va, vb = (int(t) for t in AB_MODE.split(","))
    for (cin, cout) in ((64, 64), (48, 48)):
        for rep in range(3):
            print("c%d->%d  variant %d: %7.1f us   variant %d: %7.1f us" % (cin, cout, va, run(cin, cout, 3, va, iters=60), vb, run(cin, cout, 3, vb, iters=60)), flush=True)
    sys.exit(0)
for (cin, cout, k) in ((64, 64, 3), (48, 48, 3)):
    for v in (3, 33, 34, 36, 38, 39, 1):
        print("c%d->%d k%d  %-24s %8.1f us" % (cin, cout, k, names[v], run(cin, cout, k, v)), flush=True)
print("c64->64 k3 ws full + residual epi %8.1f us" % run(64, 64, 3, 1, epi=3))
print("c64->64 k3 rs + residual epi      %8.1f us" % run(64, 64, 3, 3, epi=3))
print("c48->48 k3 rs + mask epi          %8.1f us" % run(48, 48, 3, 3, epi=1))
for (cin, cout) in ((9, 64), (3, 48), (12, 48)):
    for v in (3, 1):
        print("c%d->%d k3  %-24s %8.1f us" % (cin, cout, names[v], run(cin, cout, 3, v)), flush=True)
