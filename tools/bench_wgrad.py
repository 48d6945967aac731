#!/usr/bin/env python3
"""Tuning tool (GPU box): the bf16 weight-gradient kernels on the 1080p layers (env knobs: ZT_WGRAD_NW4, ZT_WGRAD_BLOCKS)."""
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
H, W = 1080, 1920
for (cin, cout, k, hh, ww) in ((64, 64, 3, H, W), (48, 48, 3, H, W), (48, 48, 3, H // 2, W // 2), (12, 48, 3, H, W), (9, 64, 3, H, W), (64, 3, 3, H, W)):
    x = (torch.randn(1, hh, ww, (cin + 7) // 8 * 8, device=dev) * 0.5).bfloat16()
    dz = (torch.randn(1, hh, ww, (cout + 7) // 8 * 8, device=dev) * 0.5).bfloat16()
    gw = torch.zeros(cout, cin, k, k, device=dev)
    gb = torch.zeros(cout, device=dev)
    f = lambda: ops.conv2d_wgrad_bf16(CV(x, 0, cin), CV(dz, 0, cout), cout, k, k, gw, grad_b=gb)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    alg = hh * ww * ((cin + 7) // 8 * 8 + (cout + 7) // 8 * 8) * 2
    print("wgrad %2d->%2d k%d %4dx%4d: %7.1f us (incl. slab reduce)  %.2f TB/s algorithmic  %.0f TFLOP/s" %
          (cin, cout, k, hh, ww, us, alg / us / 1e6, 2.0 * k * k * cin * cout * hh * ww / us / 1e6), flush=True)
