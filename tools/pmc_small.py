#!/usr/bin/env python3
"""Tuning tool (GPU box, under rocprofv3 --pmc): a few eager launches of two small-map RAFT layers (see tools/bench_small.py)."""
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
H, W = 45, 80
for (cin, cout, kh, kw) in ((384, 256, 1, 5), (256, 192, 3, 3)):
    x = (torch.randn(1, H, W, cin, device=dev) * 0.5).bfloat16()
    wd = ops.repack_weight_bf16(torch.randn(cout, cin, kh, kw, device=dev) * 0.05)
    out = torch.empty(1, H, W, cout, device=dev, dtype=torch.bfloat16)
    for _ in range(6):
        ops.conv2d_bf16(CV(x, 0, cin), wd, None, cout, kh, kw, (kh // 2, kw // 2), "relu", out=out, variant=2)
torch.cuda.synchronize()
