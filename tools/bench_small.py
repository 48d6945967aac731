#!/usr/bin/env python3
"""Tuning tool (GPU box): where do the 11-16 us of a small-map (45 x 80) RAFT convolution go?  Times one layer back-to-back (warm
instruction cache, weights in L2) with phase ablations of the tiled kernel (variant 64 + bits: 2 no epilogue, 4 no K loop, 8 empty kernel,
16 K range walked four times), and the same layers alternated (cold instruction cache, as inside the refinement loop)."""
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


class Layer:
    def __init__(self, cin, cout, kh, kw, act="relu"):
        self.cin, self.cout, self.kh, self.kw, self.act = cin, cout, kh, kw, act
        self.x = (torch.randn(1, H, W, (cin + 7) // 8 * 8, device=dev) * 0.5).bfloat16()
        self.wd = ops.repack_weight_bf16(torch.randn(cout, cin, kh, kw, device=dev) * 0.05)
        self.out = torch.empty(1, H, W, (cout + 7) // 8 * 8, device=dev, dtype=torch.bfloat16)

    def launch(self, variant):
        ops.conv2d_bf16(CV(self.x, 0, self.cin), self.wd, None, self.cout, self.kh, self.kw, (self.kh // 2, self.kw // 2), self.act,
                        out=self.out, variant=variant)


def timed(fn, iters=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


layers = {"gru 1x5 384->256 sigmoid": Layer(384, 256, 1, 5, "sigmoid"), "3x3 256->192 relu": Layer(256, 192, 3, 3), "3x3 128->256 relu": Layer(128, 256, 3, 3),
          "1x1 328->256 relu": Layer(328, 256, 1, 1), "7x7 8->128 relu": Layer(2, 128, 7, 7), "3x3 256->2": Layer(256, 2, 3, 3, None)}
g = torch.cuda.CUDAGraph()
for name, L in layers.items():
    row = []
    for v, tag in ((2, "full"), (66, "no-epilogue"), (68, "no-K-loop"), (70, "prologue-only"), (72, "empty"), (80, "K x4")):
        # replay 50 launches from a hipGraph: no host launch cost in the number
        L.launch(v)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50):
                L.launch(v)
        row.append("%s %5.1f" % (tag, timed(g.replay, 20) / 50))
    print("%-26s back-to-back (graph): %s us" % (name, " | ".join(row)), flush=True)
ls = list(layers.values())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        for L in ls:
            L.launch(2)
print("alternating the %d layers (graph): %.1f us per launch (sum of back-to-back times / %d for comparison)" % (len(ls), timed(g.replay, 20) / (10 * len(ls)), len(ls)), flush=True)
