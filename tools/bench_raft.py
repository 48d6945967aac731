#!/usr/bin/env python3
"""Tuning tool (GPU box): the frozen RAFT plan alone at the 1080p geometry (360 x 640 inputs -> 45 x 80 maps), replayed from a
hipGraph: total per call, the 12 refinement iterations, and one iteration's launches."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ops_mod = importlib.import_module("zero-tig_amd.ops")
lib_mod = importlib.import_module("zero-tig_amd.lib")
raft_mod = importlib.import_module("zero-tig_amd.raft")
synth = importlib.import_module("zero-tig_amd.synth")
ops = ops_mod.Ops(lib_mod.get_lib())
dev = torch.device("cuda:0")
prec = os.environ.get("ZT_PREC", "bf16")
st = synth.make_state(1)
W = {k: torch.from_numpy(np.array(v)).to(dev) for k, v in st.items() if k.startswith("raft.")}
plan = raft_mod.RaftPlan(ops, W, dev, precision=prec)
h, w = 360, 640
x2 = (torch.randn(2, h, w, 8 if prec == "bf16" else 4, device=dev) * 0.5).to(plan.adt)


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


t12 = timed(graphed(lambda: plan.run(x2, iters=12)))
t0 = timed(graphed(lambda: plan.run(x2, iters=0)))
n0 = sum(ops.lib.calls.values())
plan.run(x2, iters=1)
n1 = sum(ops.lib.calls.values())
plan.run(x2, iters=2)
n2 = sum(ops.lib.calls.values())
print("RAFT %s 360x640: %.1f us per call; encoders + corr volume + mask head + upsample %.1f us; refinement %.1f us per iteration "
      "(%d launches per iteration)" % (prec, t12, t0, (t12 - t0) / 12.0, (n2 - n1) - (n1 - n0) + (n1 - n0) - (n1 - n0) if False else (n2 - n1) - (n1 - n0)))
