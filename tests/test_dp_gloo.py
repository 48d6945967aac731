"""Data-parallel path (world_size 2, gloo, CPU + emulated kernels): one flat-bucket all-reduce per step, identical
clip+Adam on every rank.  The RCCL path on the GPUs runs the same code with backend "nccl"."""
import argparse
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

H, W = 32, 48


def _build(ops, seed=1):
    synth = importlib.import_module("zero-tig_amd.synth")
    net_mod = importlib.import_module("zero-tig_amd.network")
    net = net_mod.Network(argparse.Namespace(dataset="RLV", of_scale=1), ops=ops)
    st = synth.make_state(seed)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    return net.train(), synth


def _worker(rank, world, port, emu_so, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib_mod = importlib.import_module("zero-tig_amd.lib")
    ops_mod = importlib.import_module("zero-tig_amd.ops")
    optim = importlib.import_module("zero-tig_amd.optim")
    net, synth = _build(ops_mod.Ops(lib_mod.Lib(emu_so)))
    opt = optim.ClipAdam(net)
    assert opt.world() == world
    x = torch.from_numpy(synth.lowlight_frame(0, H, W, seed=2 + 1000 * rank))      # each rank owns its own clip
    net.is_new_seq = True
    opt.zero_grad()
    loss = net._loss(x)
    loss.backward()
    local_grad = net.flat_params().grad.clone()
    gn = opt.step()
    torch.save({"flat": net.flat_params().flat.clone(), "grad": local_grad, "gnorm": gn.clone(), "loss": loss.detach()},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_flat_bucket_allreduce(emu_ops, tmp_path):
    ops, _ = emu_ops
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ops.lib.path, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["flat"], r1["flat"]), "ranks must hold bit-identical parameters after the step"
    assert float(r0["gnorm"]) == float(r1["gnorm"]) and float(r0["loss"]) != float(r1["loss"])
    # single-process replay: average the two local gradients, same fused clip+Adam kernel
    optim = importlib.import_module("zero-tig_amd.optim")
    net, _ = _build(ops)
    opt = optim.ClipAdam(net)
    opt.zero_grad()
    net.flat_params().grad.copy_(0.5 * (r0["grad"] + r1["grad"]))
    opt.step()
    assert float((net.flat_params().flat - r0["flat"]).abs().max()) < 1e-7
    assert float((r0["grad"] - r1["grad"]).abs().max()) > 0
