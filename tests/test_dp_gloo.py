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


def _worker_trainstep(rank, world, port, emu_so, out_dir):
    """What train.py does per rank (train.py:119-138 of this repo): TrainStep (eager) over the rank's own clip, the epoch-end
    resume file with the BatchNorm merge, then -- with --reference_eval_quirk -- one more step with every rank in eval() mode."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib_mod = importlib.import_module("zero-tig_amd.lib")
    ops_mod = importlib.import_module("zero-tig_amd.ops")
    optim = importlib.import_module("zero-tig_amd.optim")
    utils = importlib.import_module("utils.utils")
    net, synth = _build(ops_mod.Ops(lib_mod.Lib(emu_so)))
    opt = optim.ClipAdam(net)
    stepper = optim.TrainStep(net, opt, use_graph=False)
    Hs, Ws = H, W                                          # new-sequence frames only: RAFT on the emulator costs minutes per step
    losses = []                                            # (the steady-state path under DP is the same optimizer code; GPU tests cover RAFT)
    for t in range(3):
        x = torch.from_numpy(synth.lowlight_frame(t, Hs, Ws, seed=2 + 1000 * rank))
        losses.append(float(stepper(x, is_new_seq=True)))
    bn = net.enhance.conv[1]
    local_bn = (bn.running_mean.clone(), bn.running_var.clone())
    utils.save_checkpoint(net, opt, os.path.join(out_dir, "resume.pt"), epoch=1, step=3)     # all ranks call it (collective inside)
    merged_bn = (bn.running_mean.clone(), bn.running_var.clone())
    net.eval()                                             # train.py: every rank switches mode together (--reference_eval_quirk keeps it)
    x = torch.from_numpy(synth.lowlight_frame(3, Hs, Ws, seed=2 + 1000 * rank))
    losses.append(float(stepper(x, is_new_seq=True)))
    torch.save({"flat": net.flat_params().flat.clone(), "m": opt.m.clone(), "v": opt.v.clone(), "t": opt.t, "losses": losses,
                "local_bn": local_bn, "merged_bn": merged_bn, "bn_after": (bn.running_mean.clone(), bn.running_var.clone()),
                "nbt": int(bn.num_batches_tracked)}, os.path.join(out_dir, "ts%d.pt" % rank))
    dist.destroy_process_group()


def test_two_rank_trainstep_checkpoint_merge_and_eval_quirk(emu_ops, tmp_path):
    """SURVEY 8(e) beyond one step: 2 ranks x (3 train-mode steps, resume file with the BN merge, 1 eval-mode
    step): parameters and Adam state stay bit-identical across ranks although every rank sees different frames, the resume file
    holds the rank-average of the BN running statistics, and the eval-mode step leaves the running statistics untouched."""
    ops, _ = emu_ops
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_trainstep, args=(2, port, ops.lib.path, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "ts0.pt"), torch.load(tmp_path / "ts1.pt")
    for k in ("flat", "m", "v"):
        assert torch.equal(r0[k], r1[k]), k
    assert r0["t"] == r1["t"] == 4 and r0["nbt"] == r1["nbt"] == 9          # 3 train-mode steps x 3 BN calls; eval step adds none
    assert r0["losses"] != r1["losses"]                                       # different clips
    assert not torch.equal(r0["local_bn"][0], r1["local_bn"][0])              # per-rank statistics before the merge (no SyncBN)
    for i in (0, 1):
        want = 0.5 * (r0["local_bn"][i] + r1["local_bn"][i])
        assert torch.equal(r0["merged_bn"][i], r1["merged_bn"][i])
        assert float((r0["merged_bn"][i] - want).abs().max()) <= 1e-7 * float(want.abs().max())
        assert torch.equal(r0["bn_after"][i], r0["merged_bn"][i])             # eval-mode BN (A-14) does not update running stats
    ck = torch.load(tmp_path / "resume.pt")
    assert ck["epoch"] == 1 and ck["step"] == 3 and ck["optimizer"]["t"] == 3 and len(ck["model"]) == 223
    assert torch.equal(ck["model"]["enhance.conv.1.running_mean"], r0["merged_bn"][0])
