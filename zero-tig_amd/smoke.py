"""One small invocation of the hot path on cuda:0, checked against the CPU oracle (driver's smoke test)."""
import argparse
import importlib
import os
import sys

import numpy as np
import torch


def run(H=128, W=160, seed=1):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import zt_oracle                       # checker only
    synth = importlib.import_module("zero-tig_amd.synth")
    net_mod = importlib.import_module("zero-tig_amd.network")
    optim = importlib.import_module("zero-tig_amd.optim")
    dev = torch.device("cuda:0")
    net = net_mod.Network(argparse.Namespace(dataset="RLV", of_scale=1))
    st = synth.make_state(seed)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    net = net.to(dev).train()
    opt = optim.ClipAdam(net)
    tr = zt_oracle.OracleTrainer(zt_oracle.to_torch_state(synth.make_state(seed)), of_scale=1)
    for t in range(2):                                 # frame 1 exercises equalize + RAFT + warp
        x = torch.from_numpy(synth.lowlight_frame(t, H, W))
        net.is_new_seq = (t == 0)
        opt.zero_grad()
        loss = net._loss(x.to(dev))
        loss.backward()
        opt.step()
        ref, _, outs, _, _, _ = tr.step(x, t == 0)
        rel = abs(float(loss.detach()) - float(ref)) / abs(float(ref))
        err = float((net.last_H3.cpu() - outs[13].detach()).abs().max())
        print("smoke frame %d: loss %.5f (oracle %.5f, rel %.2e), max|H3 - oracle| %.2e" % (t, float(loss), float(ref), rel, err))
        assert rel < 1e-3 and err < 5e-3, (rel, err)      # frame 1 follows an Adam step (+-lr sign flips, see tests)
    print("smoke ok; native library:", net._ops.lib.path)
