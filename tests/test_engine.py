"""End-to-end parity of the hand-written forward / fused loss+gradient / backward plan against the CPU oracle
and the reference-generated goldens."""
import importlib

import numpy as np
import pytest
import torch

from conftest import frames, load_golden


def _engine(ops, dev, synth, seed, is_WB=False):
    eng_mod = importlib.import_module("zero-tig_amd.engine")
    st = synth.make_state(seed)
    names = [n for n, _ in synth.enhancement_inventory() if not n.startswith("enhance.blocks")]
    params = {n: torch.from_numpy(np.array(st[n])).to(dev) for n in names if n.endswith((".weight", ".bias"))}
    bufs = {n: torch.from_numpy(np.array(st[n])).to(dev) for n in names if "running" in n or "num_batches" in n}
    return eng_mod.Engine(ops, params, bufs, is_WB=is_WB, device=dev), params, bufs


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("name,wb", [("g12_newseq_rlv_48x64", False), ("g5_newseq_wb_48x64", True)])
def test_newseq_forward_loss_grads(backend, synth, oracle, name, wb):
    ops, dev, bname = backend
    g = load_golden(name)
    H, W, seed, _ = [int(v) for v in g["meta"]]
    x = frames(synth, 1, H, W)[0].to(dev)
    eng, params, bufs = _engine(ops, dev, synth, seed, wb)
    outs = eng.forward(x)
    for i, o in enumerate(outs):
        ref = g["out%02d" % i]
        if i in (17, 18):
            assert (o.cpu().numpy() != ref).mean() <= 1e-3, i
        else:
            err = float(np.abs(o.cpu().numpy() - ref).max())
            assert err < 2e-5, (oracle.FORWARD_NAMES[i], err)
    assert float((bufs["enhance.conv.1.running_mean"].cpu() - torch.from_numpy(g["bn_running_mean"])).abs().max()) < 1e-6
    assert float((bufs["enhance.conv.1.running_var"].cpu() - torch.from_numpy(g["bn_running_var"])).abs().max()) < 1e-6
    assert int(bufs["enhance.conv.1.num_batches_tracked"]) == 3
    grads = {n: torch.zeros_like(p) for n, p in params.items()}
    loss, terms = eng.loss_grads(grads)
    # per-term check against the oracle's decomposition (same inputs), total against the reference golden
    tr = oracle.OracleTrainer(oracle.to_torch_state(synth.make_state(seed)), is_WB=wb)
    _, oterms, _, _ = tr.loss(x.cpu(), True)
    for k, nm in enumerate(importlib.import_module("zero-tig_amd.engine").TERM_NAMES):
        assert abs(float(terms[k]) - float(oterms[nm])) <= 2e-4 * abs(float(oterms[nm])) + 1e-6, (nm, float(terms[k]), float(oterms[nm]))
    assert abs(float(loss) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    gn = np.sqrt(sum(float((g["grad:" + n].astype(np.float64) ** 2).sum()) for n in grads))
    for n, gr in grads.items():
        ref = g["grad:" + n]
        if n == "enhance.conv.0.bias":      # analytically zero (bias in front of train-mode BN)
            assert float(gr.abs().max()) <= 1e-5 * gn
        else:
            assert rel_l2(gr, ref) < 1e-3, (n, rel_l2(gr, ref))
