"""Pin the CPU oracle (oracle/zt_oracle.py) to fixtures produced by the reference itself
(tools/make_golden.py, build container).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import frames


def _close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), "%s: max err %.3e (tol %.1e) at %s" % (what, err.max(), atol, np.unravel_index(err.argmax(), err.shape))


ZERO_GRAD = "enhance.conv.0.bias"     # bias in front of train-mode BN: analytically zero gradient, numerically noise


def _check_grads(grads, g, tol):
    gn = np.sqrt(sum(float((np.asarray(g["grad:" + n], np.float64) ** 2).sum()) for n in grads))
    for n, gr in grads.items():
        gr = gr.numpy() if hasattr(gr, "numpy") else gr
        if n == ZERO_GRAD:
            assert np.abs(gr).max() <= 1e-5 * gn and np.abs(g["grad:" + n]).max() <= 1e-5 * gn
        else:
            assert _rel_l2(gr, g["grad:" + n]) < tol, (n, _rel_l2(gr, g["grad:" + n]))


def _rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.mark.parametrize("name,wb", [("g12_newseq_rlv_48x64", False), ("g5_newseq_wb_48x64", True)])
def test_newseq_forward_loss_grads(oracle, synth, golden, name, wb):
    g = golden(name)
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    x = frames(synth, 1, H, W)[0]
    W1 = oracle.to_torch_state(synth.make_state(seed))
    outs, _ = oracle.network_forward(W1, {}, x, True, ofs, training=True)
    for i, o in enumerate(outs):
        if i in (17, 18):
            assert (o.numpy() != g["out%02d" % i]).mean() <= 1e-3, i
        else:
            _close(o.detach().numpy(), g["out%02d" % i], 2e-6, what=oracle.FORWARD_NAMES[i])
    # G8: BN bookkeeping (three updates of the one shared BatchNorm)
    _close(W1["enhance.conv.1.running_mean"].numpy(), g["bn_running_mean"], 1e-7, what="running_mean")
    _close(W1["enhance.conv.1.running_var"].numpy(), g["bn_running_var"], 1e-7, what="running_var")
    assert int(W1["enhance.conv.1.num_batches_tracked"]) == int(g["bn_num_batches_tracked"]) == 3
    tr = oracle.OracleTrainer(oracle.to_torch_state(synth.make_state(seed)), is_WB=wb, of_scale=ofs)
    loss, terms, _, _, grads, _ = tr.step(x, True)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    _check_grads(grads, g, 1e-4)


@pytest.mark.parametrize("name", ["g3_seq_128x160", "g4_seq_132x164"])
def test_sequence_raft_warp(oracle, synth, golden, name):
    g = golden(name)
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    xs = frames(synth, 2, H, W)
    tr = oracle.OracleTrainer(oracle.to_torch_state(synth.make_state(seed)), of_scale=ofs)
    l0, _, _, _ = tr.loss(xs[0], True)
    assert abs(float(l0) - float(g["loss0"])) <= 1e-5 * abs(float(g["loss0"]))
    for n in tr.names:
        tr.W[n].grad = None
    l1, _, outs, aux = tr.loss(xs[1], False)
    l1.backward()
    _close(aux["flow_low"].numpy(), g["flow_low"], 2e-4, what="flow_low")
    _close(aux["flow_up"].numpy(), g["flow_up"], 2e-3, what="flow_up")
    _close(aux["wpH"].numpy(), g["wpH"], 1e-4, what="wpH")
    _close(aux["wps"].numpy(), g["wps"], 1e-4, what="wps")
    _close(outs[13].detach().numpy(), g["last_H3"], 1e-4, what="H3")
    assert abs(float(l1) - float(g["loss1"])) <= 1e-4 * abs(float(g["loss1"]))
    _check_grads({n: tr.W[n].grad for n in tr.names}, g, 1e-3)
    # integer contract: warp tap indices from the reference's own flow
    taps = oracle.warp_taps(torch.from_numpy(g["flow_up"]), H, W).numpy()
    assert np.array_equal(taps[..., 0], g["warp_x0"][:, 0]) and np.array_equal(taps[..., 1], g["warp_y0"][:, 0])


def test_ops(oracle, synth, golden):
    g = golden("g6_ops")
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    a, b = oracle.pair_downsample(x)
    _close(a.numpy(), g["pd1"], 1e-7, what="pd1")
    _close(b.numpy(), g["pd2"], 1e-7, what="pd2")
    _close(oracle.gauss_kernel_2d().numpy(), g["gauss21"], 1e-9, what="gauss21")
    t = oracle.gauss_taps_1d()
    _close(torch.outer(t, t).numpy(), g["gauss21"], 2e-8, what="separable taps")
    _close(oracle.blur21(x).numpy(), g["blur"], 1e-6, what="blur")
    _close(oracle.local_mean_reflect(x).numpy(), g["localmean"], 1e-6, what="localmean")
    _close(oracle.local_variance_zero(x).numpy(), g["localvar"], 1e-6, what="localvar")
    m, ratio = oracle.texture_mask(torch.from_numpy(g["tex_in1"]), torch.from_numpy(g["tex_in2"]))
    _close(ratio.numpy(), g["texratio"], 1e-5, what="texratio")
    assert (m.numpy() != g["texmask"]).mean() <= 1e-3
    assert 0.05 < g["texmask"].mean() < 0.95
    _close(oracle.ycc_flat(x * 0.2).numpy(), g["ycc"], 1e-6, what="ycc")
    assert abs(float(oracle.smooth_loss(x * 0.2, y)) - float(g["smooth"])) < 1e-5 * float(g["smooth"])
    assert abs(float(oracle.tv_loss(y)) - float(g["tv"])) < 1e-5 * float(g["tv"])
    flow = torch.from_numpy(g["warp_flow"])
    assert np.array_equal(oracle.warp_tensor(flow, torch.from_numpy(g["warp_img"])).numpy(), g["warp_out"])
    f1 = torch.from_numpy(synth.normal("ops.f1", (1, 256, 16, 24), 0.0, 1.0, 7))
    f2 = torch.from_numpy(synth.normal("ops.f2", (1, 256, 16, 24), 0.0, 1.0, 7))
    pyr = oracle.corr_pyramid(f1, f2)
    for i, c in enumerate(pyr):
        _close((c if i else c[::7]).numpy(), g["corr_pyr%d" % i], 1e-5, what="pyr%d" % i)
    coords = torch.from_numpy(g["lookup_coords"])
    _close(oracle.corr_lookup(pyr, coords).numpy(), g["lookup_out"], 1e-5, what="lookup")
    assert np.array_equal(oracle.equalize_u8(torch.from_numpy(g["eq_in"])).numpy(), g["eq_out"])
    W3 = oracle.to_torch_state(synth.make_state(3))
    h = torch.tanh(torch.from_numpy(synth.normal("ops.h", (1, 128, 16, 24), 0.0, 1.0, 7)))
    inp = torch.relu(torch.from_numpy(synth.normal("ops.inp", (1, 128, 16, 24), 0.0, 1.0, 7)))
    ys, xs_ = torch.meshgrid(torch.arange(16), torch.arange(24), indexing="ij")
    c0 = torch.stack([xs_, ys], 0).float()[None]
    with torch.no_grad():
        h2, mask, dfl = oracle.update_block(W3, "raft.update_block", h, inp, oracle.corr_lookup(pyr, coords), coords - c0)
        up = oracle.convex_upsample(coords - c0 + dfl, mask)
    _close(h2.numpy(), g["ub_net"], 1e-5, what="ub_net")
    _close(mask[:, ::9].numpy(), g["ub_mask"], 1e-4, what="ub_mask")
    _close(dfl.numpy(), g["ub_dflow"], 1e-5, what="ub_dflow")
    _close(up.numpy(), g["ub_up"], 1e-4, what="ub_up")


def test_adam_three_steps(oracle, synth, golden):
    g = golden("g7_adam_128x160")
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    xs = frames(synth, 3, H, W)
    tr = oracle.OracleTrainer(oracle.to_torch_state(synth.make_state(seed)), of_scale=ofs)
    for t, x in enumerate(xs):
        loss, _, _, _, _, gn = tr.step(x, t == 0)
        assert abs(float(loss) - float(g["loss%d" % t])) <= 2e-4 * abs(float(g["loss%d" % t])), t
        assert abs(float(gn) - float(g["gnorm%d" % t])) <= 2e-3 * float(g["gnorm%d" % t]), t
    for n in tr.names:
        # the BN-cancelled bias sees pure rounding-noise gradients which Adam normalises to +-lr steps (3 steps here)
        tol = 6.1e-4 if n == ZERO_GRAD else 3e-5
        _close(tr.W[n].detach().numpy(), g["w:" + n], tol, what=n)   # Adam normalises tiny grads: 0.1 step
    _close(tr.W["enhance.conv.1.running_mean"].numpy(), g["bn_running_mean"], 5e-4, what="rm")   # inherits the noise-driven bias above
    assert int(tr.W["enhance.conv.1.num_batches_tracked"]) == int(g["bn_num_batches_tracked"]) == 9


def test_finetune(oracle, synth, golden):
    g = golden("g9_finetune_128x160")
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    xs = frames(synth, 2, H, W)
    Wt = oracle.to_torch_state(synth.make_state(seed))
    cache = {}
    with torch.no_grad():
        for t, x in enumerate(xs):
            H2, H3, s3 = oracle.finetune_forward(Wt, cache, x, t == 0, ofs)
            tol = 2e-6 if t == 0 else 1e-4
            _close(H2.numpy(), g["H2_%d" % t], tol, what="H2_%d" % t)
            _close(H3.numpy(), g["H3_%d" % t], tol, what="H3_%d" % t)
            _close(s3.numpy(), g["s3_%d" % t], tol, what="s3_%d" % t)


def test_equalize_properties(oracle):
    ramp = torch.arange(256, dtype=torch.uint8).repeat(3, 4, 1)          # flat histogram -> near-identity LUT
    out = oracle.equalize_u8(ramp)
    assert out.dtype == torch.uint8 and out.shape == ramp.shape
    lut = out[0, 0].to(torch.int64)
    assert (lut[1:] >= lut[:-1]).all()
    const = torch.full((3, 8, 8), 77, dtype=torch.uint8)                  # step == 0 -> unchanged
    assert torch.equal(oracle.equalize_u8(const), const)
