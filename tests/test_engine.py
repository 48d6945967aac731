"""End-to-end parity of the hand-written forward / fused loss+gradient / backward plan against the CPU oracle
and the reference-generated goldens."""
import importlib
import os

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


def _skip_heavy_emu(bname):
    """RAFT through the fiber emulator takes minutes; the CPU suite runs it only on request (ZT_EMU_FULL=1)."""
    if bname == "emu" and not os.environ.get("ZT_EMU_FULL"):
        pytest.skip("heavy emulator case (set ZT_EMU_FULL=1); covered by -m gpu")


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


def _network(ops, dev, synth, seed, dataset="RLV", of_scale=1, cls="Network", pretrain=None, precision="fp32"):
    import argparse
    net_mod = importlib.import_module("zero-tig_amd.network")
    args = argparse.Namespace(dataset=dataset, of_scale=of_scale)
    if pretrain is not None:
        args.model_pretrain = pretrain
    net = getattr(net_mod, cls)(args, ops=ops, precision=precision)
    if pretrain is None:
        st = synth.make_state(seed)
        sd = net.state_dict()
        assert set(sd.keys()) == set(st.keys())
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    return net.to(dev)


@pytest.mark.parametrize("name", ["g3_seq_128x160", "g4_seq_132x164"])
def test_sequence_raft_warp_loss_grads(backend, synth, name):
    """Two-frame sequence through the drop-in Network: frame 1 runs downscale + equalize + RAFT(12) + fused warp."""
    ops, dev, bname = backend
    _skip_heavy_emu(bname)
    g = load_golden(name)
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    xs = [f.to(dev) for f in frames(synth, 2, H, W)]
    net = _network(ops, dev, synth, seed, of_scale=ofs)
    net.train()
    net.is_new_seq = True
    l0 = net._loss(xs[0])
    assert abs(float(l0) - float(g["loss0"])) <= 1e-4 * abs(float(g["loss0"]))
    net.zero_grad()
    net.is_new_seq = False
    l1 = net._loss(xs[1])
    l1.backward()
    wpH, wps = net.last_H3_wp, net.last_s3_wp
    assert float(np.abs(wpH.cpu().numpy() - g["wpH"]).max()) < 2e-4
    assert float(np.abs(wps.cpu().numpy() - g["wps"]).max()) < 2e-4
    assert float(np.abs(net.last_H3.cpu().numpy() - g["last_H3"]).max()) < 2e-4
    assert abs(float(l1) - float(g["loss1"])) <= 2e-4 * abs(float(g["loss1"]))
    gn = np.sqrt(sum(float((g[k].astype(np.float64) ** 2).sum()) for k in g.files if k.startswith("grad:")))
    for n, p in net.named_parameters():
        if not p.requires_grad or n.startswith("enhance.blocks"):
            continue
        ref = g["grad:" + n]
        if n == "enhance.conv.0.bias":
            assert float(p.grad.abs().max()) <= 1e-5 * gn
        else:
            assert rel_l2(p.grad, ref) < 2e-3, (n, rel_l2(p.grad, ref))


def test_raft_flow_golden(backend, synth):
    """RAFT plan alone on the reference's own RAFT inputs (golden raft_img1/2): flow_low / flow_up parity."""
    ops, dev, bname = backend
    _skip_heavy_emu(bname)
    g = load_golden("g4_seq_132x164")
    net = _network(ops, dev, synth, int(g["meta"][2]), of_scale=1)
    _, rp = net._plan()
    img1, img2 = torch.from_numpy(g["raft_img1"]).to(dev), torch.from_numpy(g["raft_img2"]).to(dev)
    h, w = img1.shape[-2:]
    q = img2.to(torch.uint8).view(3, h * w).contiguous()
    lut = torch.arange(256, dtype=torch.int32, device=dev).repeat(3, 1).contiguous()
    x2 = ops.raft_pack_input(img1, q, lut, h, w)
    flow_low, flow_up = rp.run(x2)
    assert float(np.abs(flow_low.cpu().numpy() - g["flow_low"]).max()) < 5e-4
    assert float(np.abs(flow_up.cpu().numpy() - g["flow_up"]).max()) < 4e-3


def test_raft_context_encoder_bn_fold(backend, synth, monkeypatch):
    """bf16 plan: the context encoder's eval-mode BatchNorms folded into the convs in front of them (+ ReLU / residual epilogues,
    extractor.py:45-56) against the same plan with separate normalisation passes and against the fp32 plan."""
    ops, dev, bname = backend
    raft = importlib.import_module("zero-tig_amd.raft")
    st = synth.make_state(5)
    for k in list(st.keys()):           # non-trivial frozen statistics
        if k.startswith("raft.cnet") and k.endswith("running_mean"):
            st[k][:] = synth.normal(k, st[k].shape, 0.0, 0.2, 3)
        if k.startswith("raft.cnet") and k.endswith("running_var"):
            st[k][:] = synth.uniform(k, st[k].shape, 0.3, 1.5, 3)
    W = {k: torch.from_numpy(np.array(v)).to(dev) for k, v in st.items() if k.startswith("raft.")}
    h, w = 48, 64
    x = torch.from_numpy(synth.uniform("cx", (1, h, w, 8), -1.0, 1.0, 11)).to(dev)
    x[..., 3:] = 0
    outs = {}
    for mode in ("fold", "plain", "fp32"):
        monkeypatch.setenv("ZT_RAFT_FOLD_BN", "0" if mode == "plain" else "1")
        plan = raft.RaftPlan(ops, W, dev, precision="fp32" if mode == "fp32" else "bf16")
        assert bool(plan.folded) == (mode == "fold")
        xin = x if mode == "fp32" else x.to(torch.bfloat16)
        outs[mode] = plan._encoder("cnet", xin.contiguous(), "batch").float().cpu()
    assert outs["fold"].shape == outs["plain"].shape == outs["fp32"].shape
    e_fold, e_plain = rel_l2(outs["fold"], outs["fp32"]), rel_l2(outs["plain"], outs["fp32"])
    print("cnet encoder rel-L2 vs fp32: folded %.3e, separate passes %.3e" % (e_fold, e_plain))
    assert e_fold < 2e-2 and e_fold < 1.5 * e_plain + 1e-3, (e_fold, e_plain)


def test_adam_three_steps(backend, synth):
    """G7: three iterations of the reference loop (loss.backward, clip_grad_norm_(5), Adam.step) through ClipAdam."""
    ops, dev, bname = backend
    _skip_heavy_emu(bname)
    optim = importlib.import_module("zero-tig_amd.optim")
    g = load_golden("g7_adam_128x160")
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    net = _network(ops, dev, synth, seed, of_scale=ofs)
    net.train()
    opt = optim.ClipAdam(net, lr=1e-4, betas=(0.9, 0.999), weight_decay=3e-4, max_norm=5.0)
    for t, x in enumerate(frames(synth, 3, H, W)):
        net.is_new_seq = (t == 0)
        opt.zero_grad()
        loss = net._loss(x.to(dev))
        loss.backward()
        gn = opt.step()
        assert abs(float(loss) - float(g["loss%d" % t])) <= 3e-4 * abs(float(g["loss%d" % t])), (t, float(loss))
        assert abs(float(gn) - float(g["gnorm%d" % t])) <= 3e-3 * float(g["gnorm%d" % t]), t
    # Adam normalises every element's step to ~lr, so an element whose gradient is ~0 can flip sign under fp32 reduction-order
    # differences: compare the UPDATE vectors (3 steps of <= 1e-4) in rel-L2 and bound the worst element by 2.5 steps.
    st0 = synth.make_state(seed)
    for n, p in net.named_parameters():
        if p.requires_grad and not n.startswith("enhance.blocks") and n != "enhance.conv.0.bias":
            w0 = torch.from_numpy(np.array(st0[n]))
            d_ref, d_got = torch.from_numpy(g["w:" + n]) - w0, p.detach().cpu() - w0
            assert float((d_got - d_ref).abs().max()) < 2.5e-4, n
            assert rel_l2(d_got, d_ref) < 0.08, (n, rel_l2(d_got, d_ref))
    assert int(net.enhance.conv[1].num_batches_tracked) == 9


def test_finetune_golden(backend, synth, tmp_path):
    """G9: Finetunemodel (inference twin) on a new-sequence frame and a RAFT-warped second frame."""
    ops, dev, bname = backend
    _skip_heavy_emu(bname)
    g = load_golden("g9_finetune_128x160")
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    st = synth.make_state(seed)
    ck = str(tmp_path / "ck.pt")
    torch.save({k: torch.from_numpy(np.array(v)) for k, v in st.items()}, ck)
    net = _network(ops, dev, synth, seed, of_scale=ofs, cls="Finetunemodel", pretrain=ck)
    net.raft.load_state_dict({k[5:]: torch.from_numpy(np.array(v)) for k, v in st.items() if k.startswith("raft.")})
    net = net.to(dev)
    net.eval()
    with torch.no_grad():
        for t, x in enumerate(frames(synth, 2, H, W)):
            net.is_new_seq = (t == 0)
            H2, H3, s3 = net(x.to(dev))
            tol = 2e-5 if t == 0 else 2e-4
            for nm, o in (("H2", H2), ("H3", H3), ("s3", s3)):
                assert float(np.abs(o.cpu().numpy() - g["%s_%d" % (nm, t)]).max()) < tol, (nm, t)


def test_bf16_mode_psnr_gate(backend, synth, oracle):
    """Throughput mode (bf16 activations/weights, fp32 accumulate): enhanced-frame PSNR within 0.01 dB of the reference
    (evals.py:83-85 definition vs the synthetic clean frame), loss and gradients close to the fp32 reference values."""
    ops, dev, bname = backend
    g = load_golden("g12_newseq_rlv_48x64")
    H, W, seed, _ = [int(v) for v in g["meta"]]
    x = frames(synth, 1, H, W)[0]
    net = _network(ops, dev, synth, seed, of_scale=3, precision="bf16").train()
    net.is_new_seq = True
    loss = net._loss(x.to(dev))
    loss.backward()
    H3 = net.last_H3.cpu()
    ref_H3 = torch.from_numpy(g["out13"])
    clean = torch.from_numpy(synth.clean_frame(0, H, W)).float()[None]
    d_psnr = abs(oracle.psnr_u8(H3, clean) - oracle.psnr_u8(ref_H3, clean))
    assert d_psnr <= 0.01, d_psnr
    assert oracle.psnr_u8(H3, ref_H3) > 45.0
    assert abs(float(loss) - float(g["loss"])) <= 1e-2 * abs(float(g["loss"]))
    for n, p in net.named_parameters():
        if p.requires_grad and not n.startswith("enhance.blocks") and n != "enhance.conv.0.bias":
            assert rel_l2(p.grad, g["grad:" + n]) < 6e-2, (n, rel_l2(p.grad, g["grad:" + n]))


def test_bf16_sequence_flow_and_psnr(backend, synth, oracle):
    """Throughput mode through RAFT + warp: flow stays close to the fp32 reference flow, and the enhanced second frame passes
    the PSNR gate (|dPSNR| <= 0.01 dB vs the reference output, evals.py:83-85 definition against the clean frame)."""
    ops, dev, bname = backend
    _skip_heavy_emu(bname)
    g = load_golden("g3_seq_128x160")
    H, W, seed, ofs = [int(v) for v in g["meta"]]
    xs = [f.to(dev) for f in frames(synth, 2, H, W)]
    net = _network(ops, dev, synth, seed, of_scale=ofs, precision="bf16").train()
    net.is_new_seq = True
    net._loss(xs[0])
    net.is_new_seq = False
    l1 = net._loss(xs[1])
    clean = torch.from_numpy(synth.clean_frame(1, H, W)).float()[None]
    H3, ref = net.last_H3.cpu(), torch.from_numpy(g["last_H3"])
    assert abs(oracle.psnr_u8(H3, clean) - oracle.psnr_u8(ref, clean)) <= 0.01
    assert abs(float(l1) - float(g["loss1"])) <= 2e-2 * abs(float(g["loss1"]))
    wp_err = float((net.last_H3_wp.cpu() - torch.from_numpy(g["wpH"])).abs().mean())
    assert wp_err < 5e-3, wp_err


def test_eval_mode_bn_training_quirk(backend, synth, oracle):
    """SURVEY A-14: from epoch 1 on the reference trains with BatchNorm in eval mode (train.py:138 never switches back).
    Forward uses running stats, backward treats them as constants; running stats are not updated."""
    ops, dev, bname = backend
    H, W, seed = 48, 64, 1
    x = frames(synth, 1, H, W)[0]
    st = synth.make_state(seed)
    st["enhance.conv.1.running_mean"][:] = synth.normal("rm", (64,), 0.0, 0.05, 9)
    st["enhance.conv.1.running_var"][:] = synth.uniform("rv", (64,), 0.01, 0.05, 9)
    net = _network(ops, dev, synth, seed, of_scale=3)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    net = net.to(dev)
    net.eval()
    net.is_new_seq = True
    loss = net._loss(x.to(dev))
    loss.backward()
    tr = oracle.OracleTrainer(oracle.to_torch_state(st))
    tr.training = False
    ref, _, _, _ = tr.loss(x, True)
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-4 * abs(float(ref.detach()))
    for n, p in net.named_parameters():
        if p.requires_grad and not n.startswith("enhance.blocks"):
            assert rel_l2(p.grad, tr.W[n].grad) < 2e-3, (n, rel_l2(p.grad, tr.W[n].grad))
    assert int(net.enhance.conv[1].num_batches_tracked) == 0
    assert float((net.enhance.conv[1].running_mean.cpu() - torch.from_numpy(st["enhance.conv.1.running_mean"])).abs().max()) == 0.0


def test_clipadam_rebinds_stray_grads(backend, synth):
    """`model.zero_grad()` (set_to_none) detaches p.grad from the flat bucket; step() must pick the stray gradients up
    (same update as with optimizer.zero_grad()) instead of silently applying zeros, and refuse parameters that moved."""
    ops, dev, bname = backend
    optim = importlib.import_module("zero-tig_amd.optim")
    x = frames(synth, 1, 48, 64)[0].to(dev)
    flats = []
    for mode in ("opt", "model"):
        net = _network(ops, dev, synth, 1, of_scale=3).train()
        opt = optim.ClipAdam(net)
        (opt if mode == "opt" else net).zero_grad()
        net.is_new_seq = True
        net._loss(x).backward()
        gn = float(opt.step())
        assert gn > 0
        flats.append(opt.fp.flat.clone())
        assert all(p.grad.data_ptr() == opt.fp.grad.data_ptr() + 4 * o for p, o in zip(opt.fp.params, opt.fp.offsets))
    assert torch.equal(flats[0], flats[1])
    for p in opt.fp.params[:1]:
        p.data = p.data.clone()
    with pytest.raises(RuntimeError):
        opt.step()


def test_trainstep_eager_equals_manual_loop(backend, synth):
    """optim.TrainStep (gradients written straight into the flat bucket, no autograd) == zero_grad / _loss / backward / step."""
    ops, dev, bname = backend
    optim = importlib.import_module("zero-tig_amd.optim")
    xs = [f.to(dev) for f in frames(synth, 2, 48, 64)]
    out = []
    for mode in ("manual", "trainstep"):
        net = _network(ops, dev, synth, 1, of_scale=3).train()
        opt = optim.ClipAdam(net)
        ts = optim.TrainStep(net, opt, use_graph=False)
        ls = []
        for x in xs:                                   # two new-sequence frames (RAFT through the emulator takes minutes)
            if mode == "manual":
                net.is_new_seq = True
                opt.zero_grad()
                loss = net._loss(x)
                loss.backward()
                opt.step()
            else:
                loss = ts(x, is_new_seq=True)
            ls.append(float(loss.detach()))
        out.append((ls, opt.fp.flat.clone()))
    assert out[0][0] == out[1][0] and torch.equal(out[0][1], out[1][1])


@pytest.mark.gpu
def test_trainstep_hipgraph_replay_equals_eager(hip_ops, synth):
    """The captured hipGraph of the steady-state step replays to the same bits as eager launches: losses of every frame, final
    weights, Adam moments, BN running statistics and the recurrent cache (6 frames: new-sequence, eager steady-state, capture +
    replay, replay, a second new-sequence frame, replay)."""
    ops, dev = hip_ops
    optim = importlib.import_module("zero-tig_amd.optim")
    H, W = 128, 160
    host = frames(synth, 6, H, W)
    res = []
    for use_graph in (False, True):
        net = _network(ops, dev, synth, 1, of_scale=1).train()
        opt = optim.ClipAdam(net)
        ts = optim.TrainStep(net, opt, use_graph=use_graph)
        # a new sequence starts again at frame 4 (after the capture): that frame runs eagerly, the replay resumes from its cache
        ls = [float(ts(x.pin_memory() if use_graph else x.to(dev), is_new_seq=(t in (0, 4))).detach()) for t, x in enumerate(host)]
        assert (ts.graph is not None) == use_graph
        bn = net.enhance.conv[1]
        res.append((ls, opt.fp.flat.clone(), opt.m.clone(), bn.running_mean.clone(), int(bn.num_batches_tracked), net.last_H3.clone()))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for a, b in zip(res[0][1:], res[1][1:]):
        assert (a == b) if isinstance(a, int) else torch.equal(a, b)
