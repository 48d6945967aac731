"""Parity at BASELINE.json's real sizes (GPU only).
 * config 2 (540x960 frame pair, fp32): enhance + RAFT flow + backward warp against the CPU oracle on the same inputs;
 * 1080x1920: integer contracts bit-exact against the oracle (warp taps, equalize), and size-independent properties of the
   kernels where running the oracle would take minutes (adjointness, determinism, loss-term consistency, bf16 vs fp32)."""
import argparse
import importlib

import numpy as np
import pytest
import torch

from conftest import frames

pytestmark = pytest.mark.gpu


def _net(ops, dev, synth, seed, of_scale, precision="fp32", dataset="RLV"):
    net_mod = importlib.import_module("zero-tig_amd.network")
    net = net_mod.Network(argparse.Namespace(dataset=dataset, of_scale=of_scale), ops=ops, precision=precision)
    st = synth.make_state(seed)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    return net.to(dev).train()


def test_540p_pair_forward_flow_warp_vs_oracle(hip_ops, synth, oracle):
    ops, dev = hip_ops
    H, W, ofs = 540, 960, 3              # RAFT input 180x320 -> padded 184x320 (flow_up keeps the padded size; scales swapped)
    xs = frames(synth, 2, H, W)
    net = _net(ops, dev, synth, 1, ofs)
    Wt = oracle.to_torch_state(synth.make_state(1))
    cache = {}
    with torch.no_grad():
        for t, x in enumerate(xs):
            net.is_new_seq = (t == 0)
            prev = (net.last_H3, net.last_s3)
            outs = net(x.to(dev))
            ref, aux = oracle.network_forward(Wt, cache, x, t == 0, ofs, training=True)
            cache["last_H3"], cache["last_s3"] = ref[13], ref[14]
            net.update_H3(outs[13], outs[14])
            # FREE-RUNNING RAFT gate (the flow-error bound): frame 0 is a pure fp32 pipeline.  Frame 1 goes through 12 GRU
            # iterations of a randomly initialised RAFT, whose fp32 reduction-order differences move the flow by ~1e-3 px; that
            # shows up at a few high-gradient pixels of the warp.  The SURVEY 8(d) gate (2e-5) for everything downstream of RAFT
            # is held by test_teacher_forced_pair_vs_oracle below, which warps with the oracle's flow.
            tol_max, tol_mean = (2e-5, 2e-6) if t == 0 else (3e-3, 3e-5)
            for i in (2, 3, 6, 13, 14):                      # L2, s2, H2, H3, s3
                d = (outs[i].cpu() - ref[i]).abs()
                assert float(d.max()) < tol_max and float(d.mean()) < tol_mean, (t, oracle.FORWARD_NAMES[i], float(d.max()), float(d.mean()))
            if t == 1:
                d = (net.last_H3_wp.cpu() - aux["wpH"]).abs()
                assert float(d.max()) < 3e-3 and float(d.mean()) < 3e-5, (float(d.max()), float(d.mean()))
                _, rp = net._plan()
                _, _, fl, fu = rp.update_cache(prev[0], prev[1], outs[2], ofs, want_aux=True)
                efl, efu = float((fl.cpu() - aux["flow_low"]).abs().max()), float((fu.cpu() - aux["flow_up"]).abs().max())
                assert efl < 5e-3 and efu < 4e-2, (efl, efu)          # explicit flow-error bound of the free-running RAFT (pixels)
            mism = float((outs[18].cpu() != ref[18]).float().mean())
            assert mism <= 1e-3, mism


def test_1080p_integer_contracts_bit_exact(hip_ops, synth, oracle):
    ops, dev = hip_ops
    H, W = 1080, 1920
    g = torch.Generator().manual_seed(0)
    flow = torch.randn(1, 2, 360, 640, generator=g) * 3.0
    img = torch.rand(1, 3, H, W, generator=g)
    out, _, taps = ops.warp2(flow.to(dev), img.to(dev), None, want_taps=True)
    assert torch.equal(taps.cpu(), oracle.warp_taps(flow, H, W)[0])
    assert torch.equal(out.cpu(), oracle.warp_tensor(flow, img))
    x255 = (torch.rand(1, 3, 360, 640, generator=g) ** 2 * 255.0)
    q, hist, lut = ops.equalize_prepare(x255.to(dev))
    assert int(hist.sum()) == 3 * 360 * 640
    ref = oracle.equalize_u8(x255.to(torch.uint8))
    got = torch.gather(lut.cpu().long(), 1, q.cpu().long()).view(1, 3, 360, 640).to(torch.uint8)
    assert torch.equal(got, ref)
    a = ops.resize_bilinear(img.to(dev), 360, 640, 255.0)
    assert torch.equal(a.cpu(), torch.nn.functional.interpolate(img, (360, 640), mode="bilinear") * 255)


def test_1080p_adjoints(hip_ops):
    ops, dev = hip_ops
    g = torch.Generator().manual_seed(1)
    x = torch.rand(1, 3, 1080, 1920, generator=g).to(dev)
    y = torch.rand(1, 3, 1080, 1920, generator=g).to(dev)
    yh = torch.rand(1, 3, 540, 960, generator=g).to(dev)

    def dot(a, b):
        return float((a.double() * b.double()).sum())
    assert abs(dot(ops.blur21(x), y) - dot(x, ops.blur21_adj(y))) <= 1e-6 * abs(dot(x, y))
    assert abs(dot(ops.box5_reflect(x), y) - dot(x, ops.box5_reflect_adj(y))) <= 1e-6 * abs(dot(x, y))
    a, b = ops.pair_down(x)
    assert abs(dot(a, yh) + dot(b, yh) - dot(x, ops.pair_down_adj(yh, yh, 1080, 1920))) <= 1e-6 * abs(dot(x, y))


def test_1080p_step_determinism_and_consistency(hip_ops, synth):
    ops, dev = hip_ops
    x = frames(synth, 2, 1080, 1920)
    res = {}
    for prec in ("fp32", "bf16"):
        for rep in range(2 if prec == "fp32" else 1):
            net = _net(ops, dev, synth, 1, 3, prec)
            losses, grads = [], None
            for t in range(2):
                net.zero_grad()
                net.is_new_seq = (t == 0)
                loss = net._loss(x[t].to(dev))
                loss.backward()
                losses.append(float(loss.detach()))
            terms = net.last_terms.cpu()
            assert abs(float(terms.double().sum()) - losses[-1]) <= 1e-5 * abs(losses[-1])      # 17 weighted terms add up
            grads = torch.cat([p.grad.flatten() for p in net.parameters() if p.requires_grad]).cpu()
            assert torch.isfinite(grads).all()
            res[(prec, rep)] = (losses, grads, net.last_H3.cpu())
    # no float atomics anywhere: the whole 1080p step (incl. RAFT) is bit-reproducible run to run
    assert res[("fp32", 0)][0] == res[("fp32", 1)][0]
    assert torch.equal(res[("fp32", 0)][1], res[("fp32", 1)][1])
    # bf16 throughput mode tracks the fp32 parity mode
    lf, lb = res[("fp32", 0)][0][1], res[("bf16", 0)][0][1]
    assert abs(lf - lb) <= 2e-2 * abs(lf), (lf, lb)
    gf, gb = res[("fp32", 0)][1].double(), res[("bf16", 0)][1].double()
    assert float((gf - gb).norm() / gf.norm()) < 0.1
    clean = torch.from_numpy(synth.clean_frame(1, 1080, 1920)).float()[None]

    def psnr(a, b):
        a8, b8 = torch.clamp(torch.round(a * 255), 0, 255), torch.clamp(torch.round(b * 255), 0, 255)
        return 10 * np.log10(255.0 ** 2 / float(((a8 - b8) ** 2).mean()))
    assert abs(psnr(res[("fp32", 0)][2], clean) - psnr(res[("bf16", 0)][2], clean)) <= 0.01


def _teacher_force(net, ops, flow_up):
    """Test plumbing (kept out of the product path): replace the bound RAFT plan's `update_cache` by a warp with the given
    [1,2,Hp,Wp] flow (the oracle's), which isolates everything downstream of RAFT at the new-sequence tolerance."""
    _, rp = net._plan()
    rp.update_cache = lambda H3, s3, L2, of_scale, **kw: ops.warp2(flow_up.contiguous(), H3.contiguous(), s3.contiguous())


# bf16 throughput mode (the benchmarked mode) DIRECTLY against the CPU oracle, teacher-forced like the fp32 leg.  Gates:
# PSNR(H3_bf16, H3_oracle) >= 50 dB (evals.py:83-85 definition; measured 55.2-56.9 dB at 540p / 1080p, gpurun_out/r03a), loss rel <=
# 1e-3 (measured <= 2.7e-4), every parameter gradient rel-L2 <= 2e-2 (measured <= 1.4e-2, worst: enhance.in_conv.0.bias),
# |PSNR(H3_bf16, clean) - PSNR(H3_oracle, clean)| <= 0.01 dB (BASELINE.json's quality target; measured <= 0.0023 dB).
BF16_PSNR_DB, BF16_LOSS_REL, BF16_GRAD_REL, BF16_DPSNR_DB = 50.0, 1e-3, 2e-2, 0.01


def _bf16_vs_oracle(netb, ops, dev, synth, oracle, tr, t, x, cache_before, aux, ref, ref_loss, H, W):
    bad = []
    netb.zero_grad()
    netb.is_new_seq = (t == 0)
    if t == 1:
        netb.last_H3, netb.last_s3 = cache_before["last_H3"].to(dev), cache_before["last_s3"].to(dev)
        _teacher_force(netb, ops, aux["flow_up"].to(dev))
    loss = netb._loss(x.to(dev))
    loss.backward()
    sv = netb._eng.sv
    H3b, H3o = sv["H3"].cpu(), ref[13].detach()
    ps = oracle.psnr_u8(H3b, H3o)
    clean = torch.from_numpy(synth.clean_frame(t, H, W)).float()[None]
    dps = abs(oracle.psnr_u8(H3b, clean) - oracle.psnr_u8(H3o, clean))
    lr = abs(float(loss.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach()))
    print("bf16 frame %d: PSNR(H3_bf16, H3_oracle) %.2f dB, dPSNR-vs-clean %.4f dB, loss %.6f oracle %.6f rel %.2e, max-abs H3 %.2e s3 %.2e"
          % (t, ps, dps, float(loss.detach()), float(ref_loss.detach()), lr, float((H3b - H3o).abs().max()),
             float((sv["s3"].cpu() - ref[14].detach()).abs().max())))
    if ps < BF16_PSNR_DB:
        bad.append((t, "bf16 PSNR(H3)", ps))
    if dps > BF16_DPSNR_DB:
        bad.append((t, "bf16 dPSNR-vs-clean", dps))
    if lr > BF16_LOSS_REL:
        bad.append((t, "bf16 loss", lr))
    for n, p in netb.named_parameters():
        if not p.requires_grad or n.startswith("enhance.blocks") or n == "enhance.conv.0.bias":
            continue
        r = _rel_l2(p.grad, tr.W[n].grad)
        print("bf16 frame %d grad %s rel-L2 %.2e" % (t, n, r))
        if r > BF16_GRAD_REL:
            bad.append((t, "bf16 grad " + n, r))
    return bad


def _rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("H,W", [(540, 960), (1080, 1920)], ids=["540p", "1080p"])
def test_teacher_forced_pair_vs_oracle(hip_ops, synth, oracle, H, W):
    """SURVEY 8(d) gates at BASELINE sizes, fp32 mode, against the CPU oracle on the same inputs and weights:
    frame 0 (new sequence): outputs <= 2e-5, loss rel <= 1e-4, parameter gradients rel-L2 <= 1e-3;
    frame 1 (steady state): the oracle's frame-0 cache and its RAFT `flow_up` are injected, so that warp + enhancement nets + loss +
    backward are compared at the SAME gates with RAFT's reduction-order noise taken out (the free-running RAFT has its own test
    with an explicit flow-error bound).  The warp itself must then be bit-exact (integer taps, identical fp32 sequence)."""
    ops, dev = hip_ops
    ofs = 3
    xs = frames(synth, 2, H, W)
    net = _net(ops, dev, synth, 1, ofs)
    netb = _net(ops, dev, synth, 1, ofs, "bf16")
    tr = oracle.OracleTrainer(oracle.to_torch_state(synth.make_state(1)), of_scale=ofs)
    bad = []
    for t, x in enumerate(xs):
        for n in tr.names:
            tr.W[n].grad = None
        cache_before = dict(tr.cache)
        ref_loss, _, ref, aux = tr.loss(x, t == 0)
        ref_loss.backward()
        bad += _bf16_vs_oracle(netb, ops, dev, synth, oracle, tr, t, x, cache_before, aux, ref, ref_loss, H, W)
        net.zero_grad()
        net.is_new_seq = (t == 0)
        if t == 1:
            net.last_H3, net.last_s3 = cache_before["last_H3"].to(dev), cache_before["last_s3"].to(dev)
            _teacher_force(net, ops, aux["flow_up"].to(dev))
        loss = net._loss(x.to(dev))
        loss.backward()
        eng = net._eng
        if t == 1:
            assert torch.equal(net.last_H3_wp.cpu(), aux["wpH"]) and torch.equal(net.last_s3_wp.cpu(), aux["wps"])
        for nm, i in (("L2", 2), ("s2", 3), ("H2", 6), ("H3", 13), ("s3", 14)):
            got = {"L2": eng.sv["L2"], "s2": eng.sv["s2"], "H2": eng.sv["H2"], "H3": eng.sv["H3"], "s3": eng.sv["s3"]}[nm]
            e = float((got.cpu() - ref[i].detach()).abs().max())
            print("frame %d %s max-abs %.3e" % (t, nm, e))
            if e > 2e-5:
                bad.append((t, nm, e))
        lr = abs(float(loss.detach()) - float(ref_loss.detach())) / abs(float(ref_loss.detach()))
        print("frame %d loss %.6f oracle %.6f rel %.2e" % (t, float(loss.detach()), float(ref_loss.detach()), lr))
        if lr > 1e-4:
            bad.append((t, "loss", lr))
        mism = float((eng.sv["m_h"].cpu() != ref[18]).float().mean())
        if mism > 1e-3:
            bad.append((t, "mask", mism))
        gn = np.sqrt(sum(float((tr.W[n].grad.double() ** 2).sum()) for n in tr.names))
        for n, p in net.named_parameters():
            if not p.requires_grad or n.startswith("enhance.blocks"):
                continue
            if n == "enhance.conv.0.bias":                 # analytically zero (bias in front of train-mode BN)
                if float(p.grad.abs().max()) > 1e-5 * gn:
                    bad.append((t, n, float(p.grad.abs().max())))
                continue
            r = _rel_l2(p.grad, tr.W[n].grad)
            print("frame %d grad %s rel-L2 %.2e" % (t, n, r))
            if r > 1e-3:
                bad.append((t, n, r))
    assert not bad, bad


def test_256_newseq_forward_vs_oracle(hip_ops, synth, oracle):
    """BASELINE config 1: one 256x256 frame, Network.forward on a new sequence (no RAFT, no warp): all 23 outputs vs the oracle."""
    ops, dev = hip_ops
    x = frames(synth, 1, 256, 256)[0]
    net = _net(ops, dev, synth, 1, 3)
    net.is_new_seq = True
    with torch.no_grad():
        outs = net(x.to(dev))
        ref, _ = oracle.network_forward(oracle.to_torch_state(synth.make_state(1)), {}, x, True, 3, training=True)
    for i, (o, r) in enumerate(zip(outs, ref)):
        if i in (17, 18):
            assert float((o.cpu() != r).float().mean()) <= 1e-3, i
        else:
            e = float((o.cpu() - r).abs().max())
            assert e < 2e-5, (oracle.FORWARD_NAMES[i], e)


def test_4k_underwater_bf16_step(hip_ops, synth, oracle):
    """BASELINE config 5 (one rank of it): 2160x3840, dataset 'underwater' (is_WB), bf16 throughput mode.  The oracle cannot run
    a 4K network step in test time, so: integer contracts bit-exact against the oracle on the step's OWN flow / frames (warp taps
    and values, equalize), the loss kernels (is_WB branch of loss.py:26-29) against the oracle's LossFunction evaluated on the
    step's own 23 outputs, and the size-independent properties (determinism, 17 terms add up, finite gradients)."""
    ops, dev = hip_ops
    H, W, ofs = 2160, 3840, 3
    xs = [f.to(dev) for f in frames(synth, 2, H, W)]
    res = []
    for rep in range(2):
        net = _net(ops, dev, synth, 1, ofs, "bf16", dataset="underwater")
        assert net.is_WB
        losses = []
        for t in range(2):
            net.zero_grad()
            net.is_new_seq = (t == 0)
            if t == 1:
                H3_0, s3_0 = net.last_H3.clone(), net.last_s3.clone()
            loss = net._loss(xs[t])
            loss.backward()
            losses.append(float(loss.detach()))
        grads = torch.cat([p.grad.flatten() for p in net.parameters() if p.requires_grad]).cpu()
        assert torch.isfinite(grads).all() and np.isfinite(losses).all()
        terms = net.last_terms.cpu()
        assert abs(float(terms.double().sum()) - losses[-1]) <= 1e-5 * abs(losses[-1])
        res.append((losses, grads))
        if rep == 0:
            sv = net._eng.sv
            _, rp = net._plan()
            wpH, wps, _, flow_up = rp.update_cache(H3_0, s3_0, sv["L2"], ofs, want_aux=True)
            assert torch.equal(wpH, net.last_H3_wp)                                      # the step's own warp, reproduced
            fu = flow_up.cpu()
            _, _, taps = ops.warp2(flow_up, H3_0.contiguous(), None, want_taps=True)
            assert torch.equal(taps.cpu(), oracle.warp_taps(fu, H, W)[0])
            assert torch.equal(wpH.cpu(), oracle.warp_tensor(fu, H3_0.cpu()))
            b = ops.resize_bilinear(sv["L2"], H // ofs, W // ofs, 255.0)
            q, _, lut = ops.equalize_prepare(b)
            got = torch.gather(lut.cpu().long(), 1, q.cpu().long()).view(1, 3, H // ofs, W // ofs).to(torch.uint8)
            assert torch.equal(got, oracle.equalize_u8(b.cpu().to(torch.uint8)))
            c = lambda k: sv[k].cpu()
            H3p, H4p = c("H3p"), c("H4p")
            outs = (c("Lp1"), c("Lp2"), c("L2"), c("s2"), c("s21"), c("s22"), c("H2"), c("H11"), c("H12"), H3p[:, :3], H3p[:, 3:],
                    H4p[:, :3], H4p[:, 3:], c("H3"), c("s3"), H3p, H4p, None, c("m_h"), c("H2b"), c("H3b"))
            with torch.no_grad():
                ref_total, ref_terms = oracle.loss_terms(xs[1].cpu(), outs, is_WB=True)
            assert abs(losses[1] - float(ref_total)) <= 1e-4 * abs(float(ref_total)), (losses[1], float(ref_total))
            del outs, sv
        del net
        torch.cuda.empty_cache()
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_bf16_training_trajectory_tracks_fp32(hip_ops, synth, oracle):
    """Throughput mode over several optimizer steps.  (a) free-running: the bf16 loss trajectory stays next to the fp32 parity
    mode's (same clip, same ClipAdam) and the enhanced output passes the PSNR gate.  (b) along the fp32 trajectory: a bf16 net
    given the fp32 run's weights / BN statistics / recurrent cache before every frame reproduces that frame's loss and output,
    i.e. the per-step bf16 error does not grow as the weights move away from their initial values.  (The free-running OUTPUTS do
    move apart -- 32-33 dB after 8 frames: the recurrent cache -> random-init RAFT -> warp loop amplifies any perturbation, the
    weights stay together; test_free_running_drift_control_vs_oracle measures that against the oracle, with the fp32 control.)"""
    ops, dev = hip_ops
    optim = importlib.import_module("zero-tig_amd.optim")
    H, W, steps = 256, 320, 8
    xs = [f.to(dev) for f in frames(synth, steps, H, W)]
    nets = {p: _net(ops, dev, synth, 1, 1, p) for p in ("fp32", "bf16")}
    opts = {p: optim.ClipAdam(nets[p]) for p in nets}
    follower = _net(ops, dev, synth, 1, 1, "bf16")
    fopt = optim.ClipAdam(follower)                     # only for its flat bucket (never steps)
    traj = {"fp32": [], "bf16": []}
    for t in range(steps):
        a = nets["fp32"]
        fopt.fp.flat.copy_(opts["fp32"].fp.flat)
        follower.load_state_dict({k: v for k, v in a.state_dict().items() if "running" in k or "num_batches" in k}, strict=False)
        if t > 0:
            follower.last_H3, follower.last_s3 = a.last_H3.clone(), a.last_s3.clone()
        for p in ("fp32", "bf16"):
            nets[p].is_new_seq = (t == 0)
            opts[p].zero_grad()
            loss = nets[p]._loss(xs[t])
            loss.backward()
            opts[p].step()
            traj[p].append(float(loss.detach()))
        follower.is_new_seq = (t == 0)
        with torch.no_grad():
            lf = float(follower._loss(xs[t]))
        assert abs(lf - traj["fp32"][t]) <= 1e-2 * abs(traj["fp32"][t]), (t, lf, traj["fp32"][t])
        ps = oracle.psnr_u8(follower.last_H3.cpu(), a.last_H3.cpu())
        print("frame %d: bf16 follower loss %.4f (fp32 %.4f), PSNR(bf16 H3, fp32 H3) %.2f dB" % (t, lf, traj["fp32"][t], ps))
        # new-sequence frame: nets only (> 45 dB, as test_bf16_mode_psnr_gate); steady-state frames also carry the bf16 RAFT's flow
        # error through the warp: 40 dB = 2.5/255 rms, and no drift over the steps
        assert ps > (45.0 if t == 0 else 40.0), (t, ps)
    for x, y in zip(traj["fp32"], traj["bf16"]):
        assert abs(x - y) <= 2e-2 * abs(x), (traj["fp32"], traj["bf16"])
    clean = torch.from_numpy(synth.clean_frame(steps - 1, H, W)).float()[None]
    assert abs(oracle.psnr_u8(nets["fp32"].last_H3.cpu(), clean) - oracle.psnr_u8(nets["bf16"].last_H3.cpu(), clean)) <= 0.01


def test_free_running_drift_control_vs_oracle(hip_ops, synth, oracle):
    """The control the drift explanation needs: the SAME free-running 8-step run for the CPU oracle
    (train.py:119-133 restated: OracleTrainer.step), the HIP fp32 parity mode and the HIP bf16 throughput mode -- each with its own
    weights, Adam state and recurrent cache.  Reported per frame: PSNR(H3) of each HIP mode against the oracle's H3 and the loss;
    after the run: for every pair, the fraction of the 92 620 bucket elements whose accumulated update (w_8 - w_0) has the opposite
    sign, and the relative L2 distance of the updates; plus two mixed runs (bf16 nets + fp32 RAFT, fp32 nets + bf16 RAFT) that say
    which half carries the drift, and every run's own flow_up against the oracle's.

    Measured (gpurun_out/r03c_drift.log): the WEIGHTS do not drift apart -- 0.06 % (fp32) / 0.46 % (bf16) of the
    elements end on the other side, update rel-L2 0.018 / 0.037 -- so round 2's "Adam sign flips" explanation was wrong.  What
    grows is the RECURRENT state: frame t's H3 feeds frame t+1's (randomly initialised, 12-iteration) RAFT, whose flow moves the
    warp, which moves H3.  That loop amplifies ANY perturbation: HIP-fp32 starts 96 dB from the oracle and loses 20 dB on the first
    RAFT frame, then ~2 dB per frame (63 dB after 8 frames; flow error 1e-4 -> 7e-4 px); bf16 starts at 57 dB (its one-frame error,
    cf. the teacher-forced gate) and follows the same law until it saturates near 33 dB (flow error 0.02 -> 0.06 px mean).  Either
    half alone in bf16 ends at the same 33-34 dB, i.e. no single bf16 tensor carries it.  Quality is unaffected: PSNR against the
    clean frame differs by < 0.01 dB after 8 steps, and the losses agree to < 0.5 %."""
    ops, dev = hip_ops
    optim = importlib.import_module("zero-tig_amd.optim")
    H, W, steps = 256, 320, 8
    xs = frames(synth, steps, H, W)
    tr = oracle.OracleTrainer(oracle.to_torch_state(synth.make_state(1)), of_scale=1)
    w0 = {n: tr.W[n].detach().clone() for n in tr.names}
    raft_mod = importlib.import_module("zero-tig_amd.raft")

    def mixed(net, raft_prec):
        """enhancement nets in one precision, the frozen RAFT plan in the other (test plumbing: swaps the bound plan -- AFTER the
        optimizer has re-homed the parameters into its flat bucket, which re-binds the plans)"""
        net._plan()
        rw = {"raft." + k: v.data for k, v in net.raft.state_dict().items()}
        net.__dict__["_raftplan"] = raft_mod.RaftPlan(ops, rw, dev, precision=raft_prec)
        assert net._plan()[1].h == (raft_prec == "bf16")
    nets = {"fp32": _net(ops, dev, synth, 1, 1, "fp32"), "bf16": _net(ops, dev, synth, 1, 1, "bf16"),
            "bf16nets+fp32raft": _net(ops, dev, synth, 1, 1, "bf16"), "fp32nets+bf16raft": _net(ops, dev, synth, 1, 1, "fp32")}
    opts = {p: optim.ClipAdam(nets[p]) for p in nets}
    mixed(nets["bf16nets+fp32raft"], "fp32")
    mixed(nets["fp32nets+bf16raft"], "bf16")
    rows = []
    for t in range(steps):
        lo, _, outs, aux, _, _ = tr.step(xs[t], t == 0)
        row = {"t": t, "loss_oracle": float(lo)}
        for p in nets:
            nets[p].is_new_seq = (t == 0)
            opts[p].zero_grad()
            if t > 0:           # the flow this run's own RAFT produces on this run's own cache / frame, against the oracle's
                prev = (nets[p].last_H3.clone(), nets[p].last_s3.clone())
            loss = nets[p]._loss(xs[t].to(dev))
            loss.backward()
            opts[p].step()
            row["loss_" + p] = float(loss.detach())
            row["psnr_" + p] = oracle.psnr_u8(nets[p].last_H3.cpu(), outs[13].detach())
            if t > 0:
                _, rp = nets[p]._plan()
                _, _, _, fu = rp.update_cache(prev[0], prev[1], nets[p]._eng.sv["L2"], 1, want_aux=True)
                d = (fu.cpu() - aux["flow_up"]).abs()
                row["flow_" + p] = (float(d.mean()), float(d.max()))
        row["psnr_bf16_vs_fp32"] = oracle.psnr_u8(nets["bf16"].last_H3.cpu(), nets["fp32"].last_H3.cpu())
        rows.append(row)
        print("frame %d: loss oracle %.4f hip-fp32 %.4f hip-bf16 %.4f | PSNR(H3) vs oracle: fp32 %.2f dB, bf16 %.2f dB, bf16 nets + fp32 RAFT "
              "%.2f dB, fp32 nets + bf16 RAFT %.2f dB | bf16 vs hip-fp32 %.2f dB"
              % (t, row["loss_oracle"], row["loss_fp32"], row["loss_bf16"], row["psnr_fp32"], row["psnr_bf16"],
                 row["psnr_bf16nets+fp32raft"], row["psnr_fp32nets+bf16raft"], row["psnr_bf16_vs_fp32"]))
        if t > 0:
            print("         flow_up vs oracle, mean / max abs (px): " + ", ".join("%s %.4f / %.3f" % ((p,) + row["flow_" + p]) for p in nets))
    upd = {"oracle": torch.cat([(tr.W[n].detach() - w0[n]).flatten() for n in tr.names]).double()}
    for p in ("fp32", "bf16"):
        fp = opts[p].fp
        cur = {n: fp.flat[o:o + s].detach().cpu() for n, o, s in zip(fp.names, fp.offsets, fp.sizes)}
        upd[p] = torch.cat([(cur[n] - w0[n].flatten()) for n in tr.names]).double()
    live = upd["oracle"].abs() > 0
    stats = {}
    for a, b in (("fp32", "oracle"), ("bf16", "oracle"), ("bf16", "fp32")):
        flip = float(((upd[a] * upd[b]) < 0)[live].double().mean())
        rel = float((upd[a] - upd[b]).norm() / upd[b].norm())
        # the same, weighted by how much gradient signal an element carries: elements whose oracle update is a full +-lr per step
        full = upd["oracle"].abs() > 0.5 * steps * 1e-4
        flip_full = float(((upd[a] * upd[b]) < 0)[full].double().mean()) if int(full.sum()) else 0.0
        stats[(a, b)] = (flip, rel, flip_full, float(full.double().mean()))
        print("8-step update %s vs %s: opposite sign on %.2f %% of the elements (%.2f %% among the %.1f %% that moved > 4 lr), rel-L2 %.3f"
              % (a, b, 100 * flip, 100 * flip_full, 100 * stats[(a, b)][3], rel))
    # gates: what the measurement showed (DESIGN section 2 quotes the table): the two fp32 implementations must track each other
    # at least as well as bf16 tracks either; nothing may diverge (loss within 2 %, PSNR floor).
    for r in rows:
        assert abs(r["loss_fp32"] - r["loss_oracle"]) <= 1e-3 * abs(r["loss_oracle"]), r
        assert abs(r["loss_bf16"] - r["loss_oracle"]) <= 1e-2 * abs(r["loss_oracle"]), r
        assert r["psnr_fp32"] > 60.0 and min(r["psnr_bf16"], r["psnr_bf16nets+fp32raft"], r["psnr_fp32nets+bf16raft"]) > 30.0, r
    assert rows[0]["psnr_fp32"] > 90.0 and rows[0]["psnr_bf16"] > 50.0, rows[0]
    assert stats[("fp32", "oracle")][0] < 0.005 and stats[("bf16", "oracle")][0] < 0.02          # the weights stay together
    assert stats[("fp32", "oracle")][1] < 0.05 and stats[("bf16", "oracle")][1] < 0.1
    clean = torch.from_numpy(synth.clean_frame(steps - 1, H, W)).float()[None]
    po = oracle.psnr_u8(tr.cache["last_H3"], clean)
    for p in ("fp32", "bf16"):
        d = abs(oracle.psnr_u8(nets[p].last_H3.cpu(), clean) - po)
        print("after %d steps: |PSNR(H3_%s, clean) - PSNR(H3_oracle, clean)| = %.4f dB" % (steps, p, d))
        assert d <= 0.01, (p, d)


def test_scripts_train_resume_predict_pipeline(tmp_path, synth):
    """the reference's scripts end to end on a tiny BVI-RLV-layout clip (the loader resizes to 1920 x 1080): train.py
    (hipGraph stepper, per-epoch weights + resume file + result PNGs), resume, predict.py, and run_pipeline.py -> evals.py
    (device PSNR, Metrics.json)."""
    import json
    import os
    import subprocess
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = tmp_path / "data" / "RLV"
    for kind, sub, fn in (("input", "low_light_10", synth.lowlight_frame), ("gt", "normal_light_10", synth.clean_frame)):
        d = data / kind / "S01" / sub
        d.mkdir(parents=True)
        for t in range(4):
            a = np.asarray(fn(t, 270, 480), dtype=np.float32)
            im = (np.transpose(a[0] if a.ndim == 4 else a, (1, 2, 0)) * 255.0 + 0.5).astype(np.uint8)
            Image.fromarray(im).save(str(d / ("%05d.png" % (t + 1))))
    (data / "train_list.txt").write_text("S01\n")
    (data / "test_list.txt").write_text("S01\n")
    env = dict(os.environ, PYTHONPATH=root)

    def run(*cmd):
        r = subprocess.run([sys.executable] + list(cmd), cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (cmd, r.stdout[-3000:], r.stderr[-3000:])
        return r.stdout
    exp = tmp_path / "exp"
    out = run("run_pipeline.py", "--datasets", "RLV", "--base_data_dir", str(tmp_path / "data"), "--base_exp_dir", str(exp), "--epochs", "2",
              "--weights_dir", str(tmp_path / "none"))
    tr = [d for d in (exp / "RLV" / "training").iterdir() if d.name.startswith("Train-")][0]
    for f in ("initial_weights.pt", "model_epochs/weights_0.pt", "model_epochs/weights_1.pt", "model_epochs/resume.pt", "log.txt"):
        assert (tr / f).exists(), f
    assert len(list((tr / "result" / "denoise").glob("*.png"))) == 8          # 4 test frames x 2 epochs
    m = json.load(open(exp / "RLV" / "evaluation" / "Metrics.json"))
    assert m["images"] == 4 and 3.0 < m["Total_PSNR"] < 60.0
    ck = torch.load(str(tr / "model_epochs" / "resume.pt"))
    assert ck["epoch"] == 2 and ck["step"] == 8 and len(ck["model"]) == 223 and ck["optimizer"]["t"] == 8
    out = run("train.py", "--lowlight_images_path", str(data), "--save", str(tmp_path / "exp2"), "--epochs", "3", "--resume",
              str(tr / "model_epochs" / "resume.pt"))
    assert "resumed from" in out and "train-epoch 002" in out and "train-epoch 000 " not in out
    run("predict.py", "--lowlight_images_path", str(data), "--save", str(tmp_path / "pred"), "--model_pretrain", str(tr / "model_epochs" / "weights_1.pt"))
    assert len(list((tmp_path / "pred").rglob("*_denoise.png"))) == 4 and len(list((tmp_path / "pred").rglob("*_enhance.png"))) == 4


def test_bench_one_rank_through_rccl():
    """Multi-GPU readiness on the one-GPU box: bench.py started the way torch.distributed.run starts a rank (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in the environment) with ONE rank: init_process_group("nccl"), the barriers of the timed region and the
    flat-bucket all-reduce of optim.ClipAdam.step all go through RCCL (the same calls the N-rank job makes)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29600 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--height", "540",
                        "--width", "960", "--cpu-baseline", "none"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "nccl" in line["config"]["collective"], line["config"]
    assert line["roofline"] is not None and np.isfinite(line["final_loss"])


def test_train_one_epoch_did_sdsd_underwater_layouts(tmp_path, synth):
    """SURVEY 8(f)-1 on the GPU: one `train.py` epoch per remaining loader layout (DID list + folders with jpg / png, SDSD pair
    directories, the tree walk used for `--dataset underwater`), frames of another size than 1920 x 1080, so that decode workers,
    the uint8 H2D copy and the device-side PIL-exact resize + ToTensor all run inside the real training loop."""
    import os
    import subprocess
    import sys
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def frame(t, hw=(180, 320)):
        a = np.asarray(synth.lowlight_frame(t, *hw), dtype=np.float32)[0]
        return Image.fromarray((np.transpose(a, (1, 2, 0)) * 255.0 + 0.5).astype(np.uint8))

    def put(path, t):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        frame(t).save(path, quality=95)
    did, sd, uw = str(tmp_path / "did"), str(tmp_path / "sdsd"), str(tmp_path / "uw")
    for t in range(3):
        put(os.path.join(did, "input", "V1", "%d.jpg" % (t + 1)), t)
        put(os.path.join(uw, "clipA", "%d.png" % (t + 11)), t)
    put(os.path.join(did, "input", "V2", "7.png"), 5)
    for lst in ("train_list.txt", "test_list.txt"):
        open(os.path.join(did, lst), "w").write("V1\nV2\n")
    for pair, stem in (("pair5", 5), ("pair2", 2), ("pair3", 3)):
        put(os.path.join(sd, "indoor", "indoor_png", pair, "%d.png" % stem), stem)
        put(os.path.join(sd, "indoor", "indoor_png", pair, "%d_gt.png" % stem), 9)
    for ph in ("train", "test"):
        open(os.path.join(sd, "sdsd_in_%s.txt" % ph), "w").write("pair5\npair2\npair3\n")
    env = dict(os.environ, PYTHONPATH=root)
    for name, path, nfr in (("DID", did, 4), ("SDSD", sd, 3), ("underwater", uw, 3)):
        r = subprocess.run([sys.executable, "train.py", "--lowlight_images_path", path, "--dataset", name, "--epochs", "1", "--num_workers", "2",
                            "--save", str(tmp_path / ("exp_" + name))], cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (name, r.stdout[-3000:], r.stderr[-3000:])
        assert "train-epoch 000 %03d " % (nfr - 1) in r.stdout and "train-epoch 000 %03d " % nfr not in r.stdout, (name, r.stdout[-2000:])
        assert "device resize + ToTensor" in r.stdout, (name, r.stdout[-2000:])
        losses = [float(l.split()[-1]) for l in r.stdout.splitlines() if "train-epoch 000 0" in l]
        assert len(losses) == nfr and all(np.isfinite(losses)), (name, losses)
