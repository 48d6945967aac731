"""Op-level parity: every C-ABI kernel against the CPU oracle on identical inputs.
Back-end "emu" runs the same .hip sources through tests/hipemu on the CPU; back-end "hip" (marked gpu) is the
real gfx950 library and is the parity gate."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def maxerr(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


def test_warp_golden_and_taps(backend, oracle):
    ops, dev, _ = backend
    g = load_golden("g6_ops")
    flow, img = T(g["warp_flow"], dev), T(g["warp_img"], dev)
    a, b, taps = ops.warp2(flow, img, img * 0.5, want_taps=True)
    # integer contract: tap indices bit-exact with ATen's arithmetic
    ref_taps = oracle.warp_taps(torch.from_numpy(g["warp_flow"]), img.shape[-2], img.shape[-1])[0]
    assert torch.equal(taps.cpu(), ref_taps)
    assert np.array_equal(a.cpu().numpy(), g["warp_out"])            # same rounding sequence -> bit-exact values too
    assert maxerr(b, torch.from_numpy(g["warp_out"]) * 0.5) < 1e-6
    # identity-size flow (of_scale=1) and padded flow (132x164 frame, 136x168 flow) from the sequence fixtures
    for name in ("g3_seq_128x160", "g4_seq_132x164"):
        s = load_golden(name)
        fl = T(s["flow_up"], dev)
        H, W = int(s["meta"][0]), int(s["meta"][1])
        img = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(1))
        out, _, taps = ops.warp2(fl, img.to(dev), None, want_taps=True)
        assert np.array_equal(taps.cpu().numpy()[..., 0], s["warp_x0"][0, 0]) and np.array_equal(taps.cpu().numpy()[..., 1], s["warp_y0"][0, 0])
        assert torch.equal(out.cpu(), oracle.warp_tensor(torch.from_numpy(s["flow_up"]), img))


def _adj_ref(fn, x, g):
    """reference adjoint via autograd on the oracle op: d/dx <fn(x), g>"""
    x = x.clone().requires_grad_(True)
    (fn(x) * g).sum().backward()
    return x.grad


def test_stencils(backend, oracle):
    ops, dev, _ = backend
    g = load_golden("g6_ops")
    x, y = T(g["x"], dev), T(g["y"], dev)
    xc, yc = x.cpu(), y.cpu()
    a, b = ops.pair_down(x)
    assert np.array_equal(a.cpu().numpy(), g["pd1"]) and np.array_equal(b.cpu().numpy(), g["pd2"])
    g1, g2 = torch.rand(1, 3, 20, 28), torch.rand(1, 3, 20, 28)
    ref = _adj_ref(lambda t: oracle.pair_downsample(t)[0], xc, g1) + _adj_ref(lambda t: oracle.pair_downsample(t)[1], xc, g2)
    assert maxerr(ops.pair_down_adj(g1.to(dev), g2.to(dev), 40, 56), ref) < 1e-7
    odd = torch.rand(1, 3, 41, 57)
    oa, ob = ops.pair_down(odd.to(dev))
    ra, rb = oracle.pair_downsample(odd)
    assert torch.equal(oa.cpu(), ra) and torch.equal(ob.cpu(), rb)
    assert maxerr(ops.pair_down_adj(g1.to(dev), g2.to(dev), 41, 57),
                  _adj_ref(lambda t: oracle.pair_downsample(t)[0], odd, g1) + _adj_ref(lambda t: oracle.pair_downsample(t)[1], odd, g2)) < 1e-7
    # blur + adjoint
    assert maxerr(ops.blur21(x), torch.from_numpy(g["blur"])) < 1e-6
    assert maxerr(ops.gauss_taps(), oracle.gauss_taps_1d()) < 5e-8   # fp64 erf on the host vs the reference fp32 erf
    gb = torch.rand(1, 3, 40, 56)
    assert maxerr(ops.blur21_adj(gb.to(dev)), _adj_ref(oracle.blur21, xc, gb)) < 2e-6
    # local mean + adjoint
    assert maxerr(ops.box5_reflect(x), torch.from_numpy(g["localmean"])) < 1e-6
    assert maxerr(ops.box5_reflect_adj(gb.to(dev), 1.0), _adj_ref(oracle.local_mean_reflect, xc, gb)) < 1e-6
    # local variance fwd / bwd
    D, V = ops.localvar_fwd(x)
    assert maxerr(V, torch.from_numpy(g["localvar"])) < 1e-6
    D2, V2 = ops.localvar_fwd(x, y)
    assert maxerr(V2, oracle.local_variance_zero(xc - yc)) < 1e-6
    assert maxerr(ops.localvar_bwd(D, gb.to(dev)), _adj_ref(oracle.local_variance_zero, xc, gb)) < 1e-6
    assert maxerr(ops.localvar_bwd(D2, gb.to(dev), -1.0), -_adj_ref(oracle.local_variance_zero, xc - yc, gb)) < 1e-6
    # texture mask
    m, r = ops.texture_mask(T(g["tex_in1"], dev), T(g["tex_in2"], dev), want_ratio=True)
    assert maxerr(r, torch.from_numpy(g["texratio"])) < 2e-5
    assert (m.cpu().numpy() != g["texmask"]).mean() <= 1e-3
    # flat "YCbCr"
    assert maxerr(ops.ycc_flat(x * 0.2), torch.from_numpy(g["ycc"])) < 1e-6


def _nhwc(t, ld=None):
    """NCHW cpu tensor -> NHWC (optionally padded to ld channels)"""
    n = t.permute(0, 2, 3, 1).contiguous()
    if ld is not None and ld != n.shape[-1]:
        n = torch.cat([n, torch.zeros(*n.shape[:3], ld - n.shape[-1])], -1).contiguous()
    return n


CONV_CASES = [
    # Cin, Cout, KH, KW, stride, pad, H, W, act
    (3, 48, 3, 3, 1, (1, 1), 9, 37, "lrelu"),       # thin-in first layer (Denoise_1.conv1), ragged tile edges
    (48, 48, 3, 3, 1, (1, 1), 8, 33, "lrelu"),      # Denoise conv2 (NT=3)
    (64, 64, 3, 3, 1, (1, 1), 6, 40, "relu"),       # Enhancer body (NT=4)
    (64, 3, 3, 3, 1, (1, 1), 7, 20, "sigmoid_clamp"),   # thin-out (NT=1)
    (48, 6, 1, 1, 1, (0, 0), 5, 19, None),
    (3, 64, 7, 7, 2, (3, 3), 20, 36, None),         # RAFT encoder stem
    (64, 96, 3, 3, 2, (1, 1), 11, 21, None),        # stride-2 residual conv
    (64, 96, 1, 1, 2, (0, 0), 11, 21, None),        # stride-2 downsample
    (40, 32, 1, 5, 1, (0, 2), 6, 18, "sigmoid"),    # SepConvGRU horizontal (reduced channels)
    (40, 32, 5, 1, 1, (2, 0), 6, 18, "tanh"),       # SepConvGRU vertical
    (2, 128, 7, 7, 1, (3, 3), 6, 10, "relu"),       # convf1
    (20, 126, 3, 3, 1, (1, 1), 5, 9, "relu"),       # ragged Cin chunk + ragged Cout
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "c%d-%d_k%dx%d_s%d" % c[:5])
def test_conv_fwd(backend, case):
    import torch.nn.functional as F
    ops, dev, name = backend
    Cin, Cout, KH, KW, stride, pad, H, W, act = case
    g = torch.Generator().manual_seed(Cin * 131 + Cout)
    N = 2 if Cin == 3 and KH == 7 else 1
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    ref = {None: lambda t: t, "relu": torch.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2), "sigmoid": torch.sigmoid,
           "tanh": torch.tanh, "sigmoid_clamp": lambda t: torch.clamp(torch.sigmoid(t), 1e-4, 1)}[act](ref)
    ld = (Cin + 3) // 4 * 4
    xd = _nhwc(x, ld).to(dev)
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    wd = ops.repack_weight(w.to(dev))
    y = ops.conv2d(CV(xd, 0, Cin), wd, b.to(dev), Cout, KH, KW, stride, pad, act)
    got = y.cpu()[..., :Cout].permute(0, 3, 1, 2)
    assert maxerr(got, ref) < 2e-5, maxerr(got, ref)
    if Cout <= 6:   # planar epilogue used by the thin output layers
        yp = ops.conv2d(CV(xd, 0, Cin), wd, b.to(dev), Cout, KH, KW, stride, pad, act, out_planar=True)
        assert maxerr(yp, ref) < 2e-5


def test_conv_split_input_alpha_epi(backend):
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(5)
    H, W = 5, 21
    xa, xb = torch.randn(1, 16, H, W, generator=g), torch.randn(1, 24, H, W, generator=g)
    w = torch.randn(32, 40, 3, 3, generator=g) * 0.05
    aux = torch.randn(1, 32, H, W, generator=g)
    big = torch.zeros(1, H, W, 64)
    big[..., 8:32] = _nhwc(xb)
    ref = 0.25 * F.conv2d(torch.cat([xa, xb], 1), w, None, padding=1)
    wd = ops.repack_weight(w.to(dev))
    auxd = _nhwc(aux).to(dev)
    for epi, fn in ((0, lambda r: r), (1, lambda r: r * torch.where(aux > 0, 1.0, 0.2)), (2, lambda r: r * (aux > 0)), (3, lambda r: r + aux)):
        out = torch.zeros(1, H, W, 48, device=dev)
        ops.conv2d(CV(_nhwc(xa).to(dev)), wd, None, 32, 3, 3, 1, (1, 1), None, alpha=0.25, x2=CV(big.to(dev), 8, 24),
                   out=CV(out, 16, 32), aux=auxd, epi=epi)
        assert maxerr(out.cpu()[..., 16:48].permute(0, 3, 1, 2), fn(ref)) < 1e-5
        assert float(out[..., :16].abs().max()) == 0.0
    # data-gradient operator == conv with the transposed / flipped weights
    xg = torch.randn(1, 40, H, W, generator=g, requires_grad=True)
    dz = torch.randn(1, 32, H, W, generator=g)
    (F.conv2d(xg, w, None, padding=1) * dz).sum().backward()
    wdg = ops.repack_weight(w.to(dev), transpose_flip=True)
    dx = ops.conv2d(CV(_nhwc(dz).to(dev)), wdg, None, 40, 3, 3, 1, (1, 1))
    assert maxerr(dx.cpu().permute(0, 3, 1, 2), xg.grad) < 1e-5


WGRAD_CASES = [(3, 48, 3, 4), (48, 48, 3, 48), (48, 3, 1, 48), (9, 64, 3, 12), (64, 64, 3, 64), (64, 3, 3, 64), (12, 48, 3, 12), (48, 6, 1, 48)]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=lambda c: "c%d-%d_k%d" % c[:3])
def test_conv_wgrad(backend, case):
    import torch.nn.functional as F
    ops, dev, _ = backend
    Cin, Cout, K, ldx = case
    g = torch.Generator().manual_seed(Cin + 7 * Cout)
    H, W = 9, 21
    x = torch.randn(1, Cin, H, W, generator=g)
    dz = torch.randn(1, Cout, H, W, generator=g)
    w = torch.zeros(Cout, Cin, K, K, requires_grad=True)
    (F.conv2d(x, w, None, padding=K // 2) * dz).sum().backward()
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    xd = CV(_nhwc(x, ldx).to(dev), 0, Cin)
    dzd = CV(_nhwc(dz, (Cout + 3) // 4 * 4).to(dev), 0, Cout)
    gw = torch.full((Cout, Cin, K, K), 7.0, device=dev)
    gb = torch.full((Cout,), 5.0, device=dev)
    ops.conv2d_wgrad(xd, dzd, Cout, K, K, gw, accumulate=False, grad_b=gb)
    assert maxerr(gw, w.grad) < 5e-5, maxerr(gw, w.grad)
    assert maxerr(gb, dz.sum(dim=(0, 2, 3))) < 5e-5
    ops.conv2d_wgrad(xd, dzd, Cout, K, K, gw, accumulate=True, grad_b=gb)
    assert maxerr(gw, 2 * w.grad) < 1e-4 and maxerr(gb, 2 * dz.sum(dim=(0, 2, 3))) < 1e-4


def test_norms(backend):
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(11)
    # instance norm on a batch of 2 (RAFT fnet) with relu, then residual form
    x = torch.randn(2, 96, 9, 14, generator=g) * 2 + 0.7
    res = torch.randn(2, 96, 9, 14, generator=g)
    xd, rd = _nhwc(x).to(dev), _nhwc(res).to(dev)
    part = ops.chan_stats(xd, nblk=3)
    sc, sh, _, _ = ops.norm_finalize(part, 2, 96, 9 * 14, 0)
    y = ops.norm_apply(xd, sc, sh, inner_relu=True)
    assert maxerr(y.cpu().permute(0, 3, 1, 2), F.relu(F.instance_norm(x))) < 2e-5
    y = ops.norm_apply(xd, sc, sh, res=rd, inner_relu=True, outer_relu=True)
    assert maxerr(y.cpu().permute(0, 3, 1, 2), F.relu(res + F.relu(F.instance_norm(x)))) < 2e-5
    # eval BN
    C = 64
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    xb = torch.randn(1, C, 10, 13, generator=g)
    sc, sh, _, _ = ops.norm_finalize(None, 1, C, 1, 2, gam.to(dev), bet.to(dev), rm.to(dev), rv.to(dev), dev=dev)
    y = ops.norm_apply(_nhwc(xb).to(dev), sc, sh)
    assert maxerr(y.cpu().permute(0, 3, 1, 2), F.batch_norm(xb, rm, rv, gam, bet, False, 0.1, 1e-5)) < 1e-5
    # train BN + ReLU + residual: forward, running stats, backward (Enhancer block)
    f = torch.randn(1, C, 10, 13, generator=g)
    z = (torch.randn(1, C, 10, 13, generator=g) * 1.7 + 0.3).requires_grad_(True)
    gam_t, bet_t = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rm_t, rv_t = rm.clone(), rv.clone()
    out_ref = f + F.relu(F.batch_norm(z, rm_t, rv_t, gam_t, bet_t, True, 0.1, 1e-5))
    dy = torch.randn(1, C, 10, 13, generator=g)
    (out_ref * dy).sum().backward()
    rmd, rvd = rm.to(dev), rv.to(dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    zd = _nhwc(z.detach()).to(dev)
    part = ops.chan_stats(zd)
    sc, sh, mu, rs = ops.norm_finalize(part, 1, C, 130, 1, gam.to(dev), bet.to(dev), rmd, rvd, nbt)
    y = ops.norm_apply(zd, sc, sh, res=_nhwc(f).to(dev), inner_relu=True)
    assert maxerr(y.cpu().permute(0, 3, 1, 2), out_ref) < 1e-5
    assert maxerr(rmd, rm_t) < 1e-6 and maxerr(rvd, rv_t) < 1e-6 and int(nbt) == 1
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dz = ops.bn_relu_bwd(_nhwc(dy).to(dev), zd, sc, sh, mu, rs, dg, db)
    assert maxerr(dz.cpu().permute(0, 3, 1, 2), z.grad) < 2e-5
    assert maxerr(dg, gam_t.grad) < 2e-4 and maxerr(db, bet_t.grad) < 2e-4


@pytest.mark.parametrize("case", [(2, 96, 9, 14, "f32"), (2, 64, 70, 66, "bf16"), (2, 96, 23, 40, "bf16"), (2, 96, 72, 64, "bf16"), (1, 128, 45, 80, "f32")],
                         ids=lambda c: "n%d_c%d_%dx%d_%s" % c)
def test_instance_norm_one_launch(backend, case):
    """InstanceNorm scale / shift from the statistics kernel's last workgroup (RAFT feature encoder, extractor.py:117-191) ==
    the two-launch chan_stats + norm_finalize path, on every statistics kernel (fp32, bf16 16-byte fast path, bf16 generic);
    called repeatedly so the ticket counters are seen to return to zero."""
    import torch.nn.functional as F
    ops, dev, _ = backend
    N, C, H, W, kind = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.4
    if kind == "bf16":
        x = x.bfloat16().float()
    xd = _nhwc(x).to(dev)
    if kind == "bf16":
        xd = xd.bfloat16()
    part = ops.chan_stats(xd)
    sc0, sh0, _, _ = ops.norm_finalize(part, N, C, H * W, 0)
    for _ in range(3):
        sc, sh = ops.instance_norm_stats(xd)
        assert maxerr(sc, sc0) <= 1e-6 * float(sc0.abs().max()) and maxerr(sh, sh0) <= 1e-6 * max(1.0, float(sh0.abs().max()))
    y = ops.norm_apply(xd, sc, sh, inner_relu=True)
    tol = 2e-5 if kind == "f32" else 2e-2
    assert maxerr(y.float().cpu().permute(0, 3, 1, 2), F.relu(F.instance_norm(x))) < tol
    assert int(ops._tickets[xd.device][0].abs().sum()) == 0


def test_norms_bf16_fast_path(backend):
    """Enhancer block in bf16 storage at a size that takes the 16-byte-per-lane kernels (HW >= 4096, C % 8 == 0): train BN +
    ReLU + residual forward and backward against torch on the same bf16-rounded inputs; ragged pixel count (not a multiple of 4)."""
    import torch.nn.functional as F
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(23)
    C, H, W = 64, 67, 63
    bf = lambda t: t.bfloat16().float()
    f = bf(torch.randn(1, C, H, W, generator=g))
    z0 = bf(torch.randn(1, C, H, W, generator=g) * 1.7 + 0.3)
    z = z0.clone().requires_grad_(True)
    gam, bet = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    gam_t, bet_t = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    out_ref = f + F.relu(F.batch_norm(z, rm.clone(), rv.clone(), gam_t, bet_t, True, 0.1, 1e-5))
    dy = bf(torch.randn(1, C, H, W, generator=g))
    (out_ref * dy).sum().backward()
    zd, fd, dyd = (_nhwc(t).bfloat16().to(dev) for t in (z0, f, dy))
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    part = ops.chan_stats(zd)
    sc, sh, mu, rs = ops.norm_finalize(part, 1, C, H * W, 1, gam.to(dev), bet.to(dev), rm.to(dev), rv.to(dev), nbt)
    y = ops.norm_apply(zd, sc, sh, res=fd, inner_relu=True)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert float(((got - out_ref.detach()).abs() - out_ref.detach().abs() * 2 ** -8).max()) < 1e-3
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dz = ops.bn_relu_bwd(dyd, zd, sc, sh, mu, rs, dg, db)
    gz = dz.float().cpu().permute(0, 3, 1, 2)
    assert float(((gz - z.grad).abs() - z.grad.abs() * 2 ** -8).max()) < 1e-3
    assert maxerr(dg, gam_t.grad) < 2e-3 * float(gam_t.grad.abs().max()) and maxerr(db, bet_t.grad) < 2e-3 * float(bet_t.grad.abs().max())


def test_raft_ops_golden(backend, oracle, synth):
    """corr volume (as MFMA 1x1 conv) + pyramid + fused lookup, equalize, one update-block step + convex upsample (G6)."""
    from importlib import import_module
    ops, dev, _ = backend
    CV = import_module("zero-tig_amd.ops").CV
    g = load_golden("g6_ops")
    # equalize
    e_in = torch.from_numpy(g["eq_in"])
    q, hist, lut = ops.equalize_prepare(e_in.float().to(dev))
    assert torch.equal(q.cpu().view(3, 30, 44), e_in[0])
    out = torch.gather(lut.cpu().long(), 1, q.cpu().long()).view(1, 3, 30, 44).to(torch.uint8)
    assert np.array_equal(out.numpy(), g["eq_out"])
    const = torch.full((1, 3, 8, 8), 77.0)
    _, _, lut2 = ops.equalize_prepare(const.to(dev))
    assert torch.equal(lut2.cpu(), torch.arange(256, dtype=torch.int32).repeat(3, 1))        # step == 0 -> identity
    # correlation volume / pyramid / lookup
    h, w = 16, 24
    f1 = torch.from_numpy(synth.normal("ops.f1", (1, 256, h, w), 0.0, 1.0, 7))
    f2 = torch.from_numpy(synth.normal("ops.f2", (1, 256, h, w), 0.0, 1.0, 7))
    corr0 = ops.conv2d(CV(_nhwc(f1).to(dev)), f2.view(1, 256, h * w).to(dev).contiguous(), None, h * w, 1, 1, alpha=1.0 / 16.0)
    levels = ops.corr_pyramid(corr0, h, w)
    assert maxerr(corr0.cpu().view(h * w, h, w)[::7], torch.from_numpy(g["corr_pyr0"])[:, 0]) < 2e-5
    for i, lv in enumerate(levels):
        assert maxerr(lv, torch.from_numpy(g["corr_pyr%d" % (i + 1)])[:, 0]) < 2e-5
    coords = torch.from_numpy(g["lookup_coords"])
    cflat = coords[0].permute(1, 2, 0).reshape(-1, 2).contiguous().to(dev)
    look = ops.corr_lookup(corr0, levels, h, w, cflat)
    assert maxerr(look.cpu().permute(0, 3, 1, 2), torch.from_numpy(g["lookup_out"])) < 2e-5
    # lookup on the ORACLE's pyramid must reproduce ATen's taps/weights bit for bit (integer contract of corr.py:29-50)
    pyr = oracle.corr_pyramid(f1, f2)
    c0 = torch.zeros(1, h, w, h * w)
    c0[0] = pyr[0].view(h, w, h * w)
    look2 = ops.corr_lookup(c0.to(dev), [p[:, 0].contiguous().to(dev) for p in pyr[1:]], h, w, cflat)
    assert torch.equal(look2.cpu().permute(0, 3, 1, 2), oracle.corr_lookup(pyr, coords))


@pytest.mark.parametrize("Cdr,hw", [(3, (9, 41)), (6, (27, 101)), (6, (64, 96))], ids=["c3-9x41", "c6-27x101", "c6-64x96"])
def test_thin1x1_bwd_fused(backend, Cdr, hw):
    """Denoise_1/2 conv3 backward in one pass (zt_thin1x1_bwd_bf16): the data gradient must be bit-identical to the 1x1 thin
    data-gradient kernel's, the weight / bias gradients must match autograd (and the separate weight-gradient kernel)."""
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    H, W = hw
    g = torch.Generator().manual_seed(Cdr * 100 + H)
    a2 = torch.randn(1, 48, H, W, generator=g).bfloat16().float()
    dr = torch.randn(1, Cdr, H, W, generator=g).bfloat16().float()
    w3 = torch.randn(Cdr, 48, 1, 1, generator=g) * 0.2
    wT = ops.repack_weight_bf16(w3.to(dev), transpose_flip=True)
    a2d, drd = _nhwc_bf16(a2, 48).to(dev), _nhwc_bf16(dr, 8).to(dev)
    per = ops.wgrad_slab_floats(48, Cdr, 1)
    slab = torch.full((600 * per,), float("nan"), device=dev)
    dz, n = ops.thin1x1_bwd_bf16(drd, Cdr, wT, a2d, slab, 0)
    ref_dz = ops.conv2d_bf16(CV(drd, 0, Cdr), wT, None, 48, 1, 1, (0, 0), None, aux=a2d, epi=1)
    assert torch.equal(dz.cpu(), ref_dz.cpu()[..., :48])
    gw, gb = torch.zeros(Cdr, 48, 1, 1, device=dev), torch.zeros(Cdr, device=dev)
    ops.wgrad_reduce_multi([(slab, n, 48, Cdr, 1, gw, gb)], accumulate=False)
    ref_w = torch.einsum("nohw,nihw->oi", dr, a2)
    assert maxerr(gw.view(Cdr, 48), ref_w) < 2e-5 * float(ref_w.abs().max()) + 1e-4, maxerr(gw.view(Cdr, 48), ref_w)
    assert maxerr(gb, dr.sum(dim=(0, 2, 3))) < 1e-4 * (H * W) ** 0.5
    g2, b2 = torch.zeros_like(gw), torch.zeros_like(gb)
    ops.conv2d_wgrad_bf16(CV(a2d, 0, 48), CV(drd, 0, Cdr), Cdr, 1, 1, g2, grad_b=b2)
    assert maxerr(gw, g2) < 2e-5 * float(ref_w.abs().max()) + 1e-4


@pytest.mark.parametrize("pair", [(324, 256, 1, 2, 128, 7), (256, 192, 3, 128, 64, 3), (96, 64, 3, 40, 32, 3)], ids=["1x1+7x7", "3x3+3x3", "fallback"])
def test_conv_pair_equals_two_launches(backend, pair):
    """zt_conv2d_pair_nhwc_bf16 (RAFT motion encoder: convc1 || convf1, convc2 || convf2 in one launch each) against the same two
    convolutions launched separately: the same kernel body runs, so the outputs must be bit-identical (third case: a shape the
    pair kernels do not cover takes the two-launch fallback)."""
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    CinA, CoutA, KA, CinB, CoutB, KB = pair
    H, W = 9, 40
    g = torch.Generator().manual_seed(CinA + KB)

    def mk(Cin, Cout, K):
        ld = (Cin + 7) // 8 * 8
        x = torch.zeros(1, H, W, ld)
        x[..., :Cin] = torch.randn(1, H, W, Cin, generator=g)
        w = torch.randn(Cout, Cin, K, K, generator=g) * (Cin * K * K) ** -0.5
        b = torch.randn(Cout, generator=g)
        return CV(x.bfloat16().to(dev), 0, Cin), ops.repack_weight_bf16(w.to(dev)), b.to(dev)
    xA, wA, bA = mk(CinA, CoutA, KA)
    xB, wB, bB = mk(CinB, CoutB, KB)
    oA = torch.zeros(1, H, W, CoutA, dtype=torch.bfloat16, device=dev)
    oB = torch.zeros(1, H, W, CoutB + 8, dtype=torch.bfloat16, device=dev)             # a channel slice of a wider buffer
    ops.conv_pair_bf16(xA, wA, bA, CoutA, KA, oA, xB, wB, bB, CoutB, KB, CV(oB, 8, CoutB), act="relu")
    rA = ops.conv2d_bf16(xA, wA, bA, CoutA, KA, KA, (KA // 2, KA // 2), "relu")
    rB = ops.conv2d_bf16(xB, wB, bB, CoutB, KB, KB, (KB // 2, KB // 2), "relu")
    assert torch.equal(oA.cpu(), rA.cpu()[..., :CoutA]) and torch.equal(oB.cpu()[..., 8:], rB.cpu()[..., :CoutB])
    assert float(oB.float().abs().max()) > 0 and float(oB[..., :8].float().abs().max()) == 0


CORR_SIZES = [pytest.param(18, 40, id="18x40"), pytest.param(16, 32, id="16x32"), pytest.param(45, 80, id="45x80", marks=pytest.mark.gpu),
              pytest.param(23, 40, id="23x40", marks=pytest.mark.gpu), pytest.param(90, 160, id="90x160", marks=pytest.mark.gpu)]


@pytest.mark.parametrize("h,w", CORR_SIZES)
def test_corr_volume_pyramid_fused_bf16(backend, h, w):
    """corr.py:13-27, 52-60 in one launch (zt_corr.hip): all-pairs volume of two bf16 feature maps / sqrt(256) and the three
    avg_pool2d(2, 2) levels, against torch on the same bf16-rounded inputs; ragged bands / column tiles / source tiles, floor sizes
    of the pooled levels (45 -> 22 -> 11 -> 5)."""
    import torch.nn.functional as F
    ops, dev, name = backend
    if name == "emu" and h * w > 1000:
        pytest.skip("minutes on the emulator")
    g = torch.Generator().manual_seed(h * 1000 + w)
    npx = h * w
    f1 = torch.randn(npx, 256, generator=g).bfloat16()
    f2 = torch.randn((npx + 15) // 16 * 16, 256, generator=g).bfloat16()
    f2[npx:] = 0
    corr0, levels = ops.corr_volume_pyramid_bf16(f1.to(dev), f2.to(dev), h, w, 1.0 / 16.0)
    ref = (f1.float() @ f2[:npx].float().t()) / 16.0                               # [npx1][npx2]
    assert maxerr(corr0.view(npx, -1)[:, :npx], ref) < 2e-5 * float(ref.abs().max())
    lvl = ref.view(npx, 1, h, w)
    for got in levels:
        lvl = F.avg_pool2d(lvl, 2, stride=2)
        assert tuple(got.shape) == (npx, lvl.shape[2], lvl.shape[3])
        assert maxerr(got, lvl[:, 0]) < 2e-5 * float(ref.abs().max())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_raft_update_step_golden(backend, oracle, synth, precision):
    """One BasicUpdateBlock step (update.py:114-136) through the plan's HIP kernels -- fused 4-level lookup, motion encoder,
    SepConvGRU x2, flow head, mask head, convex up-sampling -- against the reference's own `ub_net` / `ub_dflow` / `ub_mask` /
    `ub_up` (golden g6: `net.raft.update_block(h, inp, corr, flow)` and `upsample_flow`)."""
    import importlib
    ops, dev, _ = backend
    CV = importlib.import_module("zero-tig_amd.ops").CV
    g = load_golden("g6_ops")
    raft_mod = importlib.import_module("zero-tig_amd.raft")
    st = synth.make_state(3)
    W = {k: torch.from_numpy(np.array(v)).to(dev) for k, v in st.items() if k.startswith("raft.")}
    plan = raft_mod.RaftPlan(ops, W, dev, precision=precision)
    bf = precision == "bf16"      # throughput mode: bf16 storage of activations / weights (3 significant digits), fp32 accumulation
    h, w = 16, 24
    hh = torch.tanh(torch.from_numpy(synth.normal("ops.h", (1, 128, h, w), 0.0, 1.0, 7)))
    inp = torch.relu(torch.from_numpy(synth.normal("ops.inp", (1, 128, h, w), 0.0, 1.0, 7)))
    f1 = torch.from_numpy(synth.normal("ops.f1", (1, 256, h, w), 0.0, 1.0, 7))
    f2 = torch.from_numpy(synth.normal("ops.f2", (1, 256, h, w), 0.0, 1.0, 7))
    coords = torch.from_numpy(g["lookup_coords"])
    HX = torch.zeros(1, h, w, 384)
    HX[..., :128], HX[..., 128:256] = _nhwc(hh), _nhwc(inp)
    state = plan.new_state(h, w, HX.to(dev).to(plan.adt), coords=coords[0].permute(1, 2, 0).reshape(-1, 2).contiguous())
    state.corr0 = ops.conv2d(CV(_nhwc(f1).to(dev)), f2.view(1, 256, h * w).to(dev).contiguous(), None, h * w, 1, 1, alpha=1.0 / 16.0)
    state.levels = ops.corr_pyramid(state.corr0, h, w)
    plan.refine_step(state)
    net_out = state.HX.float().cpu()[..., :128].permute(0, 3, 1, 2)
    e = maxerr(net_out, torch.from_numpy(g["ub_net"]))
    assert e < (1.5e-2 if bf else 5e-6), e
    dfl = state.delta.cpu()[..., :2].permute(0, 3, 1, 2)
    e = maxerr(dfl, torch.from_numpy(g["ub_dflow"]))
    assert e < (4e-3 if bf else 2e-6), e
    _, flow_up, mask = plan.finish(state, 8 * h, 8 * w)
    e = maxerr(mask.cpu().permute(0, 3, 1, 2)[:, ::9], torch.from_numpy(g["ub_mask"]))
    assert e < (1e-3 if bf else 1e-6), e
    e = maxerr(flow_up, torch.from_numpy(g["ub_up"]))
    assert e < (2e-2 if bf else 5e-5), e
    if bf:
        return
    # and the up-sampler alone on the ORACLE's mask / flow (isolates raft.py:64-75)
    fl = coords - torch.stack(torch.meshgrid(torch.arange(w), torch.arange(h), indexing="xy"), 0).float()[None]
    pyr = oracle.corr_pyramid(f1, f2)
    Wt = oracle.to_torch_state(st)
    with torch.no_grad():
        _, mask_o, dfl_o = oracle.update_block(Wt, "raft.update_block", hh, inp, oracle.corr_lookup(pyr, coords), fl)
    assert maxerr(mask_o[:, ::9], torch.from_numpy(g["ub_mask"])) < 1e-4
    f4 = torch.zeros(1, h, w, 4)
    f4[..., :2] = (fl + dfl_o)[0].permute(1, 2, 0)
    up = torch.empty(1, 2, 8 * h, 8 * w, device=dev)
    ops.lib.call("zt_convex_upsample_f32", f4.to(dev), 4, _nhwc(mask_o).to(dev), 576, up, None, h, w, None if dev.type == "cpu" else torch.cuda.current_stream().cuda_stream)
    assert maxerr(up, torch.from_numpy(g["ub_up"])) < 1e-4


def test_probe_tr16_and_bf16_mfma(backend):
    """gfx950 instruction semantics the bf16 kernels rely on, checked on whichever back-end runs (emulator model == chip)."""
    ops, dev, _ = backend
    s = None if dev.type == "cpu" else torch.cuda.current_stream().cuda_stream
    img = torch.arange(16 * 64, dtype=torch.int32).to(torch.int16).view(16, 64)
    for col0 in (0, 16, 44):
        out = torch.zeros(64 * 4, dtype=torch.int16, device=dev)
        ops.lib.call("zt_probe_tr16", img.to(dev), out, col0, s)
        got = out.cpu().view(64, 4)
        for l in range(64):
            g, i = l // 16, l % 16
            for q in range(4):     # lane i of group g receives column i of block row q
                assert int(got[l, q]) == int(img[4 * g + q, col0 + i]), (col0, l, q)
    gen = torch.Generator().manual_seed(3)
    A = torch.randn(16, 32, generator=gen).bfloat16()
    B = torch.randn(32, 16, generator=gen).bfloat16()
    D = torch.zeros(16, 16, device=dev)
    ops.lib.call("zt_probe_mfma_bf16", A.view(torch.int16).to(dev), B.view(torch.int16).to(dev), D, s)
    assert maxerr(D, A.float() @ B.float()) < 1e-4


def _nhwc_bf16(t, ld):
    n = _nhwc(t, ld)
    return n.bfloat16().contiguous()


BF16_CONV = [(3, 48, 3, 8, "lrelu"), (48, 48, 3, 48, "lrelu"), (64, 64, 3, 64, "relu"), (64, 3, 3, 64, "sigmoid_clamp"),
             (48, 6, 1, 48, None), (9, 64, 3, 16, "relu"), (12, 48, 3, 16, None)]


@pytest.mark.parametrize("variant", [2, 1, 3, 0], ids=["tiled", "ws", "rs", "auto"])
@pytest.mark.parametrize("case", BF16_CONV, ids=lambda c: "c%d-%d_k%d" % c[:3])
def test_conv_bf16(backend, case, variant):
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    Cin, Cout, K, ld, act = case
    if variant == 3 and not (K == 3 and Cout >= 48 and act in (None, "relu", "lrelu")):
        pytest.skip("register-stationary kernel covers the 3x3 48/64-cout layers")
    g = torch.Generator().manual_seed(Cin * 17 + Cout)
    H, W = (19, 37) if variant != 2 else (7, 37)          # 19 rows: ragged multiple of the persistent kernels' tile heights
    x = torch.randn(1, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w.bfloat16().float(), b, padding=K // 2)
    ref = {None: lambda t: t, "relu": torch.relu, "lrelu": lambda t: F.leaky_relu(t, 0.2),
           "sigmoid_clamp": lambda t: torch.clamp(torch.sigmoid(t), 1e-4, 1)}[act](ref)
    xd = _nhwc_bf16(x, ld).to(dev)
    wd = ops.repack_weight_bf16(w.to(dev))
    y = ops.conv2d_bf16(CV(xd, 0, Cin), wd, b.to(dev), Cout, K, K, (K // 2, K // 2), act, variant=variant)
    got = y.float().cpu()[..., :Cout].permute(0, 3, 1, 2)
    assert float(((got - ref).abs() - ref.abs() * 2 ** -8).max()) < 2e-3
    if Cout <= 6:
        yp = ops.conv2d_bf16(CV(xd, 0, Cin), wd, b.to(dev), Cout, K, K, (K // 2, K // 2), act, out_planar=True, variant=variant)
        assert maxerr(yp, ref) < 2e-4            # fp32 planar output: only the bf16 inputs differ from the fp32 reference
    # dgrad operator with the LeakyReLU-mask epilogue
    if Cin >= 48 and Cout >= 48:
        dz = torch.randn(1, Cout, H, W, generator=g).bfloat16().float()
        aux = torch.randn(1, Cin, H, W, generator=g).bfloat16().float()
        xg = x.clone().requires_grad_(True)
        (F.conv2d(xg, w.bfloat16().float(), None, padding=K // 2) * dz).sum().backward()
        refg = xg.grad * torch.where(aux > 0, 1.0, 0.2)
        wt = ops.repack_weight_bf16(w.to(dev), transpose_flip=True)
        dx = ops.conv2d_bf16(CV(_nhwc_bf16(dz, Cout).to(dev)), wt, None, Cin, K, K, (K // 2, K // 2), None,
                             aux=_nhwc_bf16(aux, Cin).to(dev), epi=1, variant=variant)
        gotg = dx.float().cpu().permute(0, 3, 1, 2)
        assert float(((gotg - refg).abs() - refg.abs() * 2 ** -8).max()) < 2e-3


@pytest.mark.parametrize("case", [(3, 48, 1), (6, 48, 1), (3, 64, 2), (5, 48, 0)], ids=lambda c: "c%d-%d_epi%d" % c)
def test_conv_bf16_thin_1x1(backend, case):
    """Streaming 1x1 kernel for thin inputs (data gradient of the 48 -> 3 / 48 -> 6 output layers): transposed weights, LeakyReLU /
    ReLU mask epilogue, NaN-filled padding lanes in the input buffer, ragged pixel count."""
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    Cin, Cout, epi = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    H, W = 37, 45
    dz = torch.randn(1, Cin, H, W, generator=g).bfloat16().float()
    w = torch.randn(Cin, Cout, 1, 1, generator=g) / Cout ** 0.5          # forward weight [Cout_fwd = Cin, Cin_fwd = Cout]
    aux = torch.randn(1, Cout, H, W, generator=g).bfloat16().float()
    ref = F.conv_transpose2d(dz, w.bfloat16().float())
    if epi:
        ref = ref * torch.where(aux > 0, 1.0, 0.2 if epi == 1 else 0.0)
    xd = torch.full((1, H, W, 8), float("nan")).bfloat16()
    xd[..., :Cin] = dz.permute(0, 2, 3, 1).bfloat16()
    wt = ops.repack_weight_bf16(w.to(dev), transpose_flip=True)
    y = ops.conv2d_bf16(CV(xd.to(dev), 0, Cin), wt, None, Cout, 1, 1, (0, 0), None, aux=_nhwc_bf16(aux, Cout).to(dev) if epi else None, epi=epi)
    got = y.float().cpu()[..., :Cout].permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert float(((got - ref).abs() - ref.abs() * 2 ** -8).max()) < 1e-3


@pytest.mark.parametrize("case", [(64, 64, "relu", 3), (64, 64, "relu", 0), (48, 48, "lrelu", 1), (9, 64, "relu", 0), (12, 48, "lrelu", 0), (3, 48, "lrelu", 2)], ids=lambda c: "c%d-%d_epi%d" % (c[0], c[1], c[3]))
def test_conv_bf16_rs_pipeline(backend, case):
    """Register-stationary persistent kernel with several tiles per workgroup (> 256 tiles): exercises the double-buffered halo
    and output staging across tile iterations, ragged right and bottom edges, every channel-chunk instantiation."""
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    Cin, Cout, act, epi = case
    g = torch.Generator().manual_seed(Cin + Cout)
    H, W = 291, 421          # 37 x 14 = 518 tiles of 8 x 32
    x = torch.randn(1, Cin, H, W, generator=g).bfloat16().float()
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w.bfloat16().float(), b, padding=1)
    ref = torch.relu(ref) if act == "relu" else F.leaky_relu(ref, 0.2)
    aux = torch.randn(1, Cout, H, W, generator=g).bfloat16().float()
    mag = ref.abs()
    if epi == 3:
        ref = ref + aux
        mag = mag * 2 + ref.abs()       # the residual is added to the bf16-staged conv output: that rounding does not cancel
    elif epi:
        ref = ref * torch.where(aux > 0, 1.0, 0.2 if epi == 1 else 0.0)
    ld = (Cin + 7) // 8 * 8
    y = ops.conv2d_bf16(CV(_nhwc_bf16(x, ld).to(dev), 0, Cin), ops.repack_weight_bf16(w.to(dev)), b.to(dev), Cout, 3, 3, (1, 1), act,
                        aux=_nhwc_bf16(aux, Cout).to(dev) if epi else None, epi=epi, variant=3)
    got = y.float().cpu()[..., :Cout].permute(0, 3, 1, 2)
    assert float(((got - ref).abs() - mag * 2 ** -8).max()) < 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("epi", [3, 2])
def test_conv_bf16_rs_many_tiles_fused_operand(hip_ops, epi):
    """64 -> 64 data-gradient form of the register-stationary kernel with 5-6 tiles per workgroup (2 929 tiles on 512 workgroups):
    the mask / residual chunks are fetched by loads the kernel orders itself -- issued in front of the next halo's DMAs and waited for
    by count (zt_common.h ZT_HIDDEN_LD16 / ZT_HIDDEN_WAIT4) -- on every tile that has a successor; the two-tile case above only
    reaches the drain path.  Same arithmetic contract as test_conv_bf16_rs_pipeline."""
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev = hip_ops
    g = torch.Generator().manual_seed(600 + epi)
    H, W, C = 403, 901, 64
    x = torch.randn(1, C, H, W, generator=g).bfloat16().float()
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    ref = torch.relu(F.conv2d(x, w.bfloat16().float(), b, padding=1))
    aux = torch.randn(1, C, H, W, generator=g).bfloat16().float()
    mag = ref.abs()
    if epi == 3:
        ref = ref + aux
        mag = mag * 2 + ref.abs()
    else:
        ref = ref * (aux > 0)
    auxd = _nhwc_bf16(aux, C).to(dev)
    xd, wd = _nhwc_bf16(x, C).to(dev), ops.repack_weight_bf16(w.to(dev))
    y = ops.conv2d_bf16(CV(xd, 0, C), wd, b.to(dev), C, 3, 3, (1, 1), "relu", aux=auxd, epi=epi, variant=3)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert float(((got - ref).abs() - mag * 2 ** -8).max()) < 2e-3
    # and bit-identical to the same launch with the loads behind the DMAs and a full drain would need a second process (the knob is
    # read once per process); determinism of the ordered form instead: two launches agree bit for bit
    y2 = ops.conv2d_bf16(CV(xd, 0, C), wd, b.to(dev), C, 3, 3, (1, 1), "relu", aux=auxd, epi=epi, variant=3)
    assert torch.equal(y, y2)


BF16_WGRAD = [(3, 48, 3, 8, 48), (48, 48, 3, 48, 48), (48, 3, 1, 48, 8), (9, 64, 3, 16, 64), (64, 64, 3, 64, 64), (64, 3, 3, 64, 8),
              (12, 48, 3, 16, 48), (48, 6, 1, 48, 8)]


@pytest.mark.parametrize("case", BF16_WGRAD, ids=lambda c: "c%d-%d_k%d" % c[:3])
def test_conv_wgrad_bf16(backend, case):
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    Cin, Cout, K, ldx, lddz = case
    g = torch.Generator().manual_seed(Cin + 13 * Cout)
    # 9 x 41: border tiles only (general staging path); 27 x 101: also interior tiles (offset-table fast path), ragged edges
    for (H, W) in ((9, 41), (27, 101)):
        x = torch.randn(1, Cin, H, W, generator=g).bfloat16().float()
        dz = torch.randn(1, Cout, H, W, generator=g).bfloat16().float()
        w = torch.zeros(Cout, Cin, K, K, requires_grad=True)
        (F.conv2d(x, w, None, padding=K // 2) * dz).sum().backward()
        gw = torch.full((Cout, Cin, K, K), 3.0, device=dev)
        gb = torch.full((Cout,), 3.0, device=dev)
        ops.conv2d_wgrad_bf16(CV(_nhwc_bf16(x, ldx).to(dev), 0, Cin), CV(_nhwc_bf16(dz, lddz).to(dev), 0, Cout), Cout, K, K, gw, grad_b=gb)
        assert maxerr(gw, w.grad) < 2e-3 * float(w.grad.abs().max()), maxerr(gw, w.grad)
        assert maxerr(gb, dz.sum(dim=(0, 2, 3))) < 1e-3 * (H * W) ** 0.5
        if K == 3 and Cin <= 16 and Cout == 64:      # folded ReLU mask == the same kernel on a pre-masked dz, bit for bit
            act = torch.relu(torch.randn(1, Cout, H, W, generator=g)).bfloat16().float()
            xd = CV(_nhwc_bf16(x, ldx).to(dev), 0, Cin)
            g1, b1 = torch.zeros((Cout, Cin, K, K), device=dev), torch.zeros((Cout,), device=dev)
            g2, b2 = torch.zeros_like(g1), torch.zeros_like(b1)
            ops.conv2d_wgrad_bf16(xd, CV(_nhwc_bf16(dz, lddz).to(dev), 0, Cout), Cout, K, K, g1, grad_b=b1, relu_mask=_nhwc_bf16(act, 64).to(dev))
            ops.conv2d_wgrad_bf16(xd, CV(_nhwc_bf16(dz * (act > 0), lddz).to(dev), 0, Cout), Cout, K, K, g2, grad_b=b2)
            assert torch.equal(g1, g2) and torch.equal(b1, b2) and float(g1.abs().max()) > 0


@pytest.mark.parametrize("ch", [64, 48])
@pytest.mark.parametrize("nblk", [3, 8, 256])
def test_conv_wgrad64_dma_double_buffer(backend, nblk, ch, monkeypatch):
    """The LDS-DMA form of the 64 -> 64 / 48 -> 48 weight gradient with FEWER workgroups than tiles (ZT_WGRAD_BLOCKS): every workgroup walks
    several tiles through both LDS buffers (nblk 3: plain tile order; 8: the XCD-banded order), ragged right / bottom tiles, against
    autograd and against the register-staged kernel (ZT_WGRAD_DMA=0) on the same operands."""
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(nblk + ch)
    for (H, W) in ((43, 70), (16, 129)):
        x = torch.randn(1, ch, H, W, generator=g).bfloat16().float()
        dz = torch.randn(1, ch, H, W, generator=g).bfloat16().float()
        w = torch.zeros(ch, ch, 3, 3, requires_grad=True)
        (F.conv2d(x, w, None, padding=1) * dz).sum().backward()
        xd, dzd = CV(_nhwc_bf16(x, ch + 8).to(dev), 0, ch), CV(_nhwc_bf16(dz, ch).to(dev), 0, ch)
        res = {}
        for dma in ("1", "0"):
            monkeypatch.setenv("ZT_WGRAD_DMA", dma)
            monkeypatch.setenv("ZT_WGRAD_BLOCKS", str(nblk))
            gw, gb = torch.full((ch, ch, 3, 3), 3.0, device=dev), torch.full((ch,), 3.0, device=dev)
            ops.conv2d_wgrad_bf16(xd, dzd, ch, 3, 3, gw, grad_b=gb)
            res[dma] = (gw.cpu(), gb.cpu())
            assert maxerr(gw, w.grad) < 2e-3 * float(w.grad.abs().max()), (dma, maxerr(gw, w.grad))
            assert maxerr(gb, dz.sum(dim=(0, 2, 3))) < 1e-3 * (H * W) ** 0.5
        # same products, fp32 accumulation in a different order
        assert float((res["1"][0] - res["0"][0]).abs().max()) < 2e-4 * float(w.grad.abs().max())


@pytest.mark.parametrize("cin,ld", [(3, 8), (12, 16)])
def test_conv_wgrad_thin48_dma_double_buffer(backend, cin, ld, monkeypatch):
    """Thin-input (8- / 16-channel pixels) -> 48 weight gradient on the DMA kernel, several tiles per workgroup (both LDS buffers),
    ragged tiles, garbage in the pixel's padding channels (they land in rows of the slab the reduction ignores)."""
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(cin)
    H, W = 43, 70
    x = torch.randn(1, cin, H, W, generator=g).bfloat16().float()
    dz = torch.randn(1, 48, H, W, generator=g).bfloat16().float()
    w = torch.zeros(48, cin, 3, 3, requires_grad=True)
    (F.conv2d(x, w, None, padding=1) * dz).sum().backward()
    xn = _nhwc_bf16(x, ld)
    xn[..., cin:] = 1000.0                                        # untrusted padding lanes
    res = {}
    for dma in ("1", "0"):
        monkeypatch.setenv("ZT_WGRAD_DMA", dma)
        monkeypatch.setenv("ZT_WGRAD_BLOCKS", "3")
        gw, gb = torch.full((48, cin, 3, 3), 3.0, device=dev), torch.full((48,), 3.0, device=dev)
        ops.conv2d_wgrad_bf16(CV(xn.to(dev), 0, cin), CV(_nhwc_bf16(dz, 48).to(dev), 0, 48), 48, 3, 3, gw, grad_b=gb)
        res[dma] = gw.cpu()
        assert maxerr(gw, w.grad) < 2e-3 * float(w.grad.abs().max()), (dma, maxerr(gw, w.grad))
        assert maxerr(gb, dz.sum(dim=(0, 2, 3))) < 1e-3 * (H * W) ** 0.5
    assert float((res["1"] - res["0"]).abs().max()) < 2e-4 * float(w.grad.abs().max())


BF16_GEO = [
    # Cin, Cout, KH, KW, stride, pad, H, W, act, ld
    (3, 64, 7, 7, 2, (3, 3), 20, 36, None, 8),
    (64, 96, 3, 3, 2, (1, 1), 11, 21, "relu", 64),
    (64, 96, 1, 1, 2, (0, 0), 11, 21, None, 64),
    (40, 32, 1, 5, 1, (0, 2), 6, 18, "sigmoid", 40),
    (40, 32, 5, 1, 1, (2, 0), 6, 18, "tanh", 40),
    (2, 128, 7, 7, 1, (3, 3), 6, 10, "relu", 8),
    (324, 256, 1, 1, 1, (0, 0), 5, 9, "relu", 328),
    (256, 126, 3, 3, 1, (1, 1), 9, 40, "relu", 256),     # wide enough to take the MT=2 path as well
]


@pytest.mark.parametrize("case", BF16_GEO, ids=lambda c: "c%d-%d_k%dx%d_s%d" % c[:5])
def test_conv_bf16_raft_geometries(backend, case):
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    Cin, Cout, KH, KW, stride, pad, H, W, act, ld = case
    g = torch.Generator().manual_seed(Cin * 7 + Cout)
    N = 2 if KH == 7 and stride == 2 else 1
    x = torch.randn(N, Cin, H, W, generator=g).bfloat16().float()
    w = torch.randn(Cout, Cin, KH, KW, generator=g) / (Cin * KH * KW) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    ref = F.conv2d(x, w.bfloat16().float(), b, stride=stride, padding=pad)
    ref = {None: lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](ref)
    xd = _nhwc_bf16(x, ld).to(dev)
    wd = ops.repack_weight_bf16(w.to(dev))
    y = ops.conv2d_bf16(CV(xd, 0, Cin), wd, b.to(dev), Cout, KH, KW, pad, act, stride=stride, out_f32=True)
    got = y.cpu()[..., :Cout].permute(0, 3, 1, 2)
    assert maxerr(got, ref) < 5e-4 * max(1.0, float(ref.abs().max())), maxerr(got, ref)


def test_conv_bf16_split_input_and_row_offset(backend):
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(9)
    H, W = 5, 21
    xa = torch.randn(1, 32, H, W, generator=g).bfloat16().float()
    xb = torch.randn(1, 24, H, W, generator=g).bfloat16().float()
    w = torch.randn(48, 56, 1, 5, generator=g) * 0.05
    ref = 0.25 * F.conv2d(torch.cat([xa, xb], 1), w.bfloat16().float(), None, padding=(0, 2))
    big = torch.zeros(1, H, W, 64)
    big[..., 8:32] = _nhwc(xb)
    wd = ops.repack_weight_bf16(w.to(dev))
    out = torch.zeros(1, H, W, 64, dtype=torch.bfloat16, device=dev)
    ops.conv2d_bf16(CV(_nhwc_bf16(xa, 32).to(dev)), wd, None, 48, 1, 5, (0, 2), None, alpha=0.25, x2=CV(big.bfloat16().to(dev), 8, 24),
                    out=CV(out, 16, 48))
    got = out.float().cpu()[..., 16:64].permute(0, 3, 1, 2)
    assert float(((got - ref).abs() - ref.abs() * 2 ** -8).max()) < 2e-3
    assert float(out[..., :16].float().abs().max()) == 0.0
    # 1x1 with an output-row offset into the weight matrix (the cnet tanh / relu split)
    w1 = torch.randn(64, 32, 1, 1, generator=g) * 0.2
    b1 = torch.randn(64, generator=g)
    ref1 = torch.relu(F.conv2d(xa, w1.bfloat16().float(), b1))[:, 32:]
    wd1 = ops.repack_weight_bf16(w1.to(dev))
    y1 = ops.conv2d_bf16(CV(_nhwc_bf16(xa, 32).to(dev)), wd1, b1[32:].contiguous().to(dev), 32, 1, 1, (0, 0), "relu", w_roff=32, out_f32=True)
    assert maxerr(y1.cpu().permute(0, 3, 1, 2), ref1) < 1e-3


@pytest.mark.parametrize("force_fused", [True, False], ids=["fused", "fallback"])
def test_conv3x3_bn_stats_bf16(backend, force_fused, monkeypatch):
    """Enhancer block head (model.py:60-62): conv3x3 64->64 with the train-mode BatchNorm statistics accumulated in the same kernel
    (register-stationary store phase) == the convolution followed by the stand-alone statistics pass; 19 x 37: ragged tiles."""
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    monkeypatch.setenv("ZT_STATS_FUSE_MIN_TILES", "1" if force_fused else "1000000000")
    g = torch.Generator().manual_seed(11)
    H, W = 19, 37
    x = torch.randn(1, 64, H, W, generator=g).bfloat16().float()
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    b = torch.randn(64, generator=g) * 0.1
    xd = _nhwc_bf16(x, 64).to(dev)
    wd = ops.repack_weight_bf16(w.to(dev))
    y, part = ops.conv3x3_bn_stats_bf16(CV(xd, 0, 64), wd, b.to(dev), 64)
    ref = F.conv2d(x, w.bfloat16().float(), b, padding=1)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert float(((got - ref).abs() - ref.abs() * 2 ** -8).max()) < 2e-3
    # the statistics are those of the STORED (bf16-rounded) values, summed in fp32
    s = part.cpu().double().sum(dim=(0, 1))
    assert float((s[0] - got.double().sum(dim=(0, 2, 3))).abs().max()) < 1e-3 * H * W ** 0.5
    assert float((s[1] - (got.double() ** 2).sum(dim=(0, 2, 3))).abs().max()) < 1e-3 * H * W
    sc, sh, mean, rstd = ops.norm_finalize(part, 1, 64, H * W, 0)
    assert maxerr(mean, got.mean(dim=(0, 2, 3)).view(1, 64)) < 1e-5
    assert maxerr(rstd, 1.0 / torch.sqrt(got.double().var(dim=(0, 2, 3), unbiased=False) + 1e-5).view(1, 64)) < 1e-4


def test_conv3x3_dgrad_bn_sums_bf16(backend):
    """Enhancer block backward: data gradient + residual with the BatchNorm-backward sums of the block below accumulated in the
    same kernel's store phase == the plain data gradient (bit-identical output) followed by the stand-alone reduce pass
    (zt_bn_bwd_reduce) on that output; 19 x 37: ragged tiles.  And the whole bn_relu_bwd with / without the fused partials."""
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(21)
    H, W = 19, 37
    mk = lambda s=1.0: _nhwc_bf16(torch.randn(1, 64, H, W, generator=g) * s, 64).to(dev)
    dz, res, zprev = mk(), mk(), mk(2.0)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    wT = ops.repack_weight_bf16(w.to(dev), transpose_flip=True)
    gamma, beta = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    zf = zprev.float()
    mean = zf.mean(dim=(0, 1, 2))
    rstd = 1.0 / torch.sqrt(zf.var(dim=(0, 1, 2), unbiased=False) + 1e-5)
    scale, shift = (gamma * rstd).contiguous(), (beta - mean * gamma * rstd).contiguous()
    df, part = ops.conv3x3_dgrad_bn_sums_bf16(dz, wT, res, zprev, scale, shift, mean.contiguous())
    ref_df = ops.conv2d_bf16(CV(dz), wT, None, 64, 3, 3, (1, 1), None, aux=res, epi=3, variant=3)
    assert torch.equal(df.cpu(), ref_df.cpu())
    # sums of the STORED gradient, masked by the block's ReLU, plain and against the centred pre-activation
    dff, m = df.float().cpu().double(), (zf * scale + shift > 0).cpu()
    gm = torch.where(m, dff, torch.zeros_like(dff))
    s = part.cpu().double().sum(dim=0)
    assert float((s[0] - gm.sum(dim=(0, 1, 2))).abs().max()) < 1e-3 * (H * W) ** 0.5
    assert float((s[1] - (gm * (zf.cpu().double() - mean.cpu().double())).sum(dim=(0, 1, 2))).abs().max()) < 2e-3 * (H * W) ** 0.5
    outs = []
    for p in (part, None):
        dg, db = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
        o = ops.bn_relu_bwd(df, zprev, scale, shift, mean.contiguous(), rstd.contiguous(), dg, db, part=p)
        outs.append((o.float().cpu(), dg.cpu(), db.cpu()))
    assert maxerr(outs[0][1], outs[1][1]) < 1e-3 * (H * W) ** 0.5 and maxerr(outs[0][2], outs[1][2]) < 1e-3 * (H * W) ** 0.5
    assert maxerr(outs[0][0], outs[1][0]) < 2e-2          # bf16 outputs: a last-bit difference of the sums moves an element by one bf16 ulp


@pytest.mark.parametrize("cin,with_bias", [(3, False), (6, False), (6, True)], ids=["c3-plain", "c6-plain", "c6-generic"])
def test_conv1x1_thin_input_bf16(backend, cin, with_bias):
    """Data gradient of Denoise_1/2's 1x1 output layer (3/6 -> 48 channels, LeakyReLU-mask epilogue): the streaming thin-input
    kernel, specialised (no bias / activation) and generic form, vs torch on the bf16-rounded operands."""
    import torch.nn.functional as F
    from importlib import import_module
    CV = import_module("zero-tig_amd.ops").CV
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(cin)
    H, W = 13, 29
    x = torch.randn(1, cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(48, cin, 1, 1, generator=g) / cin ** 0.5)
    b = torch.randn(48, generator=g) * 0.1 if with_bias else None
    aux = torch.randn(1, 48, H, W, generator=g).bfloat16().float()
    ref = F.conv2d(x, w.bfloat16().float(), b) * torch.where(aux > 0, 1.0, 0.2)
    y = ops.conv2d_bf16(CV(_nhwc_bf16(x, 8).to(dev), 0, cin), ops.repack_weight_bf16(w.to(dev)), b.to(dev) if with_bias else None, 48, 1, 1,
                        (0, 0), None, aux=_nhwc_bf16(aux, 48).to(dev), epi=1)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert float(((got - ref).abs() - ref.abs() * 2 ** -8).max()) < 2e-3


def test_raft_stem_conv_bf16(backend):
    """RAFT encoder stem (extractor.py:120): the dedicated bf16 kernel (7 px x 8 ch per kernel row as one K range) vs torch
    conv2d(stride 2, padding 3) on the bf16-rounded operands; 2 images, ragged tiles on both axes."""
    import torch.nn.functional as F
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(21)
    N, H, W = 2, 22, 76
    x = torch.randn(N, 3, H, W, generator=g).bfloat16().float()
    w = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    b = torch.randn(64, generator=g) * 0.1
    ref = F.conv2d(x, w.bfloat16().float(), b, stride=2, padding=3)
    xd = _nhwc_bf16(x, 8).to(dev)
    y = ops.raft_stem_bf16(xd, ops.raft_stem_weight_bf16(w.to(dev)), b.to(dev))
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert float(((got - ref).abs() - ref.abs() * 2 ** -8).max()) < 2e-3
    # fused ReLU (context encoder with the frozen BatchNorm folded into w / b): exactly the ReLU of the stored values
    yr = ops.raft_stem_bf16(xd, ops.raft_stem_weight_bf16(w.to(dev)), b.to(dev), relu=True)
    assert torch.equal(yr, torch.relu(y))
