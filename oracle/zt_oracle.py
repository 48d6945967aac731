"""CPU ORACLE for the Zero-TIG hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional restatement (plain torch-CPU / numpy ops over a flat {name: tensor} weight dict) of the
reference algorithm for the path `Network._loss` -> forward (+RAFT flow + backward warp) -> LossFunction
-> backward -> clip -> Adam.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the product (zero-tig_amd/) never does and fails loudly without its HIP library.

Pinning: the reference ships no tests or golden vectors.  This file is pinned by fixtures
generated in the build container by importing the reference itself (tools/make_golden.py ->
tests/golden/*.npz; checked by tests/test_oracle_golden.py).  One third-party boundary is *parity
unpinned*: torchvision==0.18.1 `equalize` (model.py:234) is not installed/installable here, so
`equalize_u8` restates its published algorithm and is what the reference run used as well.

Each function cites the reference lines (relative to /root/reference) it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS_FWD = 1e-4      # model.py:145
EPS_LOSS = 1e-9     # loss.py:24


# ----------------------------------------------------------------------------- integer ops
def equalize_channel_u8(chan):
    """torchvision 0.18.1 _functional_tensor._scale_channel (third party; restated).  chan: uint8 [H,W]."""
    hist = torch.bincount(chan.reshape(-1).to(torch.int64), minlength=256)
    nz = hist[hist != 0]
    step = torch.div(nz[:-1].sum(), 255, rounding_mode="floor")
    if int(step) == 0:
        return chan
    lut = torch.div(torch.cumsum(hist, 0) + torch.div(step, 2, rounding_mode="floor"), step, rounding_mode="floor")
    lut = torch.cat([lut.new_zeros(1), lut[:-1]]).clamp(0, 255)
    return lut[chan.to(torch.int64)].to(torch.uint8)


def equalize_u8(img):
    """torchvision.transforms.functional.equalize for uint8 [..., 3, H, W] (call site model.py:234)."""
    assert img.dtype == torch.uint8
    if img.dim() == 3:
        return torch.stack([equalize_channel_u8(img[c]) for c in range(img.shape[0])])
    return torch.stack([equalize_u8(x) for x in img])


# ----------------------------------------------------------------------------- stencils (utils/utils.py, loss.py)
def pair_downsample(x):
    """utils.py:15-24: two diagonal 2x2 stride-2 averages."""
    a = x[:, :, 0::2, 0::2][:, :, : x.shape[2] // 2, : x.shape[3] // 2]
    b = x[:, :, 0::2, 1::2][:, :, : x.shape[2] // 2, : x.shape[3] // 2]
    c = x[:, :, 1::2, 0::2][:, :, : x.shape[2] // 2, : x.shape[3] // 2]
    d = x[:, :, 1::2, 1::2][:, :, : x.shape[2] // 2, : x.shape[3] // 2]
    return 0.5 * b + 0.5 * c, 0.5 * a + 0.5 * d


def gauss_kernel_2d(kernlen=21, nsig=1):
    """utils.py:26-39 (fp32 erf CDF differences, sqrt(outer), normalised) -> [kernlen, kernlen] fp32."""
    interval = (2 * nsig + 1.0) / kernlen
    x = torch.linspace(-nsig - interval / 2.0, nsig + interval / 2.0, kernlen + 1)
    cdf = 0.5 * (1 + torch.erf(x / torch.sqrt(torch.tensor(2.0))))
    k1 = torch.diff(cdf)
    raw = torch.sqrt(torch.outer(k1, k1))
    return raw / torch.sum(raw)


def gauss_taps_1d(kernlen=21, nsig=1):
    """Rank-1 factor of gauss_kernel_2d: kernel2d == outer(t, t) up to fp32 rounding."""
    k2 = gauss_kernel_2d(kernlen, nsig).double()
    t = torch.sqrt(torch.diag(k2))
    return (t / t.sum()).float()


def blur21(x):
    """utils.py:52-58: reflect-pad 10 + depthwise 21x21 Gaussian (dense, like the reference)."""
    c = x.shape[1]
    k = gauss_kernel_2d(21, 1).view(1, 1, 21, 21).repeat(c, 1, 1, 1)
    return F.conv2d(F.pad(x, (10, 10, 10, 10), mode="reflect"), k, groups=c)


def local_mean_reflect(x, patch=5):
    """utils.py:41-50 LocalMean: reflect-pad 2, 5x5 mean."""
    p = patch // 2
    xp = F.pad(x, (p, p, p, p), mode="reflect")
    return xp.unfold(2, patch, 1).unfold(3, patch, 1).mean(dim=(4, 5))


def local_std_reflect(gray, patch=5):
    """loss.py:123-131: reflect-pad, biased 5x5 variance, sqrt(var + 1e-9)."""
    p = patch // 2
    xp = F.pad(gray, (p, p, p, p), mode="reflect")
    win = xp.unfold(2, patch, 1).unfold(3, patch, 1)
    mu = win.mean(dim=(4, 5), keepdim=True)
    return torch.sqrt(((win - mu) ** 2).mean(dim=(4, 5)) + 1e-9)


def gray_144(x):
    """loss.py:133-136: 0.144*c0 + 0.587*c1 + 0.299*c2 (sic)."""
    return (0.144 * x[:, 0] + 0.5870 * x[:, 1] + 0.299 * x[:, 2]).unsqueeze(1)


def texture_mask(a, b, c_const=1e-5, thr=0.975):
    """loss.py:99-121 TextureDifference."""
    s1, s2 = local_std_reflect(gray_144(a)), local_std_reflect(gray_144(b))
    ratio = (2 * s1 * s2) / (s1 ** 2 + s2 ** 2 + c_const)
    return (ratio > thr).to(a.dtype), ratio


def local_variance_zero(x):
    """utils.py:60-79 calculate_local_variance == box0((x - box0(x)/25)^2)/25 with zero padding."""
    avg = F.avg_pool2d(x, 5, 1, 2)                                  # count_include_pad=True
    d2 = (x - avg) ** 2
    return F.avg_pool2d(d2, 5, 1, 2)


def ycc_flat(x):
    """loss.py:178-190 rgb2yCbCr: a [*,3]x[3,3] product over the FLAT NCHW memory (not per pixel)."""
    mat = torch.tensor([[0.257, -0.148, 0.439], [0.564, -0.291, -0.368], [0.098, 0.439, -0.071]], dtype=torch.float32)
    bias = torch.tensor([16.0 / 255.0, 128.0 / 255.0, 128.0 / 255.0], dtype=torch.float32)
    return (x.contiguous().view(-1, 3).float().mm(mat) + bias).view(x.shape[0], 3, x.shape[2], x.shape[3])


SMOOTH_OFFSETS = [(1, 0), (0, 1), (1, 1), (1, -1), (2, 0), (0, 2), (2, 1), (2, -1), (1, 2), (1, -2), (2, 2), (2, -2)]


def _shift_pair(t, dy, dx):
    """Views (t[p], t[p+d]) over the common valid region for offset d=(dy>=0, dx any)."""
    H, W = t.shape[2], t.shape[3]
    if dx >= 0:
        return t[:, :, : H - dy, : W - dx], t[:, :, dy:, dx:]
    return t[:, :, : H - dy, -dx:], t[:, :, dy:, : W + dx]


def smooth_loss(L2, s2, sigma=10.0):
    """loss.py:173-311 SmoothLoss: 24 terms = 2 x 12 offsets (each listed twice with swapped operands)."""
    y = ycc_flat(L2)
    sc = -1.0 / (2 * sigma * sigma)
    total = 0.0
    for dy, dx in SMOOTH_OFFSETS:
        ya, yb = _shift_pair(y, dy, dx)
        oa, ob = _shift_pair(s2, dy, dx)
        w = torch.exp(((ya - yb) ** 2).sum(dim=1, keepdim=True) * sc)
        g = w * (oa - ob).abs().sum(dim=1, keepdim=True)
        total = total + 2.0 * g.mean()
    return total


def tv_loss(x):
    """loss.py:139-152 L_TV."""
    B, _, H, W = x.shape
    h_tv = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum()
    return 2 * (h_tv / ((H - 1) * W) + w_tv / (H * (W - 1))) / B


# ----------------------------------------------------------------------------- enhancement / denoising nets
def _conv(Wt, name, x, stride=1, padding=0):
    return F.conv2d(x, Wt[name + ".weight"], Wt[name + ".bias"], stride=stride, padding=padding)


def denoise(Wt, prefix, x):
    """model.py:15-44 Denoise_1 / Denoise_2: conv3x3+LReLU(0.2), conv3x3+LReLU, conv1x1."""
    y = F.leaky_relu(_conv(Wt, prefix + ".conv1", x, padding=1), 0.2)
    y = F.leaky_relu(_conv(Wt, prefix + ".conv2", y, padding=1), 0.2)
    return _conv(Wt, prefix + ".conv3", y)


def enhancer(Wt, x, training=True, momentum=0.1, bn_eps=1e-5):
    """model.py:47-81 Enhancer: in_conv+ReLU; 3x fea += ReLU(BN(conv(fea))) with ONE shared conv+BN; out_conv+sigmoid;
    clamp(1e-4, 1).  In training mode the shared BN's running stats are updated three times (in place in Wt)."""
    fea = F.relu(_conv(Wt, "enhance.in_conv.0", x, padding=1))
    for _ in range(3):
        z = _conv(Wt, "enhance.conv.0", fea, padding=1)
        z = F.batch_norm(z, Wt["enhance.conv.1.running_mean"], Wt["enhance.conv.1.running_var"],
                         Wt["enhance.conv.1.weight"], Wt["enhance.conv.1.bias"], training, momentum, bn_eps)
        if training:
            Wt["enhance.conv.1.num_batches_tracked"] += 1
        fea = fea + F.relu(z)
    out = torch.sigmoid(_conv(Wt, "enhance.out_conv.0", fea, padding=1))
    return torch.clamp(out, 0.0001, 1)


# ----------------------------------------------------------------------------- RAFT (frozen, eval)
def _norm(Wt, prefix, x, kind):
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    return F.batch_norm(x, Wt[prefix + ".running_mean"], Wt[prefix + ".running_var"], Wt[prefix + ".weight"],
                        Wt[prefix + ".bias"], False, 0.1, 1e-5)


def _res_block(Wt, p, x, kind, stride):
    """extractor.py:5-55 ResidualBlock."""
    y = F.relu(_norm(Wt, p + ".norm1", _conv(Wt, p + ".conv1", x, stride=stride, padding=1), kind))
    y = F.relu(_norm(Wt, p + ".norm2", _conv(Wt, p + ".conv2", y, padding=1), kind))
    if stride != 1:
        x = _norm(Wt, p + ".norm3", _conv(Wt, p + ".downsample.0", x, stride=stride), kind)
    return F.relu(x + y)


def basic_encoder(Wt, prefix, x, kind):
    """extractor.py:117-191 BasicEncoder (dropout 0)."""
    y = F.relu(_norm(Wt, prefix + ".norm1", _conv(Wt, prefix + ".conv1", x, stride=2, padding=3), kind))
    for li, stride in ((1, 1), (2, 2), (3, 2)):
        y = _res_block(Wt, "%s.layer%d.0" % (prefix, li), y, kind, stride)
        y = _res_block(Wt, "%s.layer%d.1" % (prefix, li), y, kind, 1)
    return _conv(Wt, prefix + ".conv2", y)


def corr_pyramid(f1, f2, levels=4):
    """corr.py:13-27, 52-60: all-pairs dot / sqrt(C), then 3x avg_pool2d(2,2)."""
    B, C, h, w = f1.shape
    corr = torch.matmul(f1.view(B, C, h * w).transpose(1, 2), f2.view(B, C, h * w)) / torch.sqrt(torch.tensor(C).float())
    corr = corr.reshape(B * h * w, 1, h, w)
    pyr = [corr]
    for _ in range(levels - 1):
        corr = F.avg_pool2d(corr, 2, stride=2)
        pyr.append(corr)
    return pyr


def bilinear_sampler_px(img, coords):
    """utils.py:285-299: grid_sample in pixel coordinates, align_corners=True, zeros padding."""
    H, W = img.shape[-2:]
    xg = 2 * coords[..., 0:1] / (W - 1) - 1
    yg = 2 * coords[..., 1:2] / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xg, yg], dim=-1), align_corners=True)


def corr_lookup(pyr, coords, radius=4):
    """corr.py:29-50: per level 9x9 window; delta=stack(meshgrid(dy,dx)) added to (x,y) -> first window axis moves x."""
    r = radius
    B, _, h, w = coords.shape
    c = coords.permute(0, 2, 3, 1)
    d = torch.linspace(-r, r, 2 * r + 1)
    delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1).view(1, 2 * r + 1, 2 * r + 1, 2)
    out = []
    for i, corr in enumerate(pyr):
        cl = c.reshape(B * h * w, 1, 1, 2) / 2 ** i + delta
        out.append(bilinear_sampler_px(corr, cl).view(B, h, w, -1))
    return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def update_block(Wt, p, net, inp, corr, flow):
    """update.py:79-136 BasicMotionEncoder + SepConvGRU + FlowHead + mask head."""
    e = p + ".encoder"
    cor = F.relu(_conv(Wt, e + ".convc1", corr))
    cor = F.relu(_conv(Wt, e + ".convc2", cor, padding=1))
    flo = F.relu(_conv(Wt, e + ".convf1", flow, padding=3))
    flo = F.relu(_conv(Wt, e + ".convf2", flo, padding=1))
    mot = torch.cat([F.relu(_conv(Wt, e + ".conv", torch.cat([cor, flo], 1), padding=1)), flow], 1)
    x = torch.cat([inp, mot], 1)
    g = p + ".gru"
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([net, x], 1)
        z = torch.sigmoid(_conv(Wt, g + ".convz" + sfx, hx, padding=pad))
        r = torch.sigmoid(_conv(Wt, g + ".convr" + sfx, hx, padding=pad))
        q = torch.tanh(_conv(Wt, g + ".convq" + sfx, torch.cat([r * net, x], 1), padding=pad))
        net = (1 - z) * net + z * q
    dflow = _conv(Wt, p + ".flow_head.conv2", F.relu(_conv(Wt, p + ".flow_head.conv1", net, padding=1)), padding=1)
    mask = 0.25 * _conv(Wt, p + ".mask.2", F.relu(_conv(Wt, p + ".mask.0", net, padding=1)))
    return net, mask, dflow


def convex_upsample(flow, mask):
    """raft.py:64-75 upsample_flow."""
    N, _, H, W = flow.shape
    m = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(m * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, 8 * H, 8 * W)


def raft_forward(Wt, img1, img2, iters=12, prefix="raft", return_aux=False):
    """raft.py:77-138 RAFT.forward(test_mode): centred replicate pad to /8, 2*(x/255)-1, fnet(IN) on both,
    cnet(BN eval) on img1, 12 refinement iterations; returns (flow_low, flow_up) at the PADDED size."""
    ht, wd = img1.shape[-2:]
    ph = (((ht // 8) + 1) * 8 - ht) % 8
    pw = (((wd // 8) + 1) * 8 - wd) % 8
    pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]
    i1 = 2 * (F.pad(img1, pad, mode="replicate") / 255.0) - 1.0
    i2 = 2 * (F.pad(img2, pad, mode="replicate") / 255.0) - 1.0
    fm = basic_encoder(Wt, prefix + ".fnet", torch.cat([i1, i2], 0), "instance")
    f1, f2 = fm[: i1.shape[0]].float(), fm[i1.shape[0]:].float()
    pyr = corr_pyramid(f1, f2)
    cn = basic_encoder(Wt, prefix + ".cnet", i1, "batch")
    net, inp = torch.tanh(cn[:, :128]), torch.relu(cn[:, 128:])
    N, _, H, W = i1.shape
    ys, xs = torch.meshgrid(torch.arange(H // 8), torch.arange(W // 8), indexing="ij")
    coords0 = torch.stack([xs, ys], dim=0).float()[None].repeat(N, 1, 1, 1)
    coords1 = coords0.clone()
    aux = {"fmap1": f1, "fmap2": f2, "net0": net, "inp": inp, "corr0": None}
    flow_up = None
    for it in range(iters):
        corr = corr_lookup(pyr, coords1)
        if it == 0:
            aux["corr0"] = corr
        net, mask, dflow = update_block(Wt, prefix + ".update_block", net, inp, corr, coords1 - coords0)
        coords1 = coords1 + dflow
        flow_up = convex_upsample(coords1 - coords0, mask)
    if return_aux:
        return coords1 - coords0, flow_up, aux
    return coords1 - coords0, flow_up


# ----------------------------------------------------------------------------- warp + cache update
def warp_coords(flow, h_dst, w_dst):
    """utils.py:203-222: base grid minus flow, scales SWAPPED (x by h_scale, y by w_scale), bilinear-upsampled
    coordinate maps, normalised with the align_corners=True formula.  Returns the [-1,1] grid [B,h_dst,w_dst,2]."""
    B, _, H, W = flow.shape
    h_scale = float(h_dst) / float(H)
    w_scale = float(w_dst) / float(W)
    gy, gx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    map_x = (gx[None] - flow[:, 0]) * h_scale
    map_y = (gy[None] - flow[:, 1]) * w_scale
    map_x = F.interpolate(map_x.unsqueeze(1), (h_dst, w_dst), mode="bilinear")
    map_y = F.interpolate(map_y.unsqueeze(1), (h_dst, w_dst), mode="bilinear")
    return torch.stack((map_x / ((w_dst - 1) / 2) - 1, map_y / ((h_dst - 1) / 2) - 1), dim=-1).squeeze(1)


def warp_tensor(flow, img):
    """utils.py:203-230: grid_sample(bilinear, zeros, align_corners=False) on warp_coords."""
    grid = warp_coords(flow, img.shape[-2], img.shape[-1])
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def _fma32(a, b, c):
    """fp32 fused multiply-add (one rounding), emulated through float64 (products of two fp32 are exact there)."""
    return (a.double() * b.double() + c.double()).float()


def warp_taps(flow, h_dst, w_dst):
    """Integer tap indices (x0, y0) of warp_tensor's grid_sample, with ATen's CPU arithmetic
    (GridSamplerKernel.cpp, align_corners=False): ix = fma(gx + 1, W/2, -0.5); x0 = floor(ix).
    Established bit-for-bit against torch 2.10 CPU in the build container.  int32 [B,h_dst,w_dst,2]."""
    grid = warp_coords(flow, h_dst, w_dst)
    half = torch.tensor(-0.5)
    ix = _fma32(grid[..., 0] + 1, torch.tensor(w_dst / 2.0, dtype=torch.float32), half)
    iy = _fma32(grid[..., 1] + 1, torch.tensor(h_dst / 2.0, dtype=torch.float32), half)
    return torch.stack([torch.floor(ix), torch.floor(iy)], dim=-1).to(torch.int32)


def raft_inputs(last_H3, L2, of_scale):
    """model.py:221-235: bilinear downscale; previous frame x255 (NOT equalised); current frame -> uint8 (truncating)
    -> histogram equalisation -> float."""
    ht, wd = last_H3.shape[-2] // of_scale, last_H3.shape[-1] // of_scale
    a = F.interpolate(last_H3, (ht, wd), mode="bilinear") * 255
    b8 = (F.interpolate(L2, (ht, wd), mode="bilinear") * 255).to(torch.uint8)
    return a.float(), equalize_u8(b8).float(), b8


def update_cache(Wt, last_H3, last_s3, L2, of_scale):
    """model.py:221-259."""
    a, b, _ = raft_inputs(last_H3, L2, of_scale)
    with torch.no_grad():
        flow_low, flow_up = raft_forward(Wt, a, b, iters=12)
    return warp_tensor(flow_up, last_H3), warp_tensor(flow_up, last_s3), flow_low, flow_up


# ----------------------------------------------------------------------------- Network.forward / Finetunemodel.forward
FORWARD_NAMES = ["L_pred1", "L_pred2", "L2", "s2", "s21", "s22", "H2", "H11", "H12", "H13", "s13", "H14", "s14", "H3",
                 "s3", "H3_pred", "H4_pred", "L_pred1_L_pred2_diff", "H3_denoised1_H3_denoised2_diff", "H2_blur",
                 "H3_blur", "H3_denoised1", "H3_denoised2"]


def network_forward(Wt, cache, inp, is_new_seq, of_scale=3, training=True):
    """model.py:144-203.  `cache` is a dict holding last_H3/last_s3 (detached) between calls.
    Returns the 23-tuple in the reference order plus an aux dict (flow etc.)."""
    eps = EPS_FWD
    x = inp + eps
    L11, L12 = pair_downsample(x)
    L_pred1 = L11 - denoise(Wt, "denoise_1", L11)
    L_pred2 = L12 - denoise(Wt, "denoise_1", L12)
    L2 = torch.clamp(x - denoise(Wt, "denoise_1", x), eps, 1)
    aux = {}
    if is_new_seq:
        wpH, wps = torch.zeros_like(L2), torch.zeros_like(L2)
        wpH1 = wpH2 = wps1 = wps2 = torch.zeros_like(L11)
    else:
        wpH, wps, flow_low, flow_up = update_cache(Wt, cache["last_H3"], cache["last_s3"], L2.detach(), of_scale)
        aux.update(flow_low=flow_low, flow_up=flow_up)
        wpH1, wpH2 = pair_downsample(wpH)
        wps1, wps2 = pair_downsample(wps)
    aux.update(wpH=wpH, wps=wps)
    s2 = enhancer(Wt, torch.cat([wpH, wps, L2], 1).detach(), training)
    s21, s22 = pair_downsample(s2)
    H2 = torch.clamp(x / s2, eps, 1)
    H11 = torch.clamp(L11 / s21, eps, 1)
    H12 = torch.clamp(L12 / s22, eps, 1)
    H3_pred = torch.clamp(torch.cat([H11, s21], 1).detach() - denoise(Wt, "denoise_2", torch.cat([wpH1, wps1, H11, s21], 1)), eps, 1)
    H4_pred = torch.clamp(torch.cat([H12, s22], 1).detach() - denoise(Wt, "denoise_2", torch.cat([wpH2, wps2, H12, s22], 1)), eps, 1)
    H5_pred = torch.clamp(torch.cat([H2, s2], 1).detach() - denoise(Wt, "denoise_2", torch.cat([wpH, wps, H2, s2], 1)), eps, 1)
    H13, s13 = H3_pred[:, :3], H3_pred[:, 3:]
    H14, s14 = H4_pred[:, :3], H4_pred[:, 3:]
    H3, s3 = H5_pred[:, :3], H5_pred[:, 3:]
    m_l, _ = texture_mask(L_pred1, L_pred2)
    H3d1, H3d2 = pair_downsample(H3)
    m_h, ratio = texture_mask(H3d1, H3d2)
    aux["mask_ratio"] = ratio
    H1 = torch.clamp(L2 / s2, 0, 1)
    outs = (L_pred1, L_pred2, L2, s2, s21, s22, H2, H11, H12, H13, s13, H14, s14, H3, s3, H3_pred, H4_pred,
            m_l, m_h, blur21(H1), blur21(H3), H3d1, H3d2)
    return outs, aux


def finetune_forward(Wt, cache, inp, is_new_seq, of_scale=3):
    """model.py:312-340 Finetunemodel.forward (inference twin; new-sequence D2 temporal slots = H2)."""
    eps = EPS_FWD
    x = inp + eps
    L2 = torch.clamp(x - denoise(Wt, "denoise_1", x), eps, 1)
    if is_new_seq:
        wpH, wps = torch.zeros_like(L2), torch.zeros_like(L2)
    else:
        wpH, wps, _, _ = update_cache(Wt, cache["last_H3"], cache["last_s3"], L2.detach(), of_scale)
    s2 = enhancer(Wt, torch.cat([wpH, wps, L2], 1).detach(), training=False)
    H2 = torch.clamp(x / s2, eps, 1)
    if is_new_seq:
        wpH, wps = H2.detach(), H2.detach()
    H5 = torch.clamp(torch.cat([H2, s2], 1).detach() - denoise(Wt, "denoise_2", torch.cat([wpH, wps, H2, s2], 1)), eps, 1)
    H3, s3 = H5[:, :3], H5[:, 3:]
    cache["last_H3"], cache["last_s3"] = H3.detach(), s3.detach()
    return H2, H3, s3


# ----------------------------------------------------------------------------- LossFunction
def loss_terms(inp, outs, is_WB=False):
    """loss.py:23-78; returns (total, {term: value}).  `inp` is the un-offset network input (model.py:210)."""
    (L_pred1, L_pred2, L2, s2, s21, s22, H2, H11, H12, H13, s13, H14, s14, H3, s3, H3_pred, H4_pred,
     _m_l, m_h, H2_blur, H3_blur) = outs[:21]
    eps = EPS_LOSS
    mse = F.mse_loss
    x = inp + eps
    L2d = L2.detach()
    if is_WB:
        ef = (0.3 / (torch.mean(L2d, dim=(2, 3)) + eps)).unsqueeze(2).unsqueeze(3)
    else:
        lum = L2d[:, 2] * 0.299 + L2d[:, 1] * 0.587 + L2d[:, 0] * 0.144
        ef = (0.5 / (torch.mean(lum, dim=(1, 2)) + eps)).view(-1, 1, 1, 1).repeat(1, 3, 1, 1)
    ef = torch.clamp(ef, 1, 25)
    ratio = torch.pow(0.7, -ef) / ef
    nl = torch.clamp(L2d / s2, eps, 0.8)
    eb = torch.pow(L2d * ef, ef)
    ceb = torch.clamp(eb * ratio, eps, 1)
    cal = torch.clamp(L2d * ef, eps, 1)
    t = {}
    t["enh_s2"] = mse(s2, ceb) * 700
    t["enh_norm"] = mse(nl, cal) * 1000
    t["smooth"] = smooth_loss(L2d, s2) * 5
    t["tv"] = tv_loss(s2) * 1600
    L11, L12 = pair_downsample(x)
    t["res1_a"] = mse(L11, L_pred2) * 1000
    t["res1_b"] = mse(L12, L_pred1) * 1000
    d1, d2 = pair_downsample(L2)
    t["res1_c"] = mse(L_pred1, d1) * 1000
    t["res1_d"] = mse(L_pred2, d2) * 1000
    t["res2_a"] = mse(H3_pred, torch.cat([H12.detach(), s22.detach()], 1)) * 1000
    t["res2_b"] = mse(H4_pred, torch.cat([H11.detach(), s21.detach()], 1)) * 1000
    H3d1, H3d2 = pair_downsample(H3)
    t["res2_c"] = mse(H3_pred[:, 0:3], H3d1) * 1000
    t["res2_d"] = mse(H4_pred[:, 0:3], H3d2) * 1000
    t["color"] = mse(H2_blur.detach(), H3_blur) * 10000
    t["ill"] = mse(s2.detach(), s3) * 1000
    lm1, lm2 = local_mean_reflect(H3d1), local_mean_reflect(H3d2)
    wd1 = (1 - m_h) * lm1 + H3d1 * m_h
    wd2 = (1 - m_h) * lm2 + H3d1 * m_h            # sic: H3_denoised1 (loss.py:71)
    t["inter_a"] = mse(H3d1, wd1) * 10000
    t["inter_b"] = mse(H3d2, wd2) * 10000
    t["var"] = mse(local_variance_zero(H2), local_variance_zero(H3 - H2)) * 1000
    total = 0
    for k in ["enh_s2", "enh_norm", "smooth", "tv", "res1_a", "res1_b", "res1_c", "res1_d", "res2_a", "res2_b",
              "res2_c", "res2_d", "color", "ill", "inter_a", "inter_b", "var"]:
        total = total + t[k]
    return total, t


# ----------------------------------------------------------------------------- training step (train.py:119-133)
TRAINABLE_PREFIXES = ("enhance.in_conv.0", "enhance.conv.0", "enhance.conv.1", "enhance.out_conv.0",
                      "denoise_1.conv1", "denoise_1.conv2", "denoise_1.conv3",
                      "denoise_2.conv1", "denoise_2.conv2", "denoise_2.conv3")


def trainable_names():
    return [p + s for p in TRAINABLE_PREFIXES for s in (".weight", ".bias")]


def to_torch_state(np_state, dtype=torch.float32):
    """numpy state (synth.make_state) -> torch tensors with aliasing preserved."""
    out, seen = {}, {}
    for k, v in np_state.items():
        if id(v) not in seen:
            t = torch.from_numpy(np.array(v))
            seen[id(v)] = t.to(dtype) if t.is_floating_point() else t
        out[k] = seen[id(v)]
    return out


class OracleTrainer:
    """Holds weights, the recurrent cache and Adam state; step() == one iteration of train.py:119-133."""

    def __init__(self, Wt, is_WB=False, of_scale=3, lr=1e-4, betas=(0.9, 0.999), wd=3e-4, adam_eps=1e-8):
        self.W = Wt
        self.names = trainable_names()
        for n in self.names:
            self.W[n].requires_grad_(True)
        # enhance.blocks.* aliases must follow their owner
        self.is_WB, self.of_scale = is_WB, of_scale
        self.lr, self.betas, self.wd, self.adam_eps = lr, betas, wd, adam_eps
        self.m = {n: torch.zeros_like(self.W[n]) for n in self.names}
        self.v = {n: torch.zeros_like(self.W[n]) for n in self.names}
        self.t = 0
        self.cache = {}
        self.training = True

    def loss(self, inp, is_new_seq):
        outs, aux = network_forward(self.W, self.cache, inp, is_new_seq, self.of_scale, self.training)
        total, terms = loss_terms(inp, outs, self.is_WB)
        self.cache["last_H3"], self.cache["last_s3"] = outs[13].detach(), outs[14].detach()   # model.py:214-219
        return total, terms, outs, aux

    def step(self, inp, is_new_seq, max_norm=5.0):
        for n in self.names:
            self.W[n].grad = None
        total, terms, outs, aux = self.loss(inp, is_new_seq)
        total.backward()
        grads = {n: self.W[n].grad.detach().clone() for n in self.names}
        # clip_grad_norm_(params, 5): g *= min(1, 5/(norm+1e-6))
        gn = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
        coef = torch.clamp(max_norm / (gn + 1e-6), max=1.0)
        self.t += 1
        b1, b2 = self.betas
        with torch.no_grad():
            for n in self.names:
                g = grads[n] * coef + self.wd * self.W[n]
                self.m[n].mul_(b1).add_(g, alpha=1 - b1)
                self.v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
                bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
                denom = (self.v[n].sqrt() / math.sqrt(bc2)) + self.adam_eps
                self.W[n].addcdiv_(self.m[n], denom, value=-self.lr / bc1)
        return total.detach(), terms, outs, aux, grads, gn


def psnr_u8(a, b):
    """evals.py:83-85 PSNR definition on round(x*255) uint8 images."""
    a8 = torch.clamp(torch.round(a * 255), 0, 255)
    b8 = torch.clamp(torch.round(b * 255), 0, 255)
    mse = torch.mean((a8 - b8) ** 2).item()
    return float("inf") if mse == 0 else 10 * math.log10(255.0 ** 2 / mse)
