"""Tensor-level wrappers over the C ABI: allocate outputs with torch (plumbing), launch the HIP kernels on the
current stream.  No compute happens in torch here."""
import os

import torch

from .lib import current_stream


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


# partial rows of a statistics pass over a >= 1 Mpixel map (tuning knob)
_NBLK_BIG = int(os.environ.get("ZT_NBLK_BIG", "1024"))

class CV:
    """Channel view of an NHWC fp32 buffer [N,H,W,ld]: channels [off, off+C)."""

    def __init__(self, t, off=0, C=None):
        assert t.dtype in (torch.float32, torch.bfloat16) and t.is_contiguous() and t.dim() == 4
        self.t, self.off = t, off
        self.es = t.element_size()
        self.N, self.H, self.W, self.ld = t.shape
        self.C = self.ld - off if C is None else C
        assert 0 <= off and off + self.C <= self.ld

    @property
    def ptr(self):
        return self.t.data_ptr() + self.es * self.off


def _cv(x):
    return x if isinstance(x, CV) else CV(x)


def _dt(t):
    """storage-type code of the C ABI: 0 = fp32, 1 = bf16"""
    return 1 if t.dtype == torch.bfloat16 else 0


ACT = {None: 0, "none": 0, "relu": 1, "lrelu": 2, "sigmoid": 3, "tanh": 4, "sigmoid_clamp": 5}


class Ops:
    def __init__(self, lib):
        self.lib = lib
        self._slab = {}
        self._tickets = {}

    def _s(self, t):
        return current_stream(t.device)

    # live roofline measurement (bench.py): `self.profile = {"match": {conv geometry tuple: name}, "events": {name: [(e0, e1)]}}`
    # brackets the matching launches with HIP events on the launch stream (torch's current stream IS the launch stream here)
    profile = None

    def _ev_begin(self, name):
        if self.profile is None or name is None:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (name, e0)

    def _ev_end(self, tok):
        if tok is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.profile["events"].setdefault(tok[0], []).append((tok[1], e1))

    # ---- utils/utils.py:203-230 warp_tensor (x2 fused) ---------------------------------------------------
    def warp2(self, flow, imgA, imgB=None, want_taps=False):
        """flow [1,2,Hf,Wf], imgA/imgB [1,C,H,W] -> warped A, warped B (and int32 taps [H,W,2])."""
        _f32c(flow), _f32c(imgA)
        _, C, H, W = imgA.shape
        Hf, Wf = flow.shape[-2:]
        outA = torch.empty_like(imgA)
        outB = torch.empty_like(imgB) if imgB is not None else None
        taps = torch.empty((H, W, 2), dtype=torch.int32, device=imgA.device) if want_taps else None
        tok = self._ev_begin("warp2")
        self.lib.call("zt_warp2_f32", flow, Hf, Wf, imgA, imgB, outA, outB, taps, C, H, W, self._s(imgA))
        self._ev_end(tok)
        return (outA, outB, taps) if want_taps else (outA, outB)

    # ---- stencils -------------------------------------------------------------------------------------------
    def pair_down(self, x):
        _f32c(x)
        _, C, H, W = x.shape
        o1 = torch.empty((1, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        o2 = torch.empty_like(o1)
        self.lib.call("zt_pair_down_f32", x, o1, o2, C, H, W, self._s(x))
        return o1, o2

    def pair_down_adj(self, g1, g2, H, W, out=None):
        C = g1.shape[1]
        acc = out is not None
        if out is None:
            out = torch.empty((1, C, H, W), dtype=torch.float32, device=g1.device)
        self.lib.call("zt_pair_down_adj_f32", g1, g2, out, C, H, W, int(acc), self._s(g1))
        return out

    def gauss_taps(self):
        """1-D factor of the reference's 21x21 kernel (utils.py:26-39): sqrt of CDF differences, normalised (host, fp64)."""
        if not hasattr(self, "_taps"):
            import math
            nsig, n = 1.0, 21
            interval = (2 * nsig + 1.0) / n
            xs = [(-nsig - interval / 2.0) + i * (2 * nsig + interval) / n for i in range(n + 1)]
            cdf = [0.5 * (1 + math.erf(v / math.sqrt(2.0))) for v in xs]
            k = [math.sqrt(cdf[i + 1] - cdf[i]) for i in range(n)]
            tot = sum(k)
            self._taps = torch.tensor([v / tot for v in k], dtype=torch.float32)
        return self._taps

    def blur21(self, x, tmp=None):
        _f32c(x)
        _, C, H, W = x.shape
        tmp = torch.empty_like(x) if tmp is None else tmp
        out = torch.empty_like(x)
        self.lib.call("zt_blur21_f32", x, tmp, out, self.gauss_taps(), C, H, W, self._s(x))
        return out

    def blur21_adj(self, g, out=None, tmp=None):
        _, C, H, W = g.shape
        acc = out is not None
        out = torch.empty_like(g) if out is None else out
        tmp = torch.empty_like(g) if tmp is None else tmp
        self.lib.call("zt_blur21_adj_f32", g, tmp, out, self.gauss_taps(), C, H, W, int(acc), self._s(g))
        return out

    def box5_reflect(self, x):
        _, C, H, W = x.shape
        out = torch.empty_like(x)
        self.lib.call("zt_box5_reflect_f32", x, out, C, H, W, self._s(x))
        return out

    def box5_reflect_adj(self, g, scale=1.0, out=None):
        _, C, H, W = g.shape
        acc = out is not None
        out = torch.empty_like(g) if out is None else out
        self.lib.call("zt_box5_reflect_adj_f32", g, out, C, H, W, float(scale), int(acc), self._s(g))
        return out

    def localvar_fwd(self, a, b=None, want_D=True):
        _, C, H, W = a.shape
        D = torch.empty_like(a) if want_D else None
        V = torch.empty_like(a)
        self.lib.call("zt_localvar_fwd_f32", a, b, D, V, C, H, W, self._s(a))
        return D, V

    def localvar_bwd(self, D, gV, sign=1.0, out=None):
        _, C, H, W = D.shape
        acc = out is not None
        out = torch.empty_like(D) if out is None else out
        self.lib.call("zt_localvar_bwd_f32", D, gV, out, C, H, W, float(sign), int(acc), self._s(D))
        return out

    def texture_mask(self, a, b, want_ratio=False):
        _, C, H, W = a.shape
        assert C == 3
        m = torch.empty((1, 1, H, W), dtype=torch.float32, device=a.device)
        r = torch.empty_like(m) if want_ratio else None
        self.lib.call("zt_texture_mask_f32", a, b, m, r, H, W, self._s(a))
        return (m, r) if want_ratio else m

    def ycc_flat(self, x):
        out = torch.empty_like(x)
        self.lib.call("zt_ycc_flat_f32", x, out, x.numel(), self._s(x))
        return out

    # ---- output side (zt_io.hip) ------------------------------------------------------------------------------
    def quantize_u8(self, x, mode=0):
        """[1,3,H,W] fp32 in [0,1] -> uint8 [H,W,3] on the device (mode 0: predict.py save_images truncation; 1: evals.py round)."""
        _f32c(x)
        _, C, H, W = x.shape
        assert C == 3
        out = torch.empty((H, W, 3), dtype=torch.uint8, device=x.device)
        self.lib.call("zt_quantize_u8_hwc", x, out, H, W, int(mode), self._s(x))
        return out

    def ingest_u8(self, u8, out=None, size=(1920, 1080)):
        """Decoded frame, uint8 [H0,W0,3] (or [1,H0,W0,3]) on the device -> fp32 [1,3,H,W] in [0,1]: the reference loader's
        `im.resize(size)` (PIL BICUBIC, 8-bit two-pass; skipped when the frame already has that size, as PIL does) followed by
        `ToTensor()` (multi_read_data.py:127-132), bit-identical to the host libraries.  size = (W, H) like PIL; None = keep."""
        from . import ingest
        if u8.dim() == 4:
            assert u8.shape[0] == 1
            u8 = u8[0]
        assert u8.dtype == torch.uint8 and u8.dim() == 3 and u8.shape[2] == 3 and u8.is_contiguous()
        dev, s = u8.device, self._s(u8)
        H0, W0 = int(u8.shape[0]), int(u8.shape[1])
        W, H = (W0, H0) if size is None else size
        cache = self.__dict__.setdefault("_ingest_tables", {})

        def tables(n_in, n_out):
            key = (n_in, n_out, str(dev))
            if key not in cache:
                coef, bounds, ks = ingest.pil_bicubic_tables(n_in, n_out)
                cache[key] = (torch.from_numpy(coef).to(dev), torch.from_numpy(bounds).to(dev), ks)
            return cache[key]
        if ("lut", str(dev)) not in cache:
            cache[("lut", str(dev))] = torch.from_numpy(ingest.to_tensor_lut()).to(dev)
        cur = u8
        if W0 != W:                                           # Resample.c: horizontal pass first, into an 8-bit image
            coef, bounds, ks = tables(W0, W)
            nxt = torch.empty((H0, W, 3), dtype=torch.uint8, device=dev)
            self.lib.call("zt_resample_u8_hwc", cur, nxt, H0, W0, H0, W, 1, coef, bounds, ks, s)
            cur = nxt
        if H0 != H:
            coef, bounds, ks = tables(H0, H)
            nxt = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
            self.lib.call("zt_resample_u8_hwc", cur, nxt, H0, W, H, W, 0, coef, bounds, ks, s)
            cur = nxt
        if out is None:
            out = torch.empty((1, 3, H, W), dtype=torch.float32, device=dev)
        assert tuple(out.shape) == (1, 3, H, W) and out.dtype == torch.float32 and out.is_contiguous()
        self.lib.call("zt_u8hwc_to_planar_f32", cur, out, H, W, cache[("lut", str(dev))], s)
        return out

    def psnr_u8(self, a, b):
        """evals.py:83-85: cv2.PSNR of round(a*255), round(b*255) -> python float (inf when identical); one 8-byte read-back."""
        _f32c(a), _f32c(b)
        n = a.numel()
        assert b.numel() == n
        nblk = max(1, min(1024, n // 4096))
        part = torch.empty(nblk, dtype=torch.int64, device=a.device)
        out = torch.empty(1, dtype=torch.int64, device=a.device)
        self.lib.call("zt_sqdiff_u8_f32", a, b, n, part, nblk, out, self._s(a))
        import math
        sq = int(out.item())
        return float("inf") if sq == 0 else 10.0 * math.log10(255.0 ** 2 * n / sq)

    # ---- convolution family (zt_conv.hip) ---------------------------------------------------------------------
    def repack_weight(self, w, ldw=None, co_off=0, transpose_flip=False, out=None):
        """torch [Cout,Cin,KH,KW] -> device layout [KH*KW, Cin', ldw]."""
        _f32c(w)
        Cout, Cin, KH, KW = w.shape
        n_out, n_in = (Cin, Cout) if transpose_flip else (Cout, Cin)
        if ldw is None:
            ldw = (n_out + 15) // 16 * 16
        if out is None:
            out = torch.zeros((KH * KW, n_in, ldw), dtype=torch.float32, device=w.device)
        self.lib.call("zt_repack_conv_weight_f32", w, out, Cout, Cin, KH, KW, ldw, co_off, int(transpose_flip), self._s(w))
        return out

    def conv2d(self, x, wdev, bias, Cout, KH, KW, stride=1, pad=(0, 0), act=None, alpha=1.0, x2=None, out=None,
               out_planar=False, aux=None, epi=0, w_coff=0, out2=None, esplit=0):
        """x: CV/tensor NHWC; optional x2 (CV) supplies channels >= x.C.  out: CV (nhwc) or planar tensor [N,Cout,Ho,Wo]."""
        x = _cv(x)
        Cin, csplit, ldx2, x2p = x.C, 0, 0, None
        if x2 is not None:
            x2 = _cv(x2)
            csplit, Cin, ldx2, x2p = x.C, x.C + x2.C, x2.ld, x2.ptr
        assert wdev.shape[1] == Cin and wdev.shape[0] == KH * KW, (wdev.shape, Cin, KH, KW)
        ldw = wdev.shape[2]
        Ho = (x.H + 2 * pad[0] - KH) // stride + 1
        Wo = (x.W + 2 * pad[1] - KW) // stride + 1
        dev = x.t.device
        if out_planar:
            if out is None:
                out = torch.empty((x.N, Cout, Ho, Wo), dtype=torch.float32, device=dev)
            yptr, ldy = out.data_ptr(), out.stride(1)
        else:
            if out is None:
                out = torch.empty((x.N, Ho, Wo, (Cout + 3) // 4 * 4), dtype=torch.float32, device=dev)
                if out.shape[-1] != Cout:
                    out.zero_()
            o = _cv(out)
            assert (o.N, o.H, o.W) == (x.N, Ho, Wo) and o.C >= Cout
            yptr, ldy = o.ptr, o.ld
        auxp, ldaux = None, 0
        if epi:
            av = _cv(aux)
            auxp, ldaux = av.ptr, av.ld
        tok = self._ev_begin(self.profile["match"].get((KH, KW, stride, Cin, Cout, x.H, x.W))) if self.profile else None
        if epi >= 4:            # fused SepConvGRU epilogues
            o2 = _cv(out2) if out2 is not None else None
            self.lib.call("zt_conv2d_nhwc_f32_ex", x.ptr, x2p, csplit, x.ld, ldx2, x.N, x.H, x.W, Cin, wdev.data_ptr() + 4 * w_coff, ldw, bias,
                          yptr, ldy, int(out_planar), Cout, KH, KW, stride, pad[0], pad[1], ACT[act], float(alpha), auxp, ldaux, epi,
                          o2.ptr if o2 else None, o2.ld if o2 else 0, esplit, self._s(x.t))
        else:
            self.lib.call("zt_conv2d_nhwc_f32", x.ptr, x2p, csplit, x.ld, ldx2, x.N, x.H, x.W, Cin, wdev.data_ptr() + 4 * w_coff, ldw, bias,
                          yptr, ldy, int(out_planar), Cout, KH, KW, stride, pad[0], pad[1], ACT[act], float(alpha), auxp, ldaux, epi,
                          self._s(x.t))
        self._ev_end(tok)
        return out

    def slab(self, dev, nbytes=96 << 20):
        key = (dev, nbytes)
        if key not in self._slab:
            self._slab[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        return self._slab[key]

    def conv2d_wgrad(self, x, dz, Cout, KH, KW, grad_w, accumulate=False, slab=None, grad_b=None):
        """grad_w [Cout,Cin,KH,KW] (+)= wgrad of a stride-1 same conv; x, dz: CV/tensor NHWC (N == 1)."""
        x, dz = _cv(x), _cv(dz)
        assert x.N == 1 and (x.H, x.W) == (dz.H, dz.W) and dz.C >= Cout
        assert tuple(grad_w.shape) == (Cout, x.C, KH, KW) and grad_w.is_contiguous()
        slab = self.slab(x.t.device) if slab is None else slab
        self.lib.call("zt_conv2d_wgrad_nhwc_f32", x.ptr, x.ld, dz.ptr, dz.ld, x.H, x.W, x.C, Cout, KH, KW, slab,
                      slab.numel() * 4, grad_w, grad_b, int(accumulate), self._s(x.t))
        return grad_w

    # ---- normalisation (zt_norm.hip) --------------------------------------------------------------------------
    def _nblk(self, HW):
        # one partial row per workgroup; 256 = one workgroup per CU.  More rows only lengthen the finalize kernel's serial sum
        # (10 us with 900 rows on the 180 x 320 RAFT maps, 4 us with 256) without making the statistics pass any faster.
        return max(1, min(256 if HW < (1 << 20) else _NBLK_BIG, HW // 64))

    def chan_stats(self, x, nblk=None):
        """-> partial [N, nblk, 2, C] (sum, sum of squares) for a CV/tensor NHWC."""
        x = _cv(x)
        HW = x.H * x.W
        nblk = self._nblk(HW) if nblk is None else nblk
        part = torch.empty((x.N, nblk, 2, x.C), dtype=torch.float32, device=x.t.device)
        self.lib.call("zt_chan_stats_nhwc", x.ptr, _dt(x.t), x.ld, x.N, HW, x.C, nblk, part, self._s(x.t))
        return part

    def instance_norm_stats(self, x, eps=1e-5):
        """InstanceNorm scale/shift [N, C] of a CV/tensor NHWC in one launch: the statistics kernel's last workgroup per sample
        does the finalize (zt_norm.hip instnorm_tail).  Ticket counters come from a zero-initialised pool and return to zero."""
        x = _cv(x)
        HW = x.H * x.W
        nblk = self._nblk(HW)
        dev = x.t.device
        pool = self._tickets.get(dev)
        if pool is None:
            pool = self._tickets[dev] = [torch.zeros(4096, dtype=torch.int32, device=dev), 0]
        if pool[1] + x.N > 4096:
            pool[1] = 0
        tick = pool[0][pool[1]:pool[1] + x.N]
        pool[1] += x.N
        part = torch.empty((x.N, nblk, 2, x.C), dtype=torch.float32, device=dev)
        scale = torch.empty((x.N, x.C), dtype=torch.float32, device=dev)
        shift = torch.empty_like(scale)
        self.lib.call("zt_instance_norm_stats", x.ptr, _dt(x.t), x.ld, x.N, HW, x.C, nblk, part, float(eps), tick, scale, shift,
                      self._s(x.t))
        return scale, shift

    def norm_finalize(self, part, N, C, count, mode, gamma=None, beta=None, rm=None, rv=None, nbt=None, momentum=0.1,
                      eps=1e-5, dev=None):
        dev = part.device if part is not None else dev
        scale = torch.empty((N, C), dtype=torch.float32, device=dev)
        shift, mean, rstd = torch.empty_like(scale), torch.empty_like(scale), torch.empty_like(scale)
        nblk = part.shape[1] if part is not None else 0
        self.lib.call("zt_norm_finalize_f32", part, nblk, N, C, count, eps, mode, gamma, beta, rm, rv, nbt, momentum,
                      scale, shift, mean, rstd, current_stream(dev))
        return scale, shift, mean, rstd

    def norm_apply(self, x, scale, shift, out=None, res=None, inner_relu=False, outer_relu=False):
        x = _cv(x)
        if out is None:
            out = torch.empty((x.N, x.H, x.W, x.C), dtype=x.t.dtype, device=x.t.device)
        o = _cv(out)
        rp, ldr = (None, 0) if res is None else (_cv(res).ptr, _cv(res).ld)
        assert o.t.dtype == x.t.dtype and (res is None or _cv(res).t.dtype == x.t.dtype)
        self.lib.call("zt_norm_apply_nhwc", x.ptr, _dt(x.t), x.ld, scale, shift, rp, ldr, o.ptr, o.ld, x.N, x.H * x.W, x.C,
                      int(inner_relu), int(outer_relu), self._s(x.t))
        return out

    def partial_reduce(self, part, nblk, stride, n, out=None, accumulate=False, out2=None):
        self.lib.call("zt_partial_reduce_f32", part, nblk, stride, n, out, int(accumulate), out2, self._s(part))

    def bn_relu_bwd(self, dy, z, scale, shift, mean, rstd, dgamma, dbeta, out=None, eval_mode=False, part=None):
        """backward of ReLU(BN(z)) for N == 1 (train-mode batch statistics, or eval_mode: running statistics are constants);
        accumulates dgamma/dbeta; returns dz (NHWC).  part: the reduce pass's partials as produced by the data-gradient kernel that
        wrote dy (`conv3x3_dgrad_bn_sums_bf16`: [nblk][2][C] = (sum g, sum g (z - mean))) -- the separate reduce launch is skipped."""
        dy, z = _cv(dy), _cv(z)
        HW, C = z.H * z.W, z.C
        assert dy.t.dtype == z.t.dtype
        sums = torch.empty((2, C), dtype=torch.float32, device=z.t.device)
        if part is not None:
            self.lib.call("zt_bn_bwd_sums_centered_f32", part, part.shape[0], C, rstd, dbeta, dgamma, sums, self._s(z.t))
        else:
            nblk = self._nblk(HW)
            part = torch.empty((nblk, 2, C), dtype=torch.float32, device=z.t.device)
            self.lib.call("zt_bn_bwd_reduce", dy.ptr, _dt(z.t), dy.ld, z.ptr, z.ld, scale, shift, mean, rstd, HW, C, nblk, part, self._s(z.t))
            self.lib.call("zt_bn_bwd_sums_f32", part, nblk, C, dbeta, dgamma, sums, self._s(z.t))
        if out is None:
            out = torch.empty((1, z.H, z.W, C), dtype=z.t.dtype, device=z.t.device)
        o = _cv(out)
        self.lib.call("zt_bn_bwd_apply", dy.ptr, _dt(z.t), dy.ld, z.ptr, z.ld, scale, shift, mean, rstd, sums, o.ptr, o.ld, HW, C,
                      int(eval_mode), self._s(z.t))
        return out

    # ---- RAFT specific (zt_raft.hip) --------------------------------------------------------------------------
    def resize_bilinear(self, x, h, w, mul=1.0):
        _, C, H, W = x.shape
        out = torch.empty((1, C, h, w), dtype=torch.float32, device=x.device)
        self.lib.call("zt_resize_bilinear_f32", x, out, C, H, W, h, w, float(mul), self._s(x))
        return out

    def equalize_prepare(self, x255):
        """x255: [1,C,h,w] float in [0,255] -> (uint8 truncation [C,hw], hist int32 [C,256], lut int32 [C,256])."""
        _, C, h, w = x255.shape
        q = torch.empty((C, h * w), dtype=torch.uint8, device=x255.device)
        hist = torch.empty((C, 256), dtype=torch.int32, device=x255.device)
        lut = torch.empty((C, 256), dtype=torch.int32, device=x255.device)
        self.lib.call("zt_equalize_prepare_u8", x255, q, hist, lut, C, h * w, self._s(x255))
        return q, hist, lut

    def raft_pack_input(self, img1, q2, lut, h, w, dtype=torch.float32):
        Hp, Wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
        ld = 8 if dtype == torch.bfloat16 else 4
        out = torch.empty((2, Hp, Wp, ld), dtype=dtype, device=img1.device)
        self.lib.call("zt_raft_pack_input", img1, q2, lut, out, _dt(out), ld, h, w, Hp, Wp, self._s(img1))
        return out

    def raft_stem_weight_bf16(self, w):
        """torch [64,3,7,7] fp32 -> [7,64,64] bf16 (k = kx*8 + c) for raft_stem_bf16"""
        _f32c(w)
        assert tuple(w.shape) == (64, 3, 7, 7)
        out = torch.empty((7, 64, 64), dtype=torch.bfloat16, device=w.device)
        self.lib.call("zt_repack_stem_weight_bf16", w, out, self._s(w))
        return out

    def raft_stem_bf16(self, x, wstem, bias, relu=False):
        """x: nhwc [N,H,W,8] bf16 (channels 3..7 zero) -> conv7x7 s2 p3 + bias (+ ReLU): nhwc [N,H/2,W/2,64] bf16 (extractor.py:120)"""
        assert x.dtype == torch.bfloat16 and x.shape[-1] == 8 and x.is_contiguous()
        N, H, W, _ = x.shape
        out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, 64), dtype=torch.bfloat16, device=x.device)
        self.lib.call("zt_raft_stem_conv_bf16", x, N, H, W, wstem, bias, out, 64, int(relu), self._s(x))
        return out

    def corr_pyramid(self, corr0, h, w):
        """corr0: NHWC [1,h,w,ld>=h*w] level 0 -> [level1, level2, level3] tensors [npx, hl, wl]."""
        npx, ld = h * w, corr0.shape[-1]
        levels, src, hin, win, ldin = [], corr0, h, w, ld
        for _ in range(3):
            dst = torch.empty((npx, hin // 2, win // 2), dtype=torch.float32, device=corr0.device)
            self.lib.call("zt_corr_pool_f32", src, dst, npx, hin, win, ldin, self._s(corr0))
            levels.append(dst)
            src, hin, win = dst, hin // 2, win // 2
            ldin = hin * win
        return levels

    def corr_volume_pyramid_bf16(self, fmap1, fmap2, h, w, alpha):
        """fmap1 / fmap2: bf16 [.., 256]-channel NHWC feature maps of the two frames ([npx][ld] views) -> (corr0 [1,h,w,ld0] fp32,
        [level1, level2, level3]) in one launch (corr.py:13-27, 52-60)."""
        npx = h * w
        ld0 = (npx + 3) // 4 * 4
        dev = fmap1.device
        corr0 = torch.empty((1, h, w, ld0), dtype=torch.float32, device=dev)
        dims = [(h // 2, w // 2), (h // 4, w // 4), (h // 8, w // 8)]
        levels = [torch.empty((npx, a, b), dtype=torch.float32, device=dev) for a, b in dims]
        tok = self._ev_begin(self.profile["match"].get((1, 1, 1, 256, npx, h, w))) if self.profile else None
        self.lib.call("zt_corr_volume_pyramid_bf16", fmap1, fmap1.shape[-1], fmap2, fmap2.shape[-1], h, w, float(alpha), corr0, ld0,
                      levels[0], levels[1], levels[2], self._s(fmap1))
        self._ev_end(tok)
        return corr0, levels

    def corr_lookup(self, corr0, levels, h, w, coords, out=None):
        npx = h * w
        if out is None:
            out = torch.empty((1, h, w, 324), dtype=torch.float32, device=corr0.device)
        self.lib.call("zt_corr_lookup", corr0, levels[0], levels[1], levels[2], h, w, corr0.shape[-1], coords, out, _dt(out),
                      out.shape[-1], npx, self._s(corr0))
        return out

    def corr_lookup_step(self, corr0, levels, h, w, coords, out, delta, coords_out, f4, fhx_ptr, ldfhx, fin):
        """lookup at coords + delta with the previous iteration's flow bookkeeping folded in (zt_corr_lookup_step)."""
        self.lib.call("zt_corr_lookup_step", corr0, levels[0], levels[1], levels[2], h, w, corr0.shape[-1], coords, out, _dt(out),
                      out.shape[-1], h * w, delta, 0 if delta is None else delta.shape[-1], coords_out, f4, f4.shape[-1], fhx_ptr, ldfhx,
                      fin, fin.shape[-1], self._s(corr0))
        return out

    # ---- bf16 throughput mode of the convolution family ---------------------------------------------------------
    def repack_weight_bf16(self, w, transpose_flip=False, out=None, co_off=0):
        """torch fp32 [Cout,Cin,KH,KW] -> bf16 [KH*KW, CoutP16, ldk8] (input channel fastest); zero padded."""
        _f32c(w)
        Cout, Cin, KH, KW = w.shape
        n_out, n_in = (Cin, Cout) if transpose_flip else (Cout, Cin)
        if out is None:
            CoutP, ldk = (n_out + 15) // 16 * 16, (n_in + 7) // 8 * 8
            out = torch.zeros((KH * KW, CoutP, ldk), dtype=torch.bfloat16, device=w.device)
        CoutP, ldk = out.shape[1], out.shape[2]
        self.lib.call("zt_repack_conv_weight_bf16", w, out, Cout, Cin, KH, KW, CoutP, ldk, co_off, int(transpose_flip), self._s(w))
        return out

    def conv2d_bf16(self, x, wdev, bias, Cout, KH, KW, pad=(0, 0), act=None, alpha=1.0, out=None, out_planar=False, aux=None, epi=0,
                    stride=1, x2=None, out_f32=False, w_roff=0, variant=0, out2=None, esplit=0):
        """x (and optional x2 for channels >= x.C): CV over bf16 NHWC buffers.  out: bf16 NHWC (default), fp32 NHWC (out_f32)
        or fp32 planar [N,Cout,Ho,Wo] (out_planar).  w_roff: first output-channel row of wdev to use."""
        x = _cv(x)
        assert x.t.dtype == torch.bfloat16 and wdev.dtype == torch.bfloat16 and wdev.shape[0] == KH * KW
        Cin, csplit, ldx2, x2p = x.C, 0, 0, None
        if x2 is not None:
            x2 = _cv(x2)
            assert x2.t.dtype == torch.bfloat16
            csplit, Cin, ldx2, x2p = x.C, x.C + x2.C, x2.ld, x2.ptr
        CoutP, ldk = wdev.shape[1] - w_roff, wdev.shape[2]
        assert ldk >= Cin and CoutP >= Cout and (KH * KW == 1 or w_roff == 0)
        Ho = (x.H + 2 * pad[0] - KH) // stride + 1
        Wo = (x.W + 2 * pad[1] - KW) // stride + 1
        dev = x.t.device
        if out_planar:
            if out is None:
                out = torch.empty((x.N, Cout, Ho, Wo), dtype=torch.float32, device=dev)
            yptr, ldy, mode = out.data_ptr(), out.stride(1), 1
        else:
            odt = torch.float32 if out_f32 else torch.bfloat16
            gran = 4 if out_f32 else 8
            if out is None:
                ld = (Cout + gran - 1) // gran * gran
                out = torch.empty((x.N, Ho, Wo, ld), dtype=odt, device=dev)
                if ld != Cout:
                    out.zero_()
            o = _cv(out)
            assert o.t.dtype == odt and (o.N, o.H, o.W) == (x.N, Ho, Wo) and o.C >= Cout
            yptr, ldy, mode = o.ptr, o.ld, (2 if out_f32 else 0)
        auxp, ldaux = None, 0
        if epi:
            av = _cv(aux)
            assert av.t.dtype == torch.bfloat16
            auxp, ldaux = av.ptr, av.ld
        tok = self._ev_begin(self.profile["match"].get((KH, KW, stride, Cin, Cout, x.H, x.W))) if self.profile else None
        if epi >= 4:            # fused SepConvGRU epilogues (4, 5); ResidualBlock tail relu(act(conv) + aux) (6)
            o2 = _cv(out2) if out2 is not None else None
            assert o2 is None or o2.t.dtype == torch.bfloat16
            self.lib.call("zt_conv2d_nhwc_bf16_ex", x.ptr, x2p, csplit, x.ld, ldx2, x.N, x.H, x.W, Cin,
                          wdev.data_ptr() + 2 * w_roff * ldk, CoutP, ldk, bias, yptr, ldy, mode, Cout, KH, KW, stride, pad[0], pad[1],
                          ACT[act], float(alpha), auxp, ldaux, epi, o2.ptr if o2 else None, o2.ld if o2 else 0, esplit, self._s(x.t))
            self._ev_end(tok)
            return out
        self.lib.call("zt_conv2d_nhwc_bf16_variant", x.ptr, x2p, csplit, x.ld, ldx2, x.N, x.H, x.W, Cin,
                      wdev.data_ptr() + 2 * w_roff * ldk, CoutP, ldk, bias, yptr, ldy, mode, Cout, KH, KW, stride, pad[0], pad[1],
                      ACT[act], float(alpha), auxp, ldaux, epi, variant, self._s(x.t))
        self._ev_end(tok)
        return out

    def conv_pair_bf16(self, xA, wA, bA, CoutA, KA, outA, xB, wB, bB, CoutB, KB, outB, act=None):
        """Two independent stride-1 'same' convolutions (square kernels KA / KB) over the same map in ONE launch
        (zt_conv2d_pair_nhwc_bf16): bf16 NHWC in and out, y = act(conv + bias)."""
        xA, xB, oA, oB = _cv(xA), _cv(xB), _cv(outA), _cv(outB)
        assert (xA.N, xA.H, xA.W) == (xB.N, xB.H, xB.W) == (oA.N, oA.H, oA.W) == (oB.N, oB.H, oB.W)
        assert all(t.t.dtype == torch.bfloat16 for t in (xA, xB, oA, oB)) and wA.shape[0] == KA * KA and wB.shape[0] == KB * KB
        assert wA.shape[2] >= xA.C and wB.shape[2] >= xB.C and wA.shape[1] >= CoutA and wB.shape[1] >= CoutB and oA.C >= CoutA and oB.C >= CoutB
        self.lib.call("zt_conv2d_pair_nhwc_bf16", xA.ptr, xA.ld, xA.C, wA, wA.shape[1], wA.shape[2], bA, oA.ptr, oA.ld, CoutA, KA,
                      xB.ptr, xB.ld, xB.C, wB, wB.shape[1], wB.shape[2], bB, oB.ptr, oB.ld, CoutB, KB, xA.N, xA.H, xA.W, ACT[act], self._s(xA.t))

    # ---- deferred, batched weight gradients / weight repacks (bf16 mode) -----------------------------------------------
    def wgrad_partial_bf16(self, x, dz, Cout, K, slab, slab_off, relu_mask=None):
        """Append the per-workgroup slabs of one weight-gradient call to `slab` (fp32 tensor) at float offset slab_off.
        -> number of slabs written."""
        import ctypes
        x, dz = _cv(x), _cv(dz)
        assert x.t.dtype == torch.bfloat16 and dz.t.dtype == torch.bfloat16 and x.N == 1 and (x.H, x.W) == (dz.H, dz.W) and dz.C >= Cout
        mk = _cv(relu_mask) if relu_mask is not None else None
        n = ctypes.c_int(0)
        tok = self._ev_begin(self.profile["match"].get(("wgrad", K, x.C, Cout, x.H, x.W))) if self.profile else None
        self.lib.call("zt_conv2d_wgrad_partial_bf16", x.ptr, x.ld, dz.ptr, dz.ld, x.H, x.W, x.C, Cout, K, K, slab.data_ptr() + 4 * slab_off,
                      (slab.numel() - slab_off) * 4, mk.ptr if mk else None, mk.ld if mk else 0, ctypes.byref(n), self._s(x.t))
        self._ev_end(tok)
        return n.value

    def thin1x1_bwd_bf16(self, dr, Cdr, wT, a2, slab, slab_off):
        """Denoise_1/2 conv3 backward in one pass (zt_thin1x1_bwd_bf16): dr [1,H,W,8] bf16 gradient of the 1x1 output, wT the
        transposed-weight tensor of the data gradient ([1][48][8] bf16), a2 [1,H,W,>=48] bf16.  -> (dz2 [1,H,W,48] bf16, slabs written)."""
        import ctypes
        dr, a2 = _cv(dr), _cv(a2)
        assert dr.t.dtype == a2.t.dtype == wT.dtype == torch.bfloat16 and dr.ld == 8 and dr.N == 1 and (dr.H, dr.W) == (a2.H, a2.W)
        assert tuple(wT.shape[-2:]) == (48, 8) and a2.C >= 48
        dz = torch.empty((1, a2.H, a2.W, 48), dtype=torch.bfloat16, device=a2.t.device)
        n = ctypes.c_int(0)
        self.lib.call("zt_thin1x1_bwd_bf16", dr.ptr, Cdr, wT, a2.ptr, a2.ld, dz, 48, a2.H * a2.W, slab.data_ptr() + 4 * slab_off,
                      (slab.numel() - slab_off) * 4, ctypes.byref(n), self._s(a2.t))
        return dz, n.value

    @staticmethod
    def wgrad_slab_floats(Cin, Cout, K):
        c16 = lambda c: (c + 15) // 16 * 16
        return K * K * c16(Cin) * c16(Cout) + c16(Cout)

    def wgrad_reduce_multi(self, segs, accumulate=True):
        """segs: [(slab tensor, nslab, Cin, Cout, K, grad_w, grad_b)] -- one launch reduces every layer."""
        import ctypes
        n = len(segs)
        P, I = ctypes.c_void_p * n, ctypes.c_int * n
        dev = segs[0][0].device
        self.lib.call("zt_wgrad_reduce_multi_f32", n, P(*[s[0].data_ptr() for s in segs]), I(*[s[1] for s in segs]), I(*[s[2] for s in segs]),
                      I(*[s[3] for s in segs]), I(*[s[4] for s in segs]), P(*[s[5].data_ptr() for s in segs]),
                      P(*[(s[6].data_ptr() if s[6] is not None else None) for s in segs]), int(accumulate), current_stream(dev))

    def repack_weights_bf16_multi(self, entries):
        """entries: [(w fp32 [Cout,Cin,K,K], out bf16 [K*K,CoutP,ldk], transpose_flip)] -- all repacks of a step in one launch."""
        import ctypes
        n = len(entries)
        P, I = ctypes.c_void_p * n, ctypes.c_int * n
        dev = entries[0][0].device
        self.lib.call("zt_repack_conv_weights_bf16_multi", n, P(*[e[0].data_ptr() for e in entries]), P(*[e[1].data_ptr() for e in entries]),
                      I(*[e[0].shape[0] for e in entries]), I(*[e[0].shape[1] for e in entries]), I(*[e[0].shape[2] for e in entries]),
                      I(*[e[1].shape[1] for e in entries]), I(*[e[1].shape[2] for e in entries]), I(*[int(e[2]) for e in entries]),
                      current_stream(dev))

    def conv3x3_dgrad_bn_sums_bf16(self, dz, wT, res, zprev, scale, shift, mean):
        """Enhancer block backward: df = conv3x3^T(dz) + res and the BatchNorm-backward partial sums of the block below (whose
        pre-activation is zprev, BatchNorm constants scale / shift / mean) in one pass.  -> (df bf16 [1,H,W,64], part [512,2,64])."""
        dz, res, zp = _cv(dz), _cv(res), _cv(zprev)
        assert dz.t.dtype == res.t.dtype == zp.t.dtype == wT.dtype == torch.bfloat16 and dz.N == 1 and dz.C == 64 and zp.C == 64 and res.C == 64
        dev = dz.t.device
        df = torch.empty((1, dz.H, dz.W, 64), dtype=torch.bfloat16, device=dev)
        part = torch.empty((512, 2, 64), dtype=torch.float32, device=dev)
        tok = self._ev_begin(self.profile["match"].get((3, 3, 1, 64, 64, dz.H, dz.W))) if self.profile else None
        self.lib.call("zt_conv3x3_dgrad_bn_sums_bf16", dz.ptr, dz.ld, dz.H, dz.W, wT, wT.shape[1], wT.shape[2], df, 64, res.ptr, res.ld,
                      zp.ptr, zp.ld, scale, shift, mean, part, 512, self._s(dz.t))
        self._ev_end(tok)
        return df, part

    def conv3x3_bn_stats_bf16(self, x, wdev, bias, Cout):
        """y = conv3x3(x) + bias (bf16 nhwc) together with the BatchNorm statistics of y: -> (y, partial [1, 512, 2, Cout])."""
        x = _cv(x)
        assert x.t.dtype == torch.bfloat16 and wdev.dtype == torch.bfloat16 and x.N == 1 and wdev.shape[0] == 9
        out = torch.empty((1, x.H, x.W, Cout), dtype=torch.bfloat16, device=x.t.device)
        part = torch.empty((1, 512, 2, Cout), dtype=torch.float32, device=x.t.device)
        tok = self._ev_begin(self.profile["match"].get((3, 3, 1, x.C, Cout, x.H, x.W))) if self.profile else None
        self.lib.call("zt_conv3x3_bn_stats_bf16", x.ptr, x.ld, x.H, x.W, x.C, wdev, wdev.shape[1], wdev.shape[2], bias, out, Cout, Cout, part, 512,
                      self._s(x.t))
        self._ev_end(tok)
        return out, part

    def conv2d_wgrad_bf16(self, x, dz, Cout, KH, KW, grad_w, accumulate=False, slab=None, grad_b=None, relu_mask=None):
        """relu_mask: activation tensor (nhwc bf16) of the ReLU that follows the layer; dz is used as dz * [relu_mask > 0]."""
        x, dz = _cv(x), _cv(dz)
        mk = _cv(relu_mask) if relu_mask is not None else None
        assert x.t.dtype == torch.bfloat16 and dz.t.dtype == torch.bfloat16
        assert x.N == 1 and (x.H, x.W) == (dz.H, dz.W) and dz.C >= Cout
        assert tuple(grad_w.shape) == (Cout, x.C, KH, KW) and grad_w.is_contiguous() and grad_w.dtype == torch.float32
        slab = self.slab(x.t.device) if slab is None else slab
        self.lib.call("zt_conv2d_wgrad_nhwc_bf16", x.ptr, x.ld, dz.ptr, dz.ld, x.H, x.W, x.C, Cout, KH, KW, slab, slab.numel() * 4,
                      grad_w, grad_b, int(accumulate), mk.ptr if mk else None, mk.ld if mk else 0, self._s(x.t))
        return grad_w
