"""Static execution plan of the Zero-TIG hot path over the HIP kernels.

`Engine.forward()`     = Network.forward            (reference model/model.py:144-203)
`Engine.loss_grads()`  = LossFunction.forward       (loss.py:23-78) fused with its gradient, then the hand-derived
                         backward through Denoise_2 / Enhancer / Denoise_1 (what autograd does at train.py:129)
All compute is in libzerotig_hip.so; torch only owns the buffers.  The graph is static, so the backward pass is written
out explicitly instead of being recorded by autograd: activations needed later are kept in `self.sv`.
"""
import torch

from .ops import CV

TERM_NAMES = ["enh_s2", "enh_norm", "smooth", "tv", "res1_a", "res1_b", "res1_c", "res1_d", "res2_a", "res2_b", "res2_c",
              "res2_d", "color", "ill", "inter_a", "inter_b", "var"]

import os

_FUSED_BN_BWD = os.environ.get("ZT_FUSED_BN_BWD", "1") == "1"          # A/B knob: 0 = separate BatchNorm-backward reduce pass per block
_FUSED_THIN_BWD = os.environ.get("ZT_FUSED_THIN_BWD", "1") == "1"      # A/B knob: 0 = separate 1x1 data-gradient and weight-gradient launches

D1 = "denoise_1"
D2 = "denoise_2"


class Engine:
    def __init__(self, ops, params, buffers, is_WB=False, device=None, precision="fp32"):
        """params: {name: tensor} of the 20 trainable tensors (torch layout, on device); buffers: BN running stats.
        precision: "fp32" (parity mode: exact-fp32 MFMA, fp32 activations) or "bf16" (throughput mode: bf16 activations and
        weights in HBM, fp32 accumulation / statistics / loss / optimizer)."""
        assert precision in ("fp32", "bf16")
        self.precision = precision
        self.adt = torch.bfloat16 if precision == "bf16" else torch.float32      # NHWC activation storage type
        self.dt = 1 if precision == "bf16" else 0
        self.lda = 8 if precision == "bf16" else 4                              # channel-stride granule of thin NHWC buffers
        self.ops, self.lib = ops, ops.lib
        self.p, self.buf = params, buffers
        self.is_WB = is_WB
        self.dev = device if device is not None else next(iter(params.values())).device
        self.sv = {}
        self.wd = {}
        self._wg, self._wg_slab = {}, {}
        # tuning hook (off): weight gradients on a second HIP stream, forked per call and joined before the batched slab
        # reduction.  Nothing downstream of a layer's wgrad depends on it until that reduction, but the overlap does not pay:
        # 10.47 ms/step with the side stream against 10.27 ms without (same box, gpurun_out r03a) -- the big kernels of the two
        # streams do not co-reside (LDS / VGPR footprints) and each fork adds a dependency edge.
        import os
        self._wg_side = self.dev.type == "cuda" and os.environ.get("ZT_WGRAD_STREAM", "0") == "1"
        self._wg_stream, self._wg_keep, self._wg_forked = None, [], False
        self.training = True

    # ------------------------------------------------------------------------------------------------ helpers
    def _new(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.dev)

    def _zeros(self, *shape):
        return torch.zeros(shape, dtype=torch.float32, device=self.dev)

    def _stream(self):
        from .lib import current_stream
        return current_stream(self.dev)

    def repack_weights(self):
        """torch-layout parameters -> device conv layouts (forward and data-gradient operators)."""
        o, p, wd = self.ops, self.p, self.wd
        layers = (D1 + ".conv1", D1 + ".conv2", D1 + ".conv3", D2 + ".conv1", D2 + ".conv2", D2 + ".conv3",
                  "enhance.in_conv.0", "enhance.conv.0", "enhance.out_conv.0")
        if self.dt and all(pre in wd for pre in layers):          # steady state, bf16: every repack of the step in ONE launch
            ent = []
            for pre in layers:
                w = p[pre + ".weight"]
                ent.append((w, wd[pre], False))
                if pre + "/T" in wd:
                    ent.append((w, wd[pre + "/T"], True))
            o.repack_weights_bf16_multi(ent)
            return
        rp = o.repack_weight_bf16 if self.dt else o.repack_weight
        for pre in (D1 + ".conv1", D1 + ".conv2", D1 + ".conv3", D2 + ".conv1", D2 + ".conv2", D2 + ".conv3",
                    "enhance.in_conv.0", "enhance.conv.0", "enhance.out_conv.0"):
            w = p[pre + ".weight"]
            wd[pre] = rp(w, out=wd.get(pre))
            if pre not in (D1 + ".conv1", "enhance.in_conv.0"):        # inputs of these need no gradient
                wd[pre + "/T"] = rp(w, transpose_flip=True, out=wd.get(pre + "/T"))

    def _conv(self, x, wkey, bias, cout, k, act=None, out_planar=False, aux=None, epi=0):
        """stride-1 'same' convolution of the enhancement nets in the engine's precision"""
        pad = (k // 2, k // 2)
        if self.dt:
            return self.ops.conv2d_bf16(x, self.wd[wkey], bias, cout, k, k, pad, act, out_planar=out_planar, aux=aux, epi=epi)
        return self.ops.conv2d(x, self.wd[wkey], bias, cout, k, k, 1, pad, act, out_planar=out_planar, aux=aux, epi=epi)

    def _wgrad(self, x, dz, cout, k, pre, relu_mask=None):
        """accumulate d loss / d weight and d loss / d bias (column sums of dz, same pass) of conv `pre`.
        bf16 mode: the call only APPENDS its per-workgroup slabs to the layer's slab region; `_wgrad_flush` reduces every layer
        of the backward pass in one launch (the three invocations of a shared layer are one longer slab list)."""
        if not self.dt:
            self.ops.conv2d_wgrad(x, dz, cout, k, k, self.g[pre + ".weight"], accumulate=True, grad_b=self.g[pre + ".bias"])
            return
        o = self.ops
        xv = x if isinstance(x, CV) else CV(x)
        per = o.wgrad_slab_floats(xv.C, cout, k)
        ent = self._wg.get(pre)
        if ent is None:
            ent = self._wg[pre] = {"n": 0, "Cin": xv.C, "Cout": cout, "K": k, "per": per}
        slab = self._wg_slab.get(pre)
        need = (ent["n"] + 512) * per
        if slab is None or slab.numel() < need:                   # grows to its steady-state size during the first (eager) steps
            new = torch.empty(max(need, 3 * 512 * per if slab is None else need), dtype=torch.float32, device=self.dev)
            if slab is not None and ent["n"]:
                new[:ent["n"] * per].copy_(slab[:ent["n"] * per])
            slab = self._wg_slab[pre] = new
        if self._wg_side:
            main = torch.cuda.current_stream(self.dev)
            if self._wg_stream is None:
                self._wg_stream = torch.cuda.Stream(device=self.dev)
            self._wg_stream.wait_stream(main)                   # operands (x, dz, mask) are complete on the main stream
            self._wg_keep.append((xv.t, dz.t if isinstance(dz, CV) else dz, relu_mask))
            with torch.cuda.stream(self._wg_stream):
                ent["n"] += o.wgrad_partial_bf16(xv, dz, cout, k, slab, ent["n"] * per, relu_mask=relu_mask)
            self._wg_forked = True
            return
        ent["n"] += o.wgrad_partial_bf16(xv, dz, cout, k, slab, ent["n"] * per, relu_mask=relu_mask)

    def _thin_bwd(self, a2, dr, cout, pre):
        """Denoise conv3 backward (bf16 mode): appends the layer's weight-gradient slabs like `_wgrad` and returns dz2."""
        o = self.ops
        per = o.wgrad_slab_floats(48, cout, 1)
        ent = self._wg.get(pre)
        if ent is None:
            ent = self._wg[pre] = {"n": 0, "Cin": 48, "Cout": cout, "K": 1, "per": per}
        slab = self._wg_slab.get(pre)
        need = (ent["n"] + 512) * per
        if slab is None or slab.numel() < need:
            new = torch.empty(max(need, 3 * 512 * per if slab is None else need), dtype=torch.float32, device=self.dev)
            if slab is not None and ent["n"]:
                new[:ent["n"] * per].copy_(slab[:ent["n"] * per])
            slab = self._wg_slab[pre] = new
        dz2, n = o.thin1x1_bwd_bf16(dr, cout, self.wd[pre + "/T"], a2, slab, ent["n"] * per)
        ent["n"] += n
        return dz2

    def _wgrad_flush(self):
        if self._wg_forked:
            torch.cuda.current_stream(self.dev).wait_stream(self._wg_stream)
            self._wg_forked, self._wg_keep = False, []
        if not self._wg:
            return
        segs = [(self._wg_slab[pre], e["n"], e["Cin"], e["Cout"], e["K"], self.g[pre + ".weight"], self.g[pre + ".bias"])
                for pre, e in self._wg.items()]
        self.ops.wgrad_reduce_multi(segs, accumulate=True)
        self._wg = {}

    def _newa(self, *shape):
        return torch.empty(shape, dtype=self.adt, device=self.dev)

    def _pack(self, ld, HW, srcs, H, W):
        dst = self._newa(1, H, W, ld)
        a = []
        for t in srcs:
            a += [t, t.shape[1]]
        while len(a) < 8:
            a += [None, 0]
        self.lib.call("zt_pack_nhwc", dst, self.dt, ld, HW, *a, self._stream())
        return dst

    # ------------------------------------------------------------------------------------------------ denoisers
    def _denoise_fwd(self, pre, srcs, H, W, cin, cout, key):
        """conv3x3+LReLU, conv3x3+LReLU, conv1x1 on cat(srcs) (model.py:15-44). Returns planar [1,cout,H,W]."""
        p = self.p
        ld = (cin + self.lda - 1) // self.lda * self.lda
        u = self._pack(ld, H * W, srcs, H, W)
        a1 = self._conv(CV(u, 0, cin), pre + ".conv1", p[pre + ".conv1.bias"], 48, 3, "lrelu")
        a2 = self._conv(a1, pre + ".conv2", p[pre + ".conv2.bias"], 48, 3, "lrelu")
        r = self._conv(a2, pre + ".conv3", p[pre + ".conv3.bias"], cout, 1, None, out_planar=True)
        if self.keep:
            self.sv[key] = (u, a1, a2, cin)
        return r

    def _denoise_bwd(self, pre, key, dr, cout, want_input_grad):
        """dr: NHWC gradient of the 1x1 output (first `cout` channels valid).  Accumulates parameter grads; returns the planar
        [1,cin,H,W] gradient of the packed input when requested."""
        u, a1, a2, cin = self.sv[key]
        drv = CV(dr, 0, cout)
        if self.dt and _FUSED_THIN_BWD and tuple(self.wd[pre + ".conv3/T"].shape[-2:]) == (48, 8):
            # conv3 (1x1, 48 -> 3 / 6): data gradient and weight / bias gradient slabs from ONE pass over a2 and dr
            dz2 = self._thin_bwd(a2, dr, cout, pre + ".conv3")
        else:
            self._wgrad(a2, drv, cout, 1, pre + ".conv3")
            dz2 = self._conv(drv, pre + ".conv3/T", None, 48, 1, None, aux=a2, epi=1)
        self._wgrad(a1, dz2, 48, 3, pre + ".conv2")
        dz1 = self._conv(dz2, pre + ".conv2/T", None, 48, 3, None, aux=a1, epi=1)
        self._wgrad(CV(u, 0, cin), dz1, 48, 3, pre + ".conv1")
        if want_input_grad:
            return self._conv(dz1, pre + ".conv1/T", None, cin, 3, None, out_planar=True)
        return None

    # ------------------------------------------------------------------------------------------------ enhancer
    def _enhancer_fwd(self, wpH, wps, L2, H, W):
        """model.py:47-81."""
        o, p, wd, b = self.ops, self.p, self.wd, self.buf
        u = self._pack(16 if self.dt else 12, H * W, [wpH, wps, L2], H, W)
        f = self._conv(CV(u, 0, 9), "enhance.in_conv.0", p["enhance.in_conv.0.bias"], 64, 3, "relu")
        feats, zs, stats = [f], [], []
        for _ in range(3):
            if self.training and self.dt:     # bf16 mode: BatchNorm statistics come out of the convolution's store phase
                z, part = o.conv3x3_bn_stats_bf16(f, wd["enhance.conv.0"], p["enhance.conv.0.bias"], 64)
            else:
                z = self._conv(f, "enhance.conv.0", p["enhance.conv.0.bias"], 64, 3, None)
                part = o.chan_stats(z) if self.training else None
            if self.training:
                st = o.norm_finalize(part, 1, 64, H * W, 1, p["enhance.conv.1.weight"], p["enhance.conv.1.bias"],
                                     b["enhance.conv.1.running_mean"], b["enhance.conv.1.running_var"],
                                     b["enhance.conv.1.num_batches_tracked"], 0.1)
            else:
                st = o.norm_finalize(None, 1, 64, 1, 2, p["enhance.conv.1.weight"], p["enhance.conv.1.bias"],
                                     b["enhance.conv.1.running_mean"], b["enhance.conv.1.running_var"], dev=self.dev)
            f = o.norm_apply(z, st[0], st[1], res=f, inner_relu=True)
            feats.append(f)
            zs.append(z)
            stats.append(st)
        s2 = self._conv(f, "enhance.out_conv.0", p["enhance.out_conv.0.bias"], 3, 3, "sigmoid_clamp", out_planar=True)
        if self.keep:
            self.sv["E"] = (u, feats, zs, stats)
        return s2

    def _enhancer_bwd(self, dO):
        """dO: NHWC4 gradient w.r.t. the out_conv pre-activation."""
        o, wd, g, p = self.ops, self.wd, self.g, self.p
        u, feats, zs, stats = self.sv["E"]
        H, W = feats[0].shape[1], feats[0].shape[2]
        dOv = CV(dO, 0, 3)
        self._wgrad(feats[3], dOv, 3, 3, "enhance.out_conv.0")
        df = self._conv(dOv, "enhance.out_conv.0/T", None, 64, 3, None)
        part = None
        min_tiles = int(os.environ.get("ZT_STATS_FUSE_MIN_TILES", "1024"))      # same size gate as the forward statistics fusion (tests lower it)
        fuse = self.dt and _FUSED_BN_BWD and ((W + 31) // 32) * ((H + 7) // 8) >= min_tiles
        for i in (2, 1, 0):
            sc, sh, mu, rs = stats[i]
            # eval-mode BN (the reference trains epochs >= 1 like this, train.py:138 / SURVEY A-14): running stats are constants
            dz = o.bn_relu_bwd(df, zs[i], sc, sh, mu, rs, g["enhance.conv.1.weight"], g["enhance.conv.1.bias"],
                               eval_mode=not self.training, part=part)
            self._wgrad(feats[i], dz, 64, 3, "enhance.conv.0")
            if fuse and i > 0:
                # the data gradient also reduces the NEXT block's BatchNorm-backward sums (it writes that block's output gradient):
                # one 531 MB read pass less per block
                psc, psh, pmu, _ = stats[i - 1]
                df, part = o.conv3x3_dgrad_bn_sums_bf16(dz, wd["enhance.conv.0/T"], df, zs[i - 1], psc, psh, pmu)
            else:
                df = self._conv(dz, "enhance.conv.0/T", None, 64, 3, None, aux=df, epi=3)
                part = None
        # through the in_conv ReLU (its input needs no gradient, so only the weight gradient consumes the masked df)
        if self.dt:         # bf16: the mask [feats[0] > 0] is applied while the weight-gradient kernel stages df
            self._wgrad(CV(u, 0, 9), df, 64, 3, "enhance.in_conv.0", relu_mask=feats[0])
        else:
            dz0 = self._newa(1, H, W, 64)
            self.lib.call("zt_relu_mask_nhwc", df, self.dt, 64, feats[0], 64, dz0, 64, H * W, 64, self._stream())
            self._wgrad(CV(u, 0, 9), dz0, 64, 3, "enhance.in_conv.0")

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, inp, cache_fn=None, keep=True, skip_unused=False):
        """inp: [1,3,H,W] in [0,1].  cache_fn(L2) -> (warped last_H3, warped last_s3) (model.py:164); None on a new sequence
        (zeros, model.py:155-161).  Returns the 23 outputs in the reference order (Appendix B of SURVEY.md)."""
        o, lib, s = self.ops, self.lib, self._stream()
        _, _, H, W = inp.shape
        assert H % 2 == 0 and W % 2 == 0, "H and W must be even (the reference resizes to 1920x1080)"
        h, w = H // 2, W // 2
        self.keep = keep
        self.sv = {}
        self.repack_weights()
        x, L11, L12 = self._new(1, 3, H, W), self._new(1, 3, h, w), self._new(1, 3, h, w)
        Lq11, Lq12 = self._new(1, 3, h, w), self._new(1, 3, h, w)
        lib.call("zt_prep_input_f32", inp, x, L11, L12, Lq11, Lq12, H, W, s)
        n11 = self._denoise_fwd(D1, [L11], h, w, 3, 3, "D1a")
        n12 = self._denoise_fwd(D1, [L12], h, w, 3, 3, "D1b")
        n = self._denoise_fwd(D1, [x], H, W, 3, 3, "D1c")
        L2, Lp1, Lp2 = self._new(1, 3, H, W), self._new(1, 3, h, w), self._new(1, 3, h, w)
        den1, den2 = self._new(1, 3, h, w), self._new(1, 3, h, w)
        lib.call("zt_d1_tail_f32", x, n, L11, n11, L12, n12, L2, Lp1, Lp2, den1, den2, H, W, s)
        if cache_fn is None:
            wpH, wps = self._zeros(1, 3, H, W), self._zeros(1, 3, H, W)
            wpH1 = wpH2 = wps1 = wps2 = self._zeros(1, 3, h, w)
        else:
            wpH, wps = cache_fn(L2)
            wpH1, wpH2 = o.pair_down(wpH)
            wps1, wps2 = o.pair_down(wps)
        self.last_wp = (wpH, wps)
        s2 = self._enhancer_fwd(wpH, wps, L2, H, W)
        s21, s22, H2 = self._new(1, 3, h, w), self._new(1, 3, h, w), self._new(1, 3, H, W)
        H11, H12, H1 = self._new(1, 3, h, w), self._new(1, 3, h, w), self._new(1, 3, H, W)
        lib.call("zt_post_enh_f32", x, s2, L2, L11, L12, s21, s22, H2, H11, H12, H1, H, W, s)
        r3 = self._denoise_fwd(D2, [wpH1, wps1, H11, s21], h, w, 12, 6, "D2a")
        r4 = self._denoise_fwd(D2, [wpH2, wps2, H12, s22], h, w, 12, 6, "D2b")
        r5 = self._denoise_fwd(D2, [wpH, wps, H2, s2], H, W, 12, 6, "D2c")
        H3p, H4p, H5p = self._new(1, 6, h, w), self._new(1, 6, h, w), self._new(1, 6, H, W)
        lib.call("zt_clamp_sub6_f32", H11, s21, r3, H3p, H3p[:, 3:], h * w, s)
        lib.call("zt_clamp_sub6_f32", H12, s22, r4, H4p, H4p[:, 3:], h * w, s)
        lib.call("zt_clamp_sub6_f32", H2, s2, r5, H5p, H5p[:, 3:], H * W, s)
        H3, s3 = H5p[:, :3], H5p[:, 3:]
        # L_pred1_L_pred2_diff (model.py:194) is returned by forward() but never reaches the loss (loss.py ignores it): the
        # training plan (skip_unused) does not compute it
        m_l = None if skip_unused else o.texture_mask(Lp1, Lp2)
        H3d1, H3d2 = o.pair_down(H3)
        m_h = o.texture_mask(H3d1, H3d2)
        tmp = self._new(1, 3, H, W)
        H2b, H3b = o.blur21(H1, tmp), o.blur21(H3, tmp)
        if keep:
            self.sv.update(inp=inp, x=x, n=n, L11=L11, L12=L12, Lq11=Lq11, Lq12=Lq12, L2=L2, Lp1=Lp1, Lp2=Lp2, den1=den1,
                           den2=den2, s2=s2, s21=s21, s22=s22, H2=H2, H11=H11, H12=H12, r3=r3, r4=r4, r5=r5, H3p=H3p, H4p=H4p,
                           H3=H3, s3=s3, H3d1=H3d1, H3d2=H3d2, m_h=m_h, H2b=H2b, H3b=H3b, HW=(H, W))
        return (Lp1, Lp2, L2, s2, s21, s22, H2, H11, H12, H3p[:, :3], H3p[:, 3:], H4p[:, :3], H4p[:, 3:], H3, s3, H3p, H4p,
                m_l, m_h, H2b, H3b, H3d1, H3d2)

    def forward_infer(self, inp, cache_fn=None):
        """Finetunemodel.forward (model.py:312-340): full-resolution branch, eval-mode BN; new-sequence Denoise_2 temporal
        slots are H2 (model.py:330-332).  Returns (H2, H3, s3)."""
        lib, s = self.lib, self._stream()
        _, _, H, W = inp.shape
        self.keep, self.sv, self.training = False, {}, False
        self.repack_weights()
        n3 = 3 * H * W
        x = self._new(1, 3, H, W)
        lib.call("zt_ew_f32", inp, None, x, 0, 1e-4, 0.0, n3, s)
        n = self._denoise_fwd(D1, [x], H, W, 3, 3, "D1c")
        L2 = self._new(1, 3, H, W)
        lib.call("zt_ew_f32", x, n, L2, 1, 1e-4, 1.0, n3, s)
        if cache_fn is None:
            wpH = wps = self._zeros(1, 3, H, W)
        else:
            wpH, wps = cache_fn(L2)
        s2 = self._enhancer_fwd(wpH, wps, L2, H, W)
        H2 = self._new(1, 3, H, W)
        lib.call("zt_ew_f32", x, s2, H2, 2, 1e-4, 1.0, n3, s)
        if cache_fn is None:
            wpH = wps = H2
        self.last_wp = (wpH, wps)
        r5 = self._denoise_fwd(D2, [wpH, wps, H2, s2], H, W, 12, 6, "D2c")
        H5p = self._new(1, 6, H, W)
        lib.call("zt_clamp_sub6_f32", H2, s2, r5, H5p, H5p[:, 3:], H * W, s)
        return H2, H5p[:, :3], H5p[:, 3:]

    # ------------------------------------------------------------------------------------------------ loss + backward
    def _terms(self, v, H, W):
        """Fused loss terms + direct gradients from the tensors in `v` (loss.py:23-78).  Returns (loss[1], terms[17], grads dict)."""
        o, lib, s = self.ops, self.lib, self._stream()
        h, w = H // 2, W // 2
        HW, hw = H * W, h * w
        terms = self._new(17)
        # ---- scalars (enhancement factor) and the s2 terms
        nb = max(1, min(256, HW // 4096))
        part = self._new(nb, 3)
        lib.call("zt_plane_sums_f32", v["L2"], 3, HW, nb, part, s)
        scal = self._new(8)
        lib.call("zt_loss_scalars_f32", part, nb, HW, int(self.is_WB), scal, s)
        Y = o.ycc_flat(v["L2"])
        ds2 = self._new(1, 3, H, W)
        nb1 = ((W + 63) // 64) * ((H + 3) // 4)
        p1 = self._new(nb1, 4)
        lib.call("zt_loss_s2_f32", v["L2"], v["s2"], Y, scal, H, W, ds2, p1, self.dt, s)
        o.partial_reduce(p1, nb1, 4, 4, out=terms)
        # ---- half-resolution terms
        LM1, LM2 = o.box5_reflect(v["H3d1"]), o.box5_reflect(v["H3d2"])
        dLp1, dLp2, dden1, dden2 = (self._new(1, 3, h, w) for _ in range(4))
        dH3p, dH4p = self._new(1, 6, h, w), self._new(1, 6, h, w)
        dH3d1, dH3d2, u1, u2 = (self._new(1, 3, h, w) for _ in range(4))
        nb2 = (hw + 255) // 256
        p2 = self._new(nb2, 10)
        lib.call("zt_loss_half_f32", v["Lq11"], v["Lq12"], v["Lp1"], v["Lp2"], v["den1"], v["den2"], v["H3p"], v["H4p"],
                 v["H11"], v["s21"], v["H12"], v["s22"], v["H3d1"], v["H3d2"], v["m_h"], LM1, LM2, dLp1, dLp2, dden1, dden2,
                 dH3p, dH4p, dH3d1, dH3d2, u1, u2, hw, p2, s)
        lib.call("zt_partial_reduce_f32", p2, nb2, 10, 8, terms.data_ptr() + 16, 0, None, s)
        lib.call("zt_partial_reduce_f32", p2.data_ptr() + 32, nb2, 10, 2, terms.data_ptr() + 56, 0, None, s)
        # ---- full-resolution terms
        DH2, VH2 = o.localvar_fwd(v["H2"])
        DN, VN = o.localvar_fwd(v["H3"], v["H2"])
        dH3b, ds3, gV = self._new(1, 3, H, W), self._new(1, 3, H, W), self._new(1, 3, H, W)
        nb3 = (3 * HW + 255) // 256
        p3 = self._new(nb3, 3)
        lib.call("zt_loss_full_f32", v["H2b"], v["H3b"], v["s2"], v["s3"], VH2, VN, dH3b, ds3, gV, 3 * HW, p3, s)
        lib.call("zt_partial_reduce_f32", p3, nb3, 3, 2, terms.data_ptr() + 48, 0, None, s)
        lib.call("zt_partial_reduce_f32", p3.data_ptr() + 8, nb3, 3, 1, terms.data_ptr() + 64, 0, None, s)
        loss = self._new(1)
        o.partial_reduce(terms, 17, 1, 1, out=loss)
        g = dict(ds2=ds2, dLp1=dLp1, dLp2=dLp2, dden1=dden1, dden2=dden2, dH3p=dH3p, dH4p=dH4p, dH3d1=dH3d1, dH3d2=dH3d2,
                 u1=u1, u2=u2, DH2=DH2, DN=DN, dH3b=dH3b, ds3=ds3, gV=gV)
        return loss, terms, g

    def loss_grads(self, grads):
        """Needs a preceding forward(keep=True).  `grads`: {name: zeroed tensor} receiving d loss / d parameter.
        Returns (loss 0-dim tensor, terms [17])."""
        o, lib, s, v = self.ops, self.lib, self._stream(), self.sv
        self.g = grads
        H, W = v["HW"]
        h, w = H // 2, W // 2
        HW, hw = H * W, h * w
        loss, terms, t = self._terms(v, H, W)
        ds2, dLp1, dLp2, dden1, dden2 = t["ds2"], t["dLp1"], t["dLp2"], t["dden1"], t["dden2"]
        dH3p, dH4p, dH3d1, dH3d2, u1, u2 = t["dH3p"], t["dH4p"], t["dH3d1"], t["dH3d2"], t["u1"], t["u2"]
        DH2, DN, dH3b, ds3, gV = t["DH2"], t["DN"], t["dH3b"], t["ds3"], t["gV"]
        # ---- backward: into H3 / H2
        o.box5_reflect_adj(u1, -1.0, out=dH3d1)
        o.box5_reflect_adj(u2, -1.0, out=dH3d2)
        dH3 = o.pair_down_adj(dH3d1, dH3d2, H, W)
        tmp = self._new(1, 3, H, W)
        o.blur21_adj(dH3b, out=dH3, tmp=tmp)
        o.localvar_bwd(DN, gV, -1.0, out=dH3)
        dH2x = o.localvar_bwd(DH2, gV, 1.0)
        o.localvar_bwd(DN, gV, 1.0, out=dH2x)
        # ---- through the three Denoise_2 invocations (model.py:179-192)
        dr5, dr3, dr4 = self._newa(1, H, W, 8), self._newa(1, h, w, 8), self._newa(1, h, w, 8)
        lib.call("zt_clamp_sub6_bwd", v["H2"], v["s2"], v["r5"], dH3, ds3, dr5, self.dt, 8, HW, s)
        lib.call("zt_clamp_sub6_bwd", v["H11"], v["s21"], v["r3"], dH3p, dH3p[:, 3:], dr3, self.dt, 8, hw, s)
        lib.call("zt_clamp_sub6_bwd", v["H12"], v["s22"], v["r4"], dH4p, dH4p[:, 3:], dr4, self.dt, 8, hw, s)
        dIn3 = self._denoise_bwd(D2, "D2a", dr3, 6, True)
        dIn4 = self._denoise_bwd(D2, "D2b", dr4, 6, True)
        dIn5 = self._denoise_bwd(D2, "D2c", dr5, 6, True)
        # ---- everything that reaches s2 -> Enhancer
        dO = self._newa(1, H, W, self.lda)
        lib.call("zt_post_enh_bwd", v["x"], v["s2"], v["L11"], v["L12"], v["s21"], v["s22"], dIn5, dH2x, dIn3, dIn4, ds2,
                 dO, self.dt, self.lda, None, H, W, s)
        self._enhancer_bwd(dO)
        # ---- Denoise_1 x3
        la = self.lda
        dn, dn11, dn12 = self._newa(1, H, W, la), self._newa(1, h, w, la), self._newa(1, h, w, la)
        lib.call("zt_d1_bwd_prep", v["x"], v["n"], dLp1, dLp2, dden1, dden2, dn, dn11, dn12, self.dt, la, H, W, s)
        self._denoise_bwd(D1, "D1a", dn11, 3, False)
        self._denoise_bwd(D1, "D1b", dn12, 3, False)
        self._denoise_bwd(D1, "D1c", dn, 3, False)
        self._wgrad_flush()
        return loss, terms


def _c(t):
    return t.detach().contiguous().float()


def loss_value(ops, is_WB, inp, Lp1, Lp2, L2, s2, s21, s22, H2, H11, H12, H3, s3, H3p, H4p, m_h, H2b, H3b):
    """LossFunction.forward (loss.py:23-78) on explicit tensors (the drop-in `loss.LossFunction`).  -> (loss[1], terms[17])."""
    eng = Engine(ops, {"_": _c(inp)}, {}, is_WB=is_WB, device=inp.device)
    _, _, H, W = inp.shape
    h, w = H // 2, W // 2
    lib, s = ops.lib, eng._stream()
    x, L11, L12, Lq11, Lq12 = eng._new(1, 3, H, W), eng._new(1, 3, h, w), eng._new(1, 3, h, w), eng._new(1, 3, h, w), eng._new(1, 3, h, w)
    lib.call("zt_prep_input_f32", _c(inp), x, L11, L12, Lq11, Lq12, H, W, s)
    den1, den2 = ops.pair_down(_c(L2))
    H3d1, H3d2 = ops.pair_down(_c(H3))
    v = dict(L2=_c(L2), s2=_c(s2), Lq11=Lq11, Lq12=Lq12, Lp1=_c(Lp1), Lp2=_c(Lp2), den1=den1, den2=den2, H3p=_c(H3p), H4p=_c(H4p),
             H11=_c(H11), s21=_c(s21), H12=_c(H12), s22=_c(s22), H3d1=H3d1, H3d2=H3d2, m_h=_c(m_h), H2=_c(H2), H3=_c(H3),
             H2b=_c(H2b), H3b=_c(H3b), s3=_c(s3))
    loss, terms, _ = eng._terms(v, H, W)
    return loss, terms


def smooth_tv_values(ops, L2, s2):
    """(SmoothLoss(L2, s2), L_TV(s2)) of loss.py:139-152, 173-311 (un-weighted), through the fused s2-term kernel."""
    eng = Engine(ops, {"_": _c(L2)}, {}, device=L2.device)
    _, _, H, W = L2.shape
    L2c, s2c = _c(L2), _c(s2)
    Y = ops.ycc_flat(L2c)
    scal = torch.ones(8, dtype=torch.float32, device=L2.device)
    nb1 = ((W + 63) // 64) * ((H + 3) // 4)
    p1, ds2, out = eng._new(nb1, 4), eng._new(1, 3, H, W), eng._new(4)
    ops.lib.call("zt_loss_s2_f32", L2c, s2c, Y, scal, H, W, ds2, p1, 0, eng._stream())
    ops.partial_reduce(p1, nb1, 4, 4, out=out)
    return out[2] / 5.0, out[3] / 1600.0
