"""Frozen RAFT-basic (reference model/RAFT/{raft,extractor,update,corr}.py) as a static inference plan over the HIP
kernels: instance-norm feature encoder on both frames, eval-BN context encoder, all-pairs correlation volume as an MFMA
1x1 convolution, pyramid, 12 refinement iterations (fused 4-level lookup, motion encoder, SepConvGRU, flow head) and
the convex 8x up-sampling of the last iteration only (the reference computes it 12 times and keeps the last).

precision "fp32": exact-fp32 MFMA everywhere (parity mode).  "bf16": feature maps / GRU state / weights bf16 in HBM with fp32
accumulation; the correlation volume, flow bookkeeping, up-sampling mask and everything downstream (warp) stay fp32."""
import os

import torch

from .lib import current_stream
from .ops import CV


_ONE_LAUNCH_IN = os.environ.get("ZT_INSTNORM_ONE_LAUNCH", "0") == "1"
# Correlation volume + its three pooled levels in one launch (csrc/zt_corr.hip) or as the tiled GEMM + three pooling launches.
# Measured (gpurun_out/r03f): 1080p (3 600 x 3 600 volume, 68 MB with the pyramid) 54 us fused vs 29 + 3 x 7.5 = 51.5 us -- the
# fused kernel's four DMA -> barrier -> 32-MFMA phases per workgroup are latency-bound at that size; 4K (14 400^2, 1.1 GB) 0.53 ms
# fused vs 0.39 + 3 x 0.087 = 0.65 ms.  "auto" = fused from 8 000 map pixels up; ZT_FUSED_CORR=1 / 0 force either.
_FUSED_CORR = os.environ.get("ZT_FUSED_CORR", "auto")

class _Side:
    """`with plan._side():` runs the enclosed launches on a second HIP stream, forked from / joined back into the current one
    (also under hipGraph capture, where it becomes a parallel branch of the graph).  The small-map RAFT kernels are latency /
    issue bound and leave most CUs idle, so the context encoder overlaps the feature encoder (-110 us per RAFT call).  Buffers shared across the two streams are allocated before the fork and outlive
    the join; temporaries stay on the stream that made them."""

    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        p = self.plan
        if p.dev.type != "cuda" or not p.two_streams:
            return self
        if p._s2 is None:
            p._s2 = torch.cuda.Stream(device=p.dev)
        self.main = torch.cuda.current_stream(p.dev)
        p._s2.wait_stream(self.main)
        self.ctx = torch.cuda.stream(p._s2)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        p = self.plan
        if p.dev.type != "cuda" or not p.two_streams:
            return False
        self.ctx.__exit__(*exc)
        p._pending_join = True
        return False


class RaftPlan:
    def _side(self):
        return _Side(self)

    def _join(self):
        """the current stream waits for the side branch (no-op when nothing is pending)"""
        if self._pending_join:
            torch.cuda.current_stream(self.dev).wait_stream(self._s2)
            self._pending_join = False

    def __init__(self, ops, weights, device, prefix="raft", precision="fp32"):
        import os
        self.two_streams = os.environ.get("ZT_RAFT_STREAMS", "2") != "1"
        self._s2, self._pending_join = None, False
        """weights: {name: tensor on device} with the reference's `raft.*` state-dict names."""
        self.ops, self.lib, self.dev, self.pre = ops, ops.lib, device, prefix
        self.W = dict(weights)
        self.h = precision == "bf16"
        self.adt = torch.bfloat16 if self.h else torch.float32
        self.dt = 1 if self.h else 0
        self.wd, self.bn = {}, {}
        self._prepare()

    def _w(self, name):
        return self.W[self.pre + "." + name]

    def _prepare(self):
        o = self.ops
        rp = o.repack_weight_bf16 if self.h else o.repack_weight
        for k, v in list(self.W.items()):
            if not k.startswith(self.pre + ".") or not k.endswith(".weight") or v.dim() != 4:
                continue
            name = k[len(self.pre) + 1:-7]
            if ".gru.conv" in name:
                continue
            self.wd[name] = rp(v.contiguous())
        self.wstem = {}
        if self.h:
            for enc in ("fnet", "cnet"):
                self.wstem[enc] = o.raft_stem_weight_bf16(self._w(enc + ".conv1.weight").contiguous())
        g = "update_block.gru."
        for sfx in ("1", "2"):
            wz, wr = self._w(g + "convz" + sfx + ".weight"), self._w(g + "convr" + sfx + ".weight")
            kh, kw = wz.shape[2], wz.shape[3]
            if self.h:
                buf = torch.zeros((kh * kw, 256, 384), dtype=torch.bfloat16, device=self.dev)
                o.repack_weight_bf16(wz.contiguous(), out=buf, co_off=0)
                o.repack_weight_bf16(wr.contiguous(), out=buf, co_off=128)
            else:
                buf = torch.zeros((kh * kw, 384, 256), dtype=torch.float32, device=self.dev)
                o.repack_weight(wz.contiguous(), ldw=256, co_off=0, out=buf)
                o.repack_weight(wr.contiguous(), ldw=256, co_off=128, out=buf)
            self.wd[g + "convzr" + sfx] = buf
            self.W[self.pre + "." + g + "convzr" + sfx + ".bias"] = torch.cat(
                [self._w(g + "convz" + sfx + ".bias"), self._w(g + "convr" + sfx + ".bias")]).contiguous()
            self.wd[g + "convq" + sfx] = rp(self._w(g + "convq" + sfx + ".weight").contiguous())
        # eval-mode BatchNorm of the context encoder folded to scale / shift once (running stats are frozen)
        for k in list(self.W.keys()):
            if k.startswith(self.pre + ".cnet") and k.endswith(".running_mean") and ".downsample.1." not in k:
                name = k[len(self.pre) + 1:-13]
                C = self.W[k].numel()
                sc, sh, _, _ = o.norm_finalize(None, 1, C, 1, 2, self._w(name + ".weight"), self._w(name + ".bias"),
                                               self._w(name + ".running_mean"), self._w(name + ".running_var"), dev=self.dev)
                self.bn[name] = (sc, sh)
        # bf16 mode: that scale / shift goes into the weights and bias of the conv in front of it (conv1 -> norm1, conv2 -> norm2,
        # downsample.0 -> norm3), the ReLUs and the residual add into that conv's epilogue: the context encoder's 14 normalisation
        # passes after the stem disappear (ZT_RAFT_FOLD_BN=0: separate passes, the A/B and the fp32 plan's form)
        self.folded = {}
        if self.h and os.environ.get("ZT_RAFT_FOLD_BN", "1") != "0":
            for name, (sc, sh) in list(self.bn.items()):
                if name == "cnet.norm1":                         # stem: relu(norm1(conv1(x))), extractor.py:168-170
                    w = self._w("cnet.conv1.weight").float() * sc.view(-1, 1, 1, 1)
                    self.wstem["cnet#bn"] = o.raft_stem_weight_bf16(w.contiguous())
                    self.W[self.pre + ".cnet.conv1#bn.bias"] = (self._w("cnet.conv1.bias").float() * sc + sh).contiguous()
                    continue
                if not name.startswith("cnet.layer"):
                    continue
                blk, nrm = name.rsplit(".", 1)
                conv = blk + {"norm1": ".conv1", "norm2": ".conv2", "norm3": ".downsample.0"}[nrm]
                w = self._w(conv + ".weight").float() * sc.view(-1, 1, 1, 1)
                self.wd[conv + "#bn"] = rp(w.contiguous())
                self.W[self.pre + "." + conv + "#bn.bias"] = (self._w(conv + ".bias").float() * sc + sh).contiguous()
                self.folded[conv] = True

    # ------------------------------------------------------------------------------------------------ building blocks
    def _conv(self, x, name, cout, k, stride=1, pad=None, act=None, alpha=1.0, out=None, x2=None, out_f32=False, bias=True, row0=0,
              aux=None, epi=0, out2=None, esplit=0):
        """Convolution in the plan's precision.  row0: first output channel of the packed weight to use (1x1 only)."""
        kh, kw = (k, k) if isinstance(k, int) else k
        pad = (kh // 2, kw // 2) if pad is None else pad
        b = self._w(name + ".bias") if bias else None
        if b is not None and row0:
            b = b[row0:]
        if self.h:
            return self.ops.conv2d_bf16(x, self.wd[name], b, cout, kh, kw, pad, act, alpha=alpha, out=out, stride=stride, x2=x2,
                                        out_f32=out_f32, w_roff=row0, aux=aux, epi=epi, out2=out2, esplit=esplit)
        return self.ops.conv2d(x, self.wd[name], b, cout, kh, kw, stride, pad, act, alpha=alpha, out=out, x2=x2, w_coff=row0,
                               aux=aux, epi=epi, out2=out2, esplit=esplit)

    def _norm(self, y, name, kind, inner_relu, res=None, outer_relu=False):
        o = self.ops
        if kind == "instance":
            if _ONE_LAUNCH_IN:      # measured slower: the device-scope release in every workgroup writes back its XCD's L2
                sc, sh = o.instance_norm_stats(y)
            else:
                yv = CV(y)
                part = o.chan_stats(y)
                sc, sh, _, _ = o.norm_finalize(part, yv.N, yv.C, yv.H * yv.W, 0)
        else:
            sc, sh = self.bn[name]
        return o.norm_apply(y, sc, sh, res=res, inner_relu=inner_relu, outer_relu=outer_relu)

    def _res_block(self, p, x, dim, stride, kind):
        if kind == "batch" and self.folded.get(p + ".conv1"):
            y = self._conv(x, p + ".conv1#bn", dim, 3, stride, act="relu")
            xs = self._conv(x, p + ".downsample.0#bn", dim, 1, stride, pad=(0, 0)) if stride != 1 else x
            return self._conv(y, p + ".conv2#bn", dim, 3, 1, act="relu", aux=xs, epi=6)
        y = self._conv(x, p + ".conv1", dim, 3, stride)
        y = self._norm(y, p + ".norm1", kind, True)
        y = self._conv(y, p + ".conv2", dim, 3, 1)
        if stride != 1:
            xs = self._conv(x, p + ".downsample.0", dim, 1, stride, pad=(0, 0))
            xs = self._norm(xs, p + ".norm3", kind, False)
        else:
            xs = x
        return self._norm(y, p + ".norm2", kind, True, res=xs, outer_relu=True)

    def _encoder(self, enc, x, kind):
        """extractor.py:117-191 up to (not including) the 1x1 output conv."""
        skip_norm1 = False
        if self.h and x.shape[-1] == 8:       # bf16 mode: dedicated stem kernel (7 px x 8 ch of a kernel row = one 64-wide K range)
            if kind == "batch" and (enc + "#bn") in self.wstem:
                y = self.ops.raft_stem_bf16(x, self.wstem[enc + "#bn"], self._w(enc + ".conv1#bn.bias"), relu=True)
                skip_norm1 = True
            else:
                y = self.ops.raft_stem_bf16(x, self.wstem[enc], self._w(enc + ".conv1.bias"))
        else:
            y = self._conv(CV(x, 0, 3), enc + ".conv1", 64, 7, 2)
        if not skip_norm1:
            y = self._norm(y, enc + ".norm1", kind, True)
        for li, dim, stride in ((1, 64, 1), (2, 96, 2), (3, 128, 2)):
            y = self._res_block("%s.layer%d.0" % (enc, li), y, dim, stride, kind)
            y = self._res_block("%s.layer%d.1" % (enc, li), y, dim, 1, kind)
        return y

    def _new(self, *shape, dtype=None, zero=False):
        f = torch.zeros if zero else torch.empty
        return f(shape, dtype=self.adt if dtype is None else dtype, device=self.dev)

    # ------------------------------------------------------------------------------------------------ full inference
    def run(self, x2, iters=12, want_aux=False):
        """x2: NHWC [2,Hp,Wp,ld] (both padded, normalised frames) in the plan's storage type.
        Returns (flow_low [1,2,h8,w8], flow_up [1,2,Hp,Wp]) fp32."""
        o, lib, dev, dt = self.ops, self.lib, self.dev, self.dt
        s = current_stream(dev)
        _, Hp, Wp, _ = x2.shape
        h, w = Hp // 8, Wp // 8
        npx = h * w
        # context encoder -> hidden state (tanh) and context (relu) straight into the GRU input buffer: an independent branch
        HX = self._new(1, h, w, 384, zero=True)                              # [net | inp | motion(126) | flow(2)]
        with self._side():
            c = self._encoder("cnet", x2[0:1], "batch")
            self._conv(CV(c), "cnet.conv2", 128, 1, act="tanh", out=CV(HX, 0, 128))
            self._conv(CV(c), "cnet.conv2", 128, 1, act="relu", out=CV(HX, 128, 128), row0=128)
            del c
        # feature encoder (both frames), correlation volume + pyramid
        f = self._encoder("fnet", x2, "instance")
        fmap1 = self._conv(CV(f[0:1]), "fnet.conv2", 256, 1)
        if self.h:
            npxp = (npx + 15) // 16 * 16
            fmap2 = self._new(npxp, 256, zero=True)                         # nhwc == the bf16 weight layout [CoutP][ldk]
            self._conv(CV(f[1:2]), "fnet.conv2", 256, 1, out=fmap2[:npx].view(1, h, w, 256))
            if h >= 16 and w >= 16 and (_FUSED_CORR == "1" or (_FUSED_CORR == "auto" and npx >= 8000)):
                corr0, levels = o.corr_volume_pyramid_bf16(fmap1.view(npx, 256), fmap2, h, w, 1.0 / 16.0)
            else:
                corr0 = o.conv2d_bf16(CV(fmap1), fmap2.view(1, npxp, 256), None, npx, 1, 1, alpha=1.0 / 16.0, out_f32=True)
                levels = o.corr_pyramid(corr0, h, w)
        else:
            pitch = (npx + 15) // 16 * 16
            fmap2 = self._new(1, 256, pitch, zero=True)
            o.conv2d(CV(f[1:2]), self.wd["fnet.conv2"], self._w("fnet.conv2.bias"), 256, 1, 1, out=fmap2, out_planar=True)
            corr0 = o.conv2d(CV(fmap1), fmap2, None, npx, 1, 1, alpha=1.0 / 16.0)      # corr.py:52-60: / sqrt(256)
            levels = o.corr_pyramid(corr0, h, w)
        self._join()
        st = self.new_state(h, w, HX)
        st.corr0, st.levels = corr0, levels
        aux = {}
        for it in range(iters):
            self.refine_step(st)
            if want_aux and it == 0:
                aux["corr0"] = st.CORR.clone()
        flow_low, flow_up, mask = self.finish(st, Hp, Wp)
        if want_aux:
            aux.update(fmap1=fmap1, fmap2=fmap2, HX=HX, mask=mask)
            return flow_low, flow_up, aux
        return flow_low, flow_up

    # ---- the refinement loop of raft.py:112-126, one iteration at a time (tests drive a single step against the golden) ----
    class State:
        pass

    def new_state(self, h, w, HX, coords=None):
        """Buffers of the refinement loop.  HX: [1,h,w,384] = [net | inp | motion(126) | flow(2)] with net / inp filled in.
        coords: optional [h*w,2] starting coordinates (x, y); default = the pixel grid (zero flow, raft.py:107)."""
        lib, dt, s = self.lib, self.dt, current_stream(self.dev)
        st = RaftPlan.State()
        st.h, st.w, st.npx, st.HX = h, w, h * w, HX
        st.coords1 = self._new(st.npx, 2, dtype=torch.float32)
        lib.call("zt_raft_coords_init_f32", st.coords1, h, w, s)
        st.F4 = self._new(1, h, w, 4, dtype=torch.float32, zero=True)        # fp32 flow for the up-sampler
        st.ldfin = 8 if self.h else 4
        st.FIN = self._new(1, h, w, st.ldfin, zero=True)                     # flow as the 7x7 conv input
        st.es = HX.element_size()
        delta, ldd = None, 0
        if coords is not None:                                               # flow = coords - grid, applied as a first "delta"
            delta = torch.zeros((st.npx, 4), dtype=torch.float32, device=self.dev)
            delta[:, :2] = coords.to(self.dev) - st.coords1
            ldd = 4
        lib.call("zt_raft_flow_step", st.coords1, delta, ldd, h, w, st.F4, 4, HX.data_ptr() + st.es * 382, 384, st.FIN, st.ldfin, dt, s)
        st.coords2 = torch.empty_like(st.coords1)                            # ping-pong partner (the lookup applies the pending delta)
        st.pending = False                                                   # a flow-head delta not yet added to coords1
        st.CF, st.RH, st.ZR = self._new(1, h, w, 256), self._new(1, h, w, 128), self._new(1, h, w, 256)
        st.C1, st.F1 = self._new(1, h, w, 256), self._new(1, h, w, 128)      # convc1 / convf1 outputs (re-used every iteration)
        st.CORR = self._new(1, h, w, 328 if self.h else 324, zero=True)
        st.delta = self._new(1, h, w, 4, dtype=torch.float32, zero=True) if self.h else None   # flow-head output, reused every iteration
        return st

    def refine_step(self, st):
        """corr lookup (corr.py:29-50) -> BasicUpdateBlock (update.py:114-136, mask head deferred to finish()) -> coords1 += delta."""
        o, lib, dt, s = self.ops, self.lib, self.dt, current_stream(self.dev)
        h, w, npx, HX, CF, RH, CORR = st.h, st.w, st.npx, st.HX, st.CF, st.RH, st.CORR
        e, g = "update_block.encoder.", "update_block.gru."
        # (the two halves of the motion encoder are independent too, but a fork / join per iteration costs more than the overlap
        # of two ~20 us branches returns: +6 us per iteration measured with tools/bench_raft.py)
        if st.pending:
            # the previous iteration's `coords1 += delta_flow` (raft.py:120) rides in this iteration's lookup: it reads coords1 + delta
            # and records the sum (ping-pong buffer) and the flow for the up-sampler / the 7x7 conv / the GRU input, all of which
            # are consumed later in this iteration -- one launch less per iteration than a separate zt_raft_flow_step
            o.corr_lookup_step(st.corr0, st.levels, h, w, st.coords1, CORR, st.delta, st.coords2, st.F4, HX.data_ptr() + st.es * 382, 384, st.FIN)
            st.coords1, st.coords2, st.pending = st.coords2, st.coords1, False
        else:
            o.corr_lookup(st.corr0, st.levels, h, w, st.coords1, out=CORR)
        if self.h:      # bf16 mode: the two branches of the motion encoder (update.py:89-94) are independent -> two launches, not four
            o.conv_pair_bf16(CV(CORR, 0, 324), self.wd[e + "convc1"], self._w(e + "convc1.bias"), 256, 1, st.C1,
                             CV(st.FIN, 0, 2), self.wd[e + "convf1"], self._w(e + "convf1.bias"), 128, 7, st.F1, act="relu")
            o.conv_pair_bf16(st.C1, self.wd[e + "convc2"], self._w(e + "convc2.bias"), 192, 3, CV(CF, 0, 192),
                             st.F1, self.wd[e + "convf2"], self._w(e + "convf2.bias"), 64, 3, CV(CF, 192, 64), act="relu")
        else:
            cor1 = self._conv(CV(CORR, 0, 324), e + "convc1", 256, 1, act="relu")
            flo1 = self._conv(CV(st.FIN, 0, 2), e + "convf1", 128, 7, act="relu")
            self._conv(cor1, e + "convc2", 192, 3, act="relu", out=CV(CF, 0, 192))
            self._conv(flo1, e + "convf2", 64, 3, act="relu", out=CV(CF, 192, 64))
        self._conv(CF, e + "conv", 126, 3, act="relu", out=CV(HX, 256, 126))
        for sfx, k, pad in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            # z, r = sigmoid(conv[h | x]) with r * h formed in the epilogue; q = tanh(conv[r*h | x]) with the state update
            # h = (1 - z) h + z q written in place by the epilogue (update.py:42-58): no stand-alone element-wise launches
            self._conv(HX, g + "convzr" + sfx, 256, k, pad=pad, act="sigmoid", out=st.ZR, aux=CV(HX, 0, 128), epi=4, out2=CV(RH), esplit=128)
            self._conv(CV(RH), g + "convq" + sfx, 128, k, pad=pad, act="tanh", x2=CV(HX, 128, 256), out=CV(HX, 0, 128),
                       aux=CV(st.ZR, 0, 128), epi=5)
        fh = self._conv(CV(HX, 0, 128), "update_block.flow_head.conv1", 256, 3, act="relu")
        st.delta = self._conv(fh, "update_block.flow_head.conv2", 2, 3, out_f32=True, out=st.delta)       # fp32 [..,4]
        st.pending = True

    def flush_flow(self, st):
        """apply a pending flow-head delta (after the last iteration, or when a test inspects the state between iterations)"""
        if st.pending:
            self.lib.call("zt_raft_flow_step", st.coords1, st.delta, st.delta.shape[-1], st.h, st.w, st.F4, 4,
                          st.HX.data_ptr() + st.es * 382, 384, st.FIN, st.ldfin, self.dt, current_stream(self.dev))
            st.pending = False

    def finish(self, st, Hp, Wp):
        """mask head (update.py:122-125, 134) on the final hidden state + convex 8x up-sampling (raft.py:64-75)."""
        HX, h, w = st.HX, st.h, st.w
        self.flush_flow(st)
        m1 = self._conv(CV(HX, 0, 128), "update_block.mask.0", 256, 3, act="relu")
        mask = self._conv(m1, "update_block.mask.2", 576, 1, alpha=0.25, out_f32=True)
        flow_up = self._new(1, 2, Hp, Wp, dtype=torch.float32)
        flow_low = self._new(1, 2, h, w, dtype=torch.float32)
        self.lib.call("zt_convex_upsample_f32", st.F4, 4, mask, 576, flow_up, flow_low, h, w, current_stream(self.dev))
        return flow_low, flow_up, mask

    def update_cache(self, last_H3, last_s3, L2, of_scale, want_aux=False):
        """model.py:221-259: down-scale, equalise the current frame, RAFT(12), backward-warp both cached tensors."""
        o = self.ops
        _, _, H, W = last_H3.shape
        ht, wd = H // of_scale, W // of_scale
        a = o.resize_bilinear(last_H3.contiguous(), ht, wd, 255.0)
        b = o.resize_bilinear(L2, ht, wd, 255.0)
        q, _, lut = o.equalize_prepare(b)
        x2 = o.raft_pack_input(a, q, lut, ht, wd, dtype=self.adt)
        flow_low, flow_up = self.run(x2)
        wpH, wps = o.warp2(flow_up, last_H3.contiguous(), last_s3.contiguous())
        if want_aux:
            return wpH, wps, flow_low, flow_up
        return wpH, wps
