"""Drop-in `Network` / `Finetunemodel` (reference model/model.py:84-384) on the MI355X kernels.

The module tree below exists to own the parameters under the reference's names, so `state_dict()` yields the same
223 keys (incl. the `enhance.blocks.{0,1,2}` aliases of the one shared conv+BN block and the frozen `raft.*` tensors),
`model.enhance.in_conv.apply(model.enhance_weights_init)` works (train.py:82-84) and reference checkpoints interchange.
No torch module here ever computes: `forward`/`_loss` hand the raw parameter storage to the static HIP plan
(engine.py / raft.py).  There is no CPU or eager fallback: without libzerotig_hip.so and a HIP device this raises.
"""
import torch
import torch.nn as nn

from .engine import Engine
from .lib import get_lib
from .ops import Ops
from .raft import RaftPlan


class _Holder(nn.Module):
    """Parameter container; computing with it is a bug (all compute is in the HIP plan)."""

    def forward(self, *a, **k):
        raise RuntimeError("parameter holder: compute runs in the HIP engine, not in torch modules")


class Denoise_1(_Holder):                                  # model.py:15-28
    def __init__(self, chan_embed=48):
        super().__init__()
        self.act = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        self.conv1 = nn.Conv2d(3, chan_embed, 3, padding=1)
        self.conv2 = nn.Conv2d(chan_embed, chan_embed, 3, padding=1)
        self.conv3 = nn.Conv2d(chan_embed, 3, 1)


class Denoise_2(_Holder):                                  # model.py:31-44 (built with chan_embed=48, model.py:91)
    def __init__(self, chan_embed=96):
        super().__init__()
        self.act = nn.LeakyReLU(negative_slope=0.2, inplace=True)
        self.conv1 = nn.Conv2d(12, chan_embed, 3, padding=1)
        self.conv2 = nn.Conv2d(chan_embed, chan_embed, 3, padding=1)
        self.conv3 = nn.Conv2d(chan_embed, 6, 1)


class Enhancer(_Holder):                                   # model.py:47-81
    def __init__(self, layers, channels):
        super().__init__()
        self.in_conv = nn.Sequential(nn.Conv2d(9, channels, 3, 1, 1), nn.ReLU())
        self.conv = nn.Sequential(nn.Conv2d(channels, channels, 3, 1, 1), nn.BatchNorm2d(channels), nn.ReLU())
        self.blocks = nn.ModuleList([self.conv for _ in range(layers)])       # ONE shared module, three aliases
        self.out_conv = nn.Sequential(nn.Conv2d(channels, 3, 3, 1, 1), nn.Sigmoid())


def _norm(kind, c):
    return nn.BatchNorm2d(c) if kind == "batch" else nn.InstanceNorm2d(c)


class _ResidualBlock(_Holder):                             # extractor.py:5-55
    def __init__(self, cin, planes, kind, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1, self.norm2 = _norm(kind, planes), _norm(kind, planes)
        if stride != 1:
            self.norm3 = _norm(kind, planes)
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride), self.norm3)
        else:
            self.downsample = None


class _BasicEncoder(_Holder):                              # extractor.py:117-165
    def __init__(self, output_dim, kind):
        super().__init__()
        self.norm1 = _norm(kind, 64)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(_ResidualBlock(64, 64, kind, 1), _ResidualBlock(64, 64, kind, 1))
        self.layer2 = nn.Sequential(_ResidualBlock(64, 96, kind, 2), _ResidualBlock(96, 96, kind, 1))
        self.layer3 = nn.Sequential(_ResidualBlock(96, 128, kind, 2), _ResidualBlock(128, 128, kind, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _MotionEncoder(_Holder):                             # update.py:79-87
    def __init__(self):
        super().__init__()
        self.convc1 = nn.Conv2d(324, 256, 1)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(256, 126, 3, padding=1)


class _SepConvGRU(_Holder):                                # update.py:33-42
    def __init__(self):
        super().__init__()
        for n in ("z", "r", "q"):
            setattr(self, "conv%s1" % n, nn.Conv2d(384, 128, (1, 5), padding=(0, 2)))
        for n in ("z", "r", "q"):
            setattr(self, "conv%s2" % n, nn.Conv2d(384, 128, (5, 1), padding=(2, 0)))


class _FlowHead(_Holder):                                  # update.py:6-14
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(128, 256, 3, padding=1)
        self.conv2 = nn.Conv2d(256, 2, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)


class _UpdateBlock(_Holder):                               # update.py:114-125
    def __init__(self):
        super().__init__()
        self.encoder, self.gru, self.flow_head = _MotionEncoder(), _SepConvGRU(), _FlowHead()
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 576, 1))


class RAFT(_Holder):                                       # raft.py:23-48
    def __init__(self, args):
        super().__init__()
        self.args = args
        args.corr_levels, args.corr_radius = 4, 4
        for k, v in (("of_scale", 3), ("dropout", 0), ("alternate_corr", False), ("mixed_precision", False)):
            if k not in args:
                setattr(args, k, v)
        self.hidden_dim = self.context_dim = 128
        self.fnet = _BasicEncoder(256, "instance")
        self.cnet = _BasicEncoder(256, "batch")
        self.update_block = _UpdateBlock()
        self.__dict__["_plan_cache"] = None

    def forward(self, image1, image2, iters=12, test_mode=True, ops=None):
        """raft.py:77-130: frames in [0,255], [1,3,h,w] -> (flow_low [1,2,h8,w8], flow_up [1,2,Hp,Wp] at the padded size)."""
        dev = image1.device
        key = (dev, self.fnet.conv1.weight.data_ptr())
        if self._plan_cache is None or self._plan_cache[0] != key:
            ops = ops if ops is not None else Ops(get_lib())
            rw = {"raft." + k: v.data for k, v in self.state_dict().items()}
            self.__dict__["_plan_cache"] = (key, RaftPlan(ops, rw, dev), ops)
        _, plan, ops = self._plan_cache
        _, _, h, w = image1.shape
        Hp, Wp = (h + 7) // 8 * 8, (w + 7) // 8 * 8
        x2 = torch.empty((2, Hp, Wp, 4), dtype=torch.float32, device=dev)
        from .lib import current_stream
        ops.lib.call("zt_raft_pack_pair", image1.detach().contiguous().float(), image2.detach().contiguous().float(), x2, 0, 4, h, w,
                     Hp, Wp, current_stream(dev))
        return plan.run(x2, iters=iters)


TRAINABLE = ("enhance.in_conv.0", "enhance.conv.0", "enhance.conv.1", "enhance.out_conv.0", "denoise_1.conv1",
             "denoise_1.conv2", "denoise_1.conv3", "denoise_2.conv1", "denoise_2.conv2", "denoise_2.conv3")


class _StepFn(torch.autograd.Function):
    """`loss = model._loss(x); loss.backward()` (train.py:128-129): forward runs the fused forward+loss+gradient plan,
    backward hands the already-computed parameter gradients to autograd's accumulation."""

    @staticmethod
    def forward(ctx, net, inp, *params):
        loss, grads = net._loss_and_grads(inp)
        ctx.grads, ctx.net = grads, net
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        fp, scratch = net.__dict__.get("_flat"), net.__dict__.get("_scratch")
        if (fp is not None and scratch is not None and fp.intact() and ctx.grads and ctx.grads[0].data_ptr() == scratch.data_ptr()
                and all(p.grad is not None and p.grad.data_ptr() == fp.grad.data_ptr() + 4 * o for p, o in zip(fp.params, fp.offsets))):
            # p.grad are views of the optimizer's flat bucket: ONE launch accumulates all 20 gradients (bucket += gout * scratch)
            from .lib import current_stream
            net._ops.lib.call("zt_axpy_dev_f32", fp.grad, scratch, gout.detach().float().contiguous(), fp.n, current_stream(fp.grad.device))
            return (None, None) + (None,) * len(ctx.grads)
        return (None, None) + tuple(g * gout for g in ctx.grads)


class _ZeroTIGBase(nn.Module):
    def _build_nets(self, args):
        self.enhance = Enhancer(layers=3, channels=64)
        self.denoise_1 = Denoise_1(chan_embed=48)
        self.denoise_2 = Denoise_2(chan_embed=48)

    def _finish_init(self, args, ops, precision=None):
        import os
        self.precision = precision or getattr(args, "precision", None) or os.environ.get("ZEROTIG_PRECISION", "fp32")
        assert self.precision in ("fp32", "bf16")
        self.last_H3 = self.last_H3_wp = self.last_s3 = self.last_s3_wp = None
        self.is_new_seq = True
        self.raft = self.load_raft(args)
        self.of_scale = args.of_scale
        self.__dict__["_ops"] = ops
        self.__dict__["_eng"] = None
        self.__dict__["_raftplan"] = None
        self.__dict__["_sig"] = None

    def load_raft(self, args):                              # model.py:109-115
        raft = RAFT(args)
        raft.eval()
        for p in raft.parameters():
            p.requires_grad = False
        return raft

    # ---- engine plumbing ---------------------------------------------------------------------------------------
    def _trainable(self):
        sd = dict(self.named_parameters())
        return [(n + s, sd[n + s]) for n in TRAINABLE for s in (".weight", ".bias")]

    def _plan(self):
        """(Re)bind the HIP plans to the current parameter storage (it moves on .cuda()/.to())."""
        tr = self._trainable()
        dev = tr[0][1].device
        sig = (dev, tuple(p.data_ptr() for _, p in tr), self.raft.fnet.conv1.weight.data_ptr(), self.precision)
        if self._sig != sig:
            if self._ops is None:
                if dev.type != "cuda":
                    raise RuntimeError("zero-tig_amd runs on a HIP device only; move the model with .cuda() first")
                self.__dict__["_ops"] = Ops(get_lib())
            params = {n: p.data for n, p in tr}
            bn = self.enhance.conv[1]
            bufs = {"enhance.conv.1.running_mean": bn.running_mean, "enhance.conv.1.running_var": bn.running_var,
                    "enhance.conv.1.num_batches_tracked": bn.num_batches_tracked}
            self.__dict__["_eng"] = Engine(self._ops, params, bufs, is_WB=getattr(self, "is_WB", False), device=dev,
                                           precision=self.precision)
            rw = {"raft." + k: v.data for k, v in self.raft.state_dict().items()}
            self.__dict__["_raftplan"] = RaftPlan(self._ops, rw, dev, precision=self.precision)
            self.__dict__["_sig"] = sig
        return self._eng, self._raftplan

    def flat_params(self):
        """One contiguous bucket for all trainable tensors (see optim.FlatParams); rebuilt if the storage moved."""
        from .optim import FlatParams
        fp = self.__dict__.get("_flat")
        if fp is None or not fp.intact():
            fp = FlatParams(self._trainable())
            self.__dict__["_flat"] = fp
        return fp

    def update_H3(self, H3, s3):                            # model.py:217-219
        st = self.__dict__.get("_static_cache")
        if st is not None and tuple(st[0].shape) == tuple(H3.shape):
            # hipGraph mode (optim.TrainStep): the recurrent cache keeps its address from frame to frame
            st[0].copy_(H3.detach())
            st[1].copy_(s3.detach())
            self.last_H3, self.last_s3 = st
            return
        self.last_H3 = H3.detach()
        self.last_s3 = s3.detach()

    def enable_static_cache(self, shape):
        """Allocate fixed buffers for last_H3 / last_s3 ([1,3,H,W]); update_H3 then copies into them (hipGraph replay needs stable
        addresses).  An existing cache is carried over."""
        dev = self._trainable()[0][1].device
        st = (torch.zeros((1, 3) + tuple(shape[-2:]), dtype=torch.float32, device=dev),
              torch.zeros((1, 3) + tuple(shape[-2:]), dtype=torch.float32, device=dev))
        if self.last_H3 is not None and tuple(self.last_H3.shape) == tuple(st[0].shape):
            st[0].copy_(self.last_H3)
            st[1].copy_(self.last_s3)
            self.last_H3, self.last_s3 = st
        self.__dict__["_static_cache"] = st

    def update_cache(self, last_H3, last_s3, L2):           # model.py:221-259
        _, rp = self._plan()
        return rp.update_cache(last_H3, last_s3, L2, self.of_scale)

    def enhance_weights_init(self, m):                      # model.py:123-130
        if isinstance(m, nn.Conv2d):
            m.weight.data.normal_(0.0, 0.02)
            if m.bias is not None:
                m.bias.data.zero_()
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.normal_(1.0, 0.02)

    denoise_weights_init = enhance_weights_init


class Network(_ZeroTIGBase):
    """Training-time model (model.py:84-259).  `ops` is for tests only (emulated backend); leave None in production."""

    def __init__(self, args, ops=None, precision=None):
        super().__init__()
        self._build_nets(args)
        self._l2_loss, self._l1_loss = nn.MSELoss(), nn.L1Loss()
        self.is_WB = "underwater" == args.dataset
        self._finish_init(args, ops, precision)

    def _forward_impl(self, input, keep):
        eng, rp = self._plan()
        eng.training = self.training
        x = input.detach().contiguous().float()
        if self.is_new_seq or self.last_H3 is None:
            outs = eng.forward(x, keep=keep, skip_unused=keep)
            self.last_H3_wp, self.last_s3_wp = eng.last_wp
        else:
            # the cache update needs L2 of the CURRENT frame (model.py:164), which the engine produces first
            outs = eng.forward(x, keep=keep, skip_unused=keep,
                               cache_fn=lambda L2: rp.update_cache(self.last_H3, self.last_s3, L2, self.of_scale))
            self.last_H3_wp, self.last_s3_wp = eng.last_wp
        return outs

    def forward(self, input):
        """-> the reference's 23-tuple (model.py:203); values only (the training gradient path is `_loss`)."""
        return self._forward_impl(input, keep=False)

    def _loss_and_grads(self, input, into=None):
        """-> (loss[1], [gradient tensors in _trainable() order]).  into: flat fp32 buffer (the optimizer's gradient bucket) that
        is zeroed and receives the gradients directly; default: a scratch bucket / fresh tensors for autograd to hand over."""
        outs = self._forward_impl(input, keep=True)
        eng = self._eng
        fp = self.__dict__.get("_flat")
        if into is not None:
            assert fp is not None and fp.intact() and into.numel() == fp.n
            into.zero_()
            grads = fp.grad_views(into)
        elif fp is not None and fp.intact():
            scratch = self.__dict__.get("_scratch")
            if scratch is None or scratch.numel() != fp.n or scratch.device != fp.flat.device:
                scratch = torch.empty_like(fp.flat)
                self.__dict__["_scratch"] = scratch
            scratch.zero_()
            grads = fp.grad_views(scratch)
        else:
            grads = {n: torch.zeros_like(p) for n, p in self._trainable()}
        loss, terms = eng.loss_grads(grads)
        self.__dict__["last_terms"] = terms
        self.update_H3(outs[13], outs[14])                  # model.py:214
        return loss, [grads[n] for n, _ in self._trainable()]

    def _loss(self, input):
        if torch.is_grad_enabled():
            return _StepFn.apply(self, input, *[p for _, p in self._trainable()])
        loss, _ = self._loss_and_grads(input)
        return loss.reshape(())


class Finetunemodel(_ZeroTIGBase):
    """Inference twin (model.py:262-384): full-resolution branch only; returns (H2, H3, s3)."""

    def __init__(self, args, ops=None, precision=None):
        super().__init__()
        self._build_nets(args)
        weights = getattr(args, "model_pretrain", None)
        if weights is not None:
            dev = "cuda:0" if torch.cuda.is_available() else "cpu"
            base = torch.load(weights, map_location=dev)
            md = self.state_dict()
            md.update({k: v for k, v in base.items() if k in md})
            self.load_state_dict(md)
        self._finish_init(args, ops, precision)

    weights_init = _ZeroTIGBase.enhance_weights_init

    def forward(self, input):
        eng, rp = self._plan()
        eng.training = False
        x = input.detach().contiguous().float()
        new = self.is_new_seq or self.last_H3 is None
        cache_fn = None if new else (lambda L2: rp.update_cache(self.last_H3, self.last_s3, L2, self.of_scale))
        H2, H3, s3 = eng.forward_infer(x, cache_fn)
        self.last_H3_wp, self.last_s3_wp = eng.last_wp
        self.update_H3(H3, s3)
        return H2, H3, s3
