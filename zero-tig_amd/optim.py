"""Flat-bucket optimizer for the Zero-TIG training loop (reference train.py:98, 126-131).

`FlatParams` re-homes the 20 trainable tensors into ONE contiguous fp32 buffer (parameters and gradients become views),
so that (a) the data-parallel exchange is a single RCCL all-reduce of 370 KB per step and (b) gradient clipping + Adam are
two kernel launches.  `ClipAdam.step()` == `clip_grad_norm_(params, max_norm)` + `Adam.step()` of the reference."""
import torch

from .lib import current_stream


class FlatParams:
    def __init__(self, named_params):
        """named_params: [(name, nn.Parameter)] (trainable, unique)."""
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.n = sum(self.sizes)
        self.flat = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        off = 0
        self.offsets = []
        for p, sz in zip(self.params, self.sizes):
            self.flat[off:off + sz].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + sz].view(p.shape)
            p.grad = self.grad[off:off + sz].view(p.shape)
            self.offsets.append(off)
            off += sz

    def intact(self):
        return all(p.data_ptr() == self.flat.data_ptr() + 4 * o for p, o in zip(self.params, self.offsets))

    def grad_views(self, buf):
        return {n: buf[o:o + s].view(p.shape) for n, p, o, s in zip(self.names, self.params, self.offsets, self.sizes)}


class ClipAdam:
    """clip_grad_norm_(max_norm) + Adam(lr, betas, eps, weight_decay as L2) over a model's flat bucket, on the HIP stream.
    With torch.distributed initialised, gradients are all-reduced (mean) over RCCL first: one bucket, one collective."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=3e-4, max_norm=5.0, process_group=None):
        self.model = model
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.pg = process_group
        self.t = 0
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay, "capturable": True}]
        self._bind()

    def _bind(self):
        self.fp = self.model.flat_params()
        dev = self.fp.flat.device
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.partial = torch.empty(256, dtype=torch.float32, device=dev)
        self.gnorm = torch.zeros(1, dtype=torch.float32, device=dev)

    def state_dict(self):
        """Adam moments and step count over the flat bucket, keyed by parameter name (resume state; the reference saves none)."""
        fp = self.fp
        sl = lambda buf: {n: buf[o:o + s].view(p.shape).detach().cpu().clone() for n, p, o, s in zip(fp.names, fp.params, fp.offsets, fp.sizes)}
        return {"t": self.t, "exp_avg": sl(self.m), "exp_avg_sq": sl(self.v), "lr": self.param_groups[0]["lr"]}

    def load_state_dict(self, sd):
        fp = self.fp
        for n, p, o, s in zip(fp.names, fp.params, fp.offsets, fp.sizes):
            self.m[o:o + s].copy_(sd["exp_avg"][n].reshape(-1))
            self.v[o:o + s].copy_(sd["exp_avg_sq"][n].reshape(-1))
        self.t = int(sd["t"])

    def zero_grad(self, set_to_none=False):
        self.fp.grad.zero_()
        for p, o, s in zip(self.fp.params, self.fp.offsets, self.fp.sizes):
            if p.grad is None or p.grad.data_ptr() != self.fp.grad.data_ptr() + 4 * o:
                p.grad = self.fp.grad[o:o + s].view(p.shape)

    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if dist.is_available() and dist.is_initialized() else 1

    def _check_binding(self):
        """The parameters and their .grad must still be views of the flat bucket (they stop being so after `model.to()/.cuda()`,
        `model.zero_grad(set_to_none=True)` or `p.grad = None`).  Stray gradients are copied back into the bucket and re-bound;
        parameters that moved cannot be repaired silently (Adam moments belong to the old storage): that is an error."""
        fp = self.fp
        if not fp.intact():
            raise RuntimeError("ClipAdam: the model's parameters no longer live in the optimizer's flat bucket (model.to()/.cuda() "
                               "after the optimizer was built?); build the optimizer after moving the model")
        base = fp.grad.data_ptr()
        for p, o, s in zip(fp.params, fp.offsets, fp.sizes):
            view = fp.grad[o:o + s].view(p.shape)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != base + 4 * o:
                view.copy_(p.grad)
            else:
                continue
            p.grad = view

    def step(self):
        import torch.distributed as dist
        self._check_binding()
        fp = self.fp
        ws = self.world()
        if dist.is_available() and dist.is_initialized():        # also with ONE rank: the same RCCL call the N-rank job makes
            dist.all_reduce(fp.grad, op=dist.ReduceOp.SUM, group=self.pg)
        self.t += 1
        lib = self.model._ops.lib
        lib.call("zt_clip_adam_f32", fp.flat, fp.grad, self.m, self.v, fp.n, self.partial, 128, 1.0 / ws, float(self.max_norm),
                 float(self.param_groups[0]["lr"]), self.betas[0], self.betas[1], self.eps, self.wd, self.t, self.gnorm,
                 current_stream(fp.flat.device))
        return self.gnorm


class TrainStep:
    """One iteration of the reference loop (train.py:119-131) as a callable: `loss = step(frame, is_new_seq)`.

    use_graph=False: eager launches (~550 kernel launches through ctypes per step).
    use_graph=True : the steady-state step (zero_grad + forward incl. RAFT + loss + backward into the flat gradient bucket) is
    captured ONCE into a hipGraph on first use and replayed afterwards -- the launch-bound host loop (~14 us x 550 launches)
    becomes one graph launch.  The frame is copied into a static input buffer (host -> HBM or HBM -> HBM) and the recurrent
    cache `last_H3 / last_s3` lives in static buffers, so the captured addresses stay valid from frame to frame.  New-sequence
    frames (rare: once per clip) and the gradient all-reduce + clip + Adam launch (which takes the host-side step count) stay
    eager.  The model must not be moved, and `last_H3 / last_s3` must not be re-assigned by the caller, after the capture."""

    def __init__(self, model, optimizer, use_graph=True, ingest_size=(1920, 1080)):
        """ingest_size: (W, H) that decoded uint8 frames are resized to (the reference loader's fixed 1920 x 1080); None = keep."""
        self.model, self.opt, self.use_graph, self.ingest_size = model, optimizer, use_graph, ingest_size
        self.graph, self.x, self.loss, self.n_eager_steady, self._mode = None, None, None, 0, None

    def _body(self, x):
        """zero_grad + loss + gradients straight into the optimizer's flat bucket (no autograd bookkeeping, no ATen math)"""
        opt = self.opt
        opt._check_binding()
        loss, _ = self.model._loss_and_grads(x, into=opt.fp.grad)
        return loss.reshape(())

    def _load(self, frame, dev, out=None):
        """frame -> fp32 [1,3,H,W] on the device.  uint8 [1,H0,W0,3] / [H0,W0,3] frames (the loaders' device-ingest mode: decoded
        only) go through the ingest kernels: PIL-exact resize to 1920x1080 + ToTensor (multi_read_data.py:127-132), written
        straight into `out` when given."""
        import torch
        if frame.dtype == torch.uint8:
            self.model._plan()
            return self.model._ops.ingest_u8(frame.to(dev, non_blocking=True), out=out, size=self.ingest_size)
        if out is None:
            return frame.to(dev, non_blocking=True)
        out.copy_(frame, non_blocking=True)
        return out

    def __call__(self, frame, is_new_seq=False):
        import torch
        m = self.model
        m.is_new_seq = bool(is_new_seq)
        dev = self.opt.fp.flat.device
        if not self.use_graph:
            with torch.no_grad():
                loss = self._body(self._load(frame, dev))
            self.opt.step()
            return loss
        if frame.dtype == torch.uint8:
            Wi, Hi = self.ingest_size if self.ingest_size is not None else (frame.shape[-2], frame.shape[-3])
            shape = (1, 3, Hi, Wi)
        else:
            shape = tuple(frame.shape)
        if self.graph is not None and (self._mode != m.training or tuple(self.x.shape) != shape):
            # BatchNorm mode (the reference trains epochs >= 1 in eval mode, train.py:138) or the frame size changed: the captured
            # launch sequence no longer applies -> capture again after one eager step
            self.graph, self.n_eager_steady = None, 0
            if tuple(self.x.shape) != shape:
                self.x = None
        if self.x is None:
            self.x = torch.empty(shape, dtype=torch.float32, device=dev)
            m.enable_static_cache(shape)
        self._load(frame, dev, out=self.x)
        if is_new_seq or m.last_H3 is None or self.n_eager_steady < 1:
            # eager: new-sequence frames, and the first steady-state frame (loads every kernel's code object, sizes the slabs and
            # the plan's persistent buffers before anything is captured)
            if not (is_new_seq or m.last_H3 is None):
                self.n_eager_steady += 1
            with torch.no_grad():
                loss = self._body(self.x)
            self.opt.step()
            return loss
        if self.graph is None:
            torch.cuda.synchronize(dev)
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.loss = self._body(self.x)
            self.graph, self._mode = g, m.training
        self.graph.replay()
        self.opt.step()
        return self.loss


class FramePrefetcher:
    """Host -> HBM copies of the NEXT frame on a copy stream while the current step computes (train.py:125 does a blocking
    `.cuda()` per step: 24.9 MB at 1080p, ~0.8 ms on the compute stream).  `frames` is any iterable of items whose first element
    (or the item itself) is a pinned [1,3,H,W] fp32 tensor -- or, in the loaders' device-ingest mode, the decoded uint8 [1,H0,W0,3]
    frame (6 MB instead of 25 MB at 1080p; `TrainStep` runs the ingest kernels on it); iteration yields the same items with that tensor replaced by a device
    tensor that is ready on the current stream.  Two device buffers alternate."""

    def __init__(self, frames, device):
        import torch
        self.it, self.dev = iter(frames), device
        self.copy_stream = torch.cuda.Stream(device=device)
        self.buf = [None, None]
        self.k = 0
        self.pending = None
        self._issue()

    def _issue(self):
        import torch
        try:
            item = next(self.it)
        except StopIteration:
            self.pending = None
            return
        host = item[0] if isinstance(item, (tuple, list)) else item
        b = self.buf[self.k]
        if b is None or b.shape != host.shape or b.dtype != host.dtype:
            b = self.buf[self.k] = torch.empty(host.shape, dtype=host.dtype, device=self.dev)
        self.copy_stream.wait_stream(torch.cuda.current_stream(self.dev))       # the buffer's previous consumer has been enqueued
        with torch.cuda.stream(self.copy_stream):
            b.copy_(host, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        self.pending = (item, b, ev, host)
        self.k ^= 1

    def __iter__(self):
        return self

    def __next__(self):
        import torch
        if self.pending is None:
            raise StopIteration
        item, b, ev, _host = self.pending
        torch.cuda.current_stream(self.dev).wait_event(ev)
        self._issue()                                          # next frame's copy overlaps this frame's step
        if isinstance(item, (tuple, list)):
            return (b,) + tuple(item[1:])
        return b
