"""Tensor-level wrappers over the C ABI: allocate outputs with torch (plumbing), launch the HIP kernels on the
current stream.  No compute happens in torch here."""
import torch

from .lib import current_stream


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


class Ops:
    def __init__(self, lib):
        self.lib = lib

    def _s(self, t):
        return current_stream(t.device)

    # ---- utils/utils.py:203-230 warp_tensor (x2 fused) ---------------------------------------------------
    def warp2(self, flow, imgA, imgB=None, want_taps=False):
        """flow [1,2,Hf,Wf], imgA/imgB [1,C,H,W] -> warped A, warped B (and int32 taps [H,W,2])."""
        _f32c(flow), _f32c(imgA)
        _, C, H, W = imgA.shape
        Hf, Wf = flow.shape[-2:]
        outA = torch.empty_like(imgA)
        outB = torch.empty_like(imgB) if imgB is not None else None
        taps = torch.empty((H, W, 2), dtype=torch.int32, device=imgA.device) if want_taps else None
        self.lib.call("zt_warp2_f32", flow, Hf, Wf, imgA, imgB, outA, outB, taps, C, H, W, self._s(imgA))
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
