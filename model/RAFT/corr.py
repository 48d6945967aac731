"""Drop-in for the reference `model/RAFT/corr.py` CorrBlock (corr.py:12-60) and the `alt_cuda_corr.forward` seam (corr.py:86):
all-pairs correlation volume as an MFMA 1x1 convolution, 3 pooled levels, fused 4-level 9x9 bilinear lookup."""
import importlib

import torch

_ops_mod = importlib.import_module("zero-tig_amd.ops")
_lib_mod = importlib.import_module("zero-tig_amd.lib")


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, ops=None):
        assert num_levels == 4 and radius == 4 and fmap1.shape[0] == 1, "RAFT-basic geometry (raft.py:30-31)"
        self.ops = ops if ops is not None else _ops_mod.Ops(_lib_mod.get_lib())
        _, C, h, w = fmap1.shape
        self.h, self.w = h, w
        npx = h * w
        pitch = (npx + 15) // 16 * 16
        f1 = fmap1.float().permute(0, 2, 3, 1).contiguous()
        f2 = torch.zeros((1, C, pitch), dtype=torch.float32, device=fmap1.device)
        f2[:, :, :npx] = fmap2.float().reshape(1, C, npx)
        self.corr0 = self.ops.conv2d(_ops_mod.CV(f1), f2, None, npx, 1, 1, alpha=1.0 / float(C) ** 0.5)
        self.levels = self.ops.corr_pyramid(self.corr0, h, w)
        self.corr_pyramid = [self.corr0[..., :npx].reshape(npx, 1, h, w)] + [l.unsqueeze(1) for l in self.levels]

    def __call__(self, coords):
        c = coords[0].permute(1, 2, 0).reshape(-1, 2).contiguous().float()
        out = self.ops.corr_lookup(self.corr0, self.levels, self.h, self.w, c)
        return out.permute(0, 3, 1, 2).contiguous()
