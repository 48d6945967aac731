"""Drop-in for the reference `loss.py` (model.py:5 `from loss import LossFunction, TextureDifference`).

Values only: in training the gradient of the loss is produced together with its value by the fused HIP loss kernels inside
`Network._loss` (zero-tig_amd/engine.py).  These classes expose the same call signatures for code that evaluates the
loss terms on their own."""
import importlib

import torch
import torch.nn as nn

_eng = importlib.import_module("zero-tig_amd.engine")
_ops_mod = importlib.import_module("zero-tig_amd.ops")
_lib_mod = importlib.import_module("zero-tig_amd.lib")

EPS = 1e-9


def _ops(ops):
    return ops if ops is not None else _ops_mod.Ops(_lib_mod.get_lib())


class LossFunction(nn.Module):
    def __init__(self, is_WB, ops=None):
        super().__init__()
        self.is_WB = is_WB
        self.__dict__["_ops"] = ops

    def forward(self, input, L_pred1, L_pred2, L2, s2, s21, s22, H2, H11, H12, H13, s13, H14, s14, H3, s3, H3_pred, H4_pred,
                L_pred1_L_pred2_diff, H3_denoised1_H3_denoised2_diff, H2_blur, H3_blur):
        """loss.py:23-78; returns the scalar loss (no autograd graph)."""
        loss, _ = _eng.loss_value(_ops(self._ops), self.is_WB, input, L_pred1, L_pred2, L2, s2, s21, s22, H2, H11, H12, H3, s3,
                                  H3_pred, H4_pred, H3_denoised1_H3_denoised2_diff, H2_blur, H3_blur)
        return loss.reshape(())


class TextureDifference(nn.Module):
    def __init__(self, patch_size=5, constant_C=1e-5, threshold=0.975, ops=None):
        super().__init__()
        assert (patch_size, constant_C, threshold) == (5, 1e-5, 0.975), "kernel is specialised to the reference constants (loss.py:100)"
        self.__dict__["_ops"] = ops

    def forward(self, image1, image2):
        return _ops(self._ops).texture_mask(image1.contiguous().float(), image2.contiguous().float())


class SmoothLoss(nn.Module):
    def __init__(self, ops=None):
        super().__init__()
        self.sigma = 10
        self.__dict__["_ops"] = ops

    def forward(self, input, output):
        return _eng.smooth_tv_values(_ops(self._ops), input, output)[0]


class L_TV(nn.Module):
    def __init__(self, TVLoss_weight=1, ops=None):
        super().__init__()
        self.TVLoss_weight = TVLoss_weight
        self.__dict__["_ops"] = ops

    def forward(self, x):
        return self.TVLoss_weight * _eng.smooth_tv_values(_ops(self._ops), x, x)[1]
