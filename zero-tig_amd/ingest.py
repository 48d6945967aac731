"""Host half of the device-side frame ingest (reference dataloader/multi_read_data.py:127-132: `im.resize((1920, 1080))` +
`transforms.ToTensor()`): the tables the kernels of csrc/zt_ingest.hip consume.

`pil_bicubic_tables(in_size, out_size)` restates Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (src/libImaging/Resample.c,
BICUBIC filter a = -0.5, support 2, 8-bit path with PRECISION_BITS = 22) in the same double-precision operation order, so the
integer coefficients are the ones Pillow itself would use; the kernels then do Pillow's integer arithmetic.  tests/test_ingest.py
checks the result against PIL byte for byte."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_tables(in_size, out_size):
    """-> (coef int32 [out_size, ksize], bounds int32 [out_size, 2] = (first source index, tap count), ksize)."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    coef = np.zeros((out_size, ksize), np.int32)
    bounds = np.zeros((out_size, 2), np.int32)
    ss = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            coef[xx, x] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)      # C's (int) truncates toward zero
        bounds[xx] = (xmin, xmax)
    return coef, bounds, ksize


def to_tensor_lut():
    """ToTensor's `byte / 255` as a table: float32(k) / float32(255), correctly rounded by the host's IEEE division (what torch's
    CPU `div` computes), so the device result is bit-identical."""
    return (np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.float32)
