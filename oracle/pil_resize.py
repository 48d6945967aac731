"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reference loader's host-side frame preparation
(/root/reference/dataloader/multi_read_data.py:127-132: `Image.open(f).convert('RGB').resize((1920, 1080))` + `transforms.ToTensor()`).

The arithmetic lives in two third-party libraries: Pillow (`Image.resize`, default filter for RGB = BICUBIC; 8-bit two-pass
resampler of src/libImaging/Resample.c) and torchvision's `ToTensor` (uint8 HWC -> float CHW, `.div(255)`).  Pillow IS installed
in this image (12.2.0), so this restatement is PINNED: tests/test_ingest.py compares it with `PIL.Image.resize` byte for byte on
random images (up-, down-scaling, odd sizes).  Only tests/ may import this module; the product's tables come from
zero-tig_amd/ingest.py and its arithmetic from csrc/zt_ingest.hip."""
import math

import numpy as np

PB = 22          # PRECISION_BITS = 32 - 8 - 2 (Resample.c)


def _bicubic(x):           # Resample.c bicubic_filter, a = -0.5
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def _coeffs(n_in, n_out):  # Resample.c precompute_coeffs + normalize_coeffs_8bpc
    scale = fs = n_in / n_out
    fs = max(fs, 1.0)
    support = 2.0 * fs
    out = []
    for xx in range(n_out):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), n_in) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * (1.0 / fs)) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        k = []
        for v in w:
            v = v / ww if ww != 0.0 else v
            k.append(int(-0.5 + v * (1 << PB)) if v < 0 else int(0.5 + v * (1 << PB)))
        out.append((xmin, np.array(k, dtype=np.int64)))
    return out


def _pass(a, n_out, axis):  # ImagingResampleHorizontal_8bpc / Vertical_8bpc
    a = np.moveaxis(a, axis, 0).astype(np.int64)
    out = np.empty((n_out,) + a.shape[1:], np.uint8)
    for xx, (xmin, k) in enumerate(_coeffs(a.shape[0], n_out)):
        acc = (1 << (PB - 1)) + np.tensordot(k, a[xmin:xmin + len(k)], axes=(0, 0))
        out[xx] = np.clip(acc >> PB, 0, 255)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic_u8(a, size):
    """a: uint8 [H,W,C]; size = (W, H) as PIL takes it.  Horizontal pass first, then vertical; a pass is skipped when that
    dimension does not change (PIL returns a plain copy when neither does)."""
    W, H = size
    if a.shape[1] != W:
        a = _pass(a, W, 1)
    if a.shape[0] != H:
        a = _pass(a, H, 0)
    return a


def load_frame(a, size=(1920, 1080)):
    """decoded uint8 [H,W,3] -> float32 [3,H',W'] in [0,1] exactly as the reference loader delivers it."""
    r = pil_resize_bicubic_u8(np.asarray(a, dtype=np.uint8), size)
    return np.ascontiguousarray(np.transpose(r, (2, 0, 1))).astype(np.float32) / np.float32(255.0)
