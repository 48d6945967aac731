"""SURVEY 8(f)-1, device-side ingest: `im.resize((1920, 1080))` (PIL BICUBIC, 8-bit two-pass) + `ToTensor()` of the reference
loader (dataloader/multi_read_data.py:127-132) on the device, bit-identical to the host libraries; the loaders' uint8 mode."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import ROOT


def _ref_pil(a, size):
    """what the reference loader computes on the host: PIL resize (default filter) + ToTensor"""
    im = Image.fromarray(a).convert("RGB").resize(size)
    return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)[None]


CASES = [((27, 48), (96, 54)), ((60, 88), (64, 48)), ((37, 53), (52, 80)), ((54, 96), (96, 54)), ((108, 192), (96, 54))]


@pytest.mark.parametrize("hw,size", CASES)
def test_oracle_resize_pinned_to_pil(hw, size):
    from oracle import pil_resize
    a = np.random.default_rng(hw[0]).integers(0, 256, hw + (3,), dtype=np.uint8)
    assert np.array_equal(pil_resize.pil_resize_bicubic_u8(a, size), np.asarray(Image.fromarray(a).resize(size)))
    assert torch.equal(torch.from_numpy(pil_resize.load_frame(a, size))[None], _ref_pil(a, size))


@pytest.mark.parametrize("hw,size", CASES)
def test_ingest_matches_pil_and_oracle(backend, hw, size):
    ops, dev, _ = backend
    from oracle import pil_resize
    a = np.random.default_rng(7 + hw[1]).integers(0, 256, hw + (3,), dtype=np.uint8)
    got = ops.ingest_u8(torch.from_numpy(a).to(dev), size=size).cpu()
    assert torch.equal(got, torch.from_numpy(pil_resize.load_frame(a, size))[None])
    assert torch.equal(got, _ref_pil(a, size))


@pytest.mark.gpu
@pytest.mark.parametrize("hw", [(1080, 1920), (1440, 2560), (270, 480), (2160, 3840), (1000, 1504)], ids=lambda v: "%dx%d" % v)
def test_ingest_full_size_bit_exact(hip_ops, hw):
    """BASELINE size: any decoded frame -> [1,3,1080,1920] exactly as PIL + ToTensor deliver it (1080p input: no resampling, as PIL)."""
    ops, dev = hip_ops
    a = np.random.default_rng(hw[0]).integers(0, 256, hw + (3,), dtype=np.uint8)
    got = ops.ingest_u8(torch.from_numpy(a).to(dev)).cpu()
    assert got.shape == (1, 3, 1080, 1920)
    assert torch.equal(got, _ref_pil(a, (1920, 1080)))


def test_loader_uint8_mode_and_last_path(tmp_path):
    """The loaders' device-ingest mode: items carry the DECODED frame (uint8 HWC, any size) instead of the resized float tensor,
    and `last_img_path` no longer depends on per-process state, so DataLoader workers return what a sequential walk returns."""
    import argparse
    import sys
    sys.path.insert(0, ROOT)
    from dataloader.create_data import CreateDataset
    root = tmp_path / "RLV"
    d = root / "input" / "S01" / "low_light_10"
    d.mkdir(parents=True)
    rng = np.random.default_rng(0)
    for t in range(5):
        Image.fromarray(rng.integers(0, 256, (54, 96, 3), dtype=np.uint8)).save(str(d / ("%05d.png" % (t + 1))))
    (root / "train_list.txt").write_text("S01\n")
    args = argparse.Namespace(lowlight_images_path=str(root), dataset="RLV", device_ingest=True)
    ds = CreateDataset(args, task="train")
    seq = [ds[i] for i in range(len(ds))]
    assert all(it[0].dtype == torch.uint8 and tuple(it[0].shape) == (54, 96, 3) for it in seq)
    dl = torch.utils.data.DataLoader(ds, batch_size=1, shuffle=False, num_workers=2)
    got = list(dl)
    assert [g[3][0] for g in got] == [s[3] for s in seq] == [seq[0][2]] + [s[2] for s in seq[:-1]]
    assert all(torch.equal(g[0][0], s[0]) for g, s in zip(got, seq))
    # the reference-compatible float mode delivers the same frame PIL + ToTensor make of it
    args.device_ingest = False
    it = CreateDataset(args, task="train")[2]
    assert it[0].dtype == torch.float32 and tuple(it[0].shape) == (3, 1080, 1920)
    assert torch.equal(it[0][None], _ref_pil(np.asarray(Image.open(seq[2][2])), (1920, 1080)))
