"""Minimal, API-compatible frame loaders (reference dataloader/multi_read_data.py): items are
(tensor[3,1080,1920] in [0,1], img_name, img_path, last_img_path), frames in temporal order.  PIL only (no torchvision).

Device-ingest mode (`args.device_ingest = True`, set by this repo's train.py / predict.py / evals.py): the item's first element is
the DECODED frame, uint8 [H0,W0,3], and nothing else happens on the host -- `resize((1920, 1080))` and `ToTensor()` run on the
GPU, bit-identical (zero-tig_amd/csrc/zt_ingest.hip, `Ops.ingest_u8`), so that DataLoader workers only decode and the frame
crosses PCIe as bytes (6 MB at 1080p instead of 25 MB)."""
import glob
import os

import numpy as np
import torch
import torch.utils.data
from PIL import Image


def _natural(files):
    def key(f):
        stem = os.path.splitext(os.path.basename(f))[0]
        return (0, int(stem)) if stem.isdigit() else (1, stem)
    return sorted(files, key=key)


class _Base(torch.utils.data.Dataset):
    size = (1920, 1080)

    def initialize(self, args, task):
        self.args, self.task = args, task
        self.low_img_dir = args.lowlight_images_path
        assert os.path.exists(self.low_img_dir), "Input directory does not exist!"
        self.files = self.list_files(self.low_img_dir, task)
        assert self.files, "No input data."
        self.device_ingest = bool(getattr(args, "device_ingest", False))

    def load(self, f):
        im = Image.open(f).convert("RGB")
        if self.device_ingest:                                       # decode only; resize + ToTensor run on the device
            return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy())
        im = im.resize(self.size)                                    # multi_read_data.py:127-132 (PIL default filter)
        return torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)

    def __getitem__(self, i):
        # the reference keeps "the path fetched last" as per-process state (multi_read_data.py:137-140), which under an in-order
        # walk is files[i - 1] (files[0] for the first item) -- stated by index here so that DataLoader workers agree with it
        path = self.files[i]
        last = self.files[i - 1] if i > 0 else self.files[0]
        return self.load(path), os.path.splitext(os.path.basename(path))[0], path, last

    def __len__(self):
        return len(self.files)


class RLVDataLoader(_Base):
    """BVI-RLV layout: <root>/<task>_list.txt names scene folders; frames under input/<scene>/low_light_{10,20}/*.png."""

    def name(self):
        return "BVI-RLV"

    def list_files(self, root, task):
        assert task in ("train", "test"), "Invalid phase: " + str(task)
        out = []
        with open(os.path.join(root, task + "_list.txt")) as fh:
            for scene in [l.strip() for l in fh if l.strip()]:
                for sub in ("low_light_10", "low_light_20"):
                    out += _natural(glob.glob(os.path.join(root, "input", scene, sub, "*.png")))
        return out


def _list_file(root, name):
    with open(os.path.join(root, name)) as fh:
        lines = [l.strip() for l in fh if l.strip()]
    assert lines, "No input data."
    return lines


class DidDataloader(_Base):
    """DID layout (reference multi_read_data.py:144-199): <root>/<task>_list.txt names folders; frames are
    input/<folder>/*.{jpg,png}, sorted by their integer file stem inside each folder."""

    def name(self):
        return "DID"

    def list_files(self, root, task):
        assert task in ("train", "test"), "Invalid phase: " + str(task)
        out = []
        for folder in _list_file(root, task + "_list.txt"):
            out += _natural(glob.glob(os.path.join(root, "input", folder, "*.jpg")) + glob.glob(os.path.join(root, "input", folder, "*.png")))
        return out


class SDSDDataloader(_Base):
    """SDSD layout (reference multi_read_data.py:202-318): <root>/{indoor,outdoor}/{indoor,outdoor}_png/<pair>/ with the pair
    directories of a phase listed in <root>/sdsd_{in,out}_{train,test}.txt; ONE low-light image per pair directory (the first
    file whose path contains neither 'gt' nor 'normal', else the first file); each subset sorted by integer file stem."""

    def name(self):
        return "SDSD"

    def list_files(self, root, task):
        assert task in ("train", "test"), "Invalid phase: " + str(task)
        out = []
        for subset, prefix in (("indoor", "in"), ("outdoor", "out")):
            lst = "sdsd_%s_%s.txt" % (prefix, task)
            png = os.path.join(root, subset, subset + "_png")
            if not (os.path.isdir(os.path.join(root, subset)) and os.path.exists(os.path.join(root, lst)) and os.path.isdir(png)):
                continue
            sub = []
            for pair in _list_file(root, lst):
                d = os.path.join(png, pair)
                if not os.path.isdir(d):
                    continue
                files = sorted(glob.glob(os.path.join(d, "*.png"))) + sorted(glob.glob(os.path.join(d, "*.jpg")))
                low = [f for f in files if "gt" not in f.lower() and "normal" not in f.lower()]
                if low or files:
                    sub.append((low or files)[0])
            out += _natural(sub)
        return out


class DefaultDataset(_Base):
    """Any other dataset name (e.g. `underwater`): every file below the root, hidden files skipped, sorted by integer file stem
    (reference multi_read_data.py:29-71 -- which is truncated upstream: no `name()` / `__len__`, so `--dataset underwater`
    crashes there at create_data.py:16; this is the loader that code was evidently meant to be)."""

    def name(self):
        return "Default"

    def list_files(self, root, task):
        out = []
        for r, _, names in os.walk(root):
            out += [os.path.join(r, n) for n in names if not n.startswith(".") and n.lower().endswith((".png", ".jpg", ".jpeg", ".bmp"))]
        return _natural(out)


class FolderSequenceDataset(_Base):
    """Generic layout: <root>/<sequence>/*.png (every sub-folder is one sequence)."""

    def name(self):
        return "FolderSequence"

    def list_files(self, root, task):
        out = []
        for d in sorted(glob.glob(os.path.join(root, "*"))):
            if os.path.isdir(d):
                out += _natural(glob.glob(os.path.join(d, "*.png")))
        return out or _natural(glob.glob(os.path.join(root, "*.png")))
