"""Import the upstream reference (read-only, /root/reference) in THIS build container only.

Used by tools/make_golden.py to produce committed fixtures under tests/golden/.  Nothing here
travels to the GPU box in a usable form (the reference tree does not exist there) and nothing
from the reference is copied: we only register three shims before importing it:

  1. an empty `cv2` module (imported at model.py:11 / utils.py:8, used only by dead code),
  2. `torchvision.transforms.functional.equalize` = the build's integer restatement
     (oracle/zt_oracle.py:equalize_u8; torchvision is not installed here -> "parity unpinned"
     at that one third-party boundary),
  3. `torch.Tensor.cuda` -> identity (utils.py:31, loss.py:182/184 hard-code .cuda()).
"""
import argparse
import os
import sys
import types

import torch

REF = os.environ.get("ZT_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def import_reference():
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree %s not present (goldens can only be generated in the build container)" % REF)
    sys.path.insert(0, ROOT)
    from oracle import zt_oracle

    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvf.equalize = zt_oracle.equalize_u8
    tv.transforms = tvt
    tvt.functional = tvf
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tvt
    sys.modules["torchvision.transforms.functional"] = tvf
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    # the reference's top-level module names (model, loss, utils) collide with this repo's drop-in
    # modules of the same names: make sure the reference wins inside this process.
    for name in list(sys.modules):
        if name in ("model", "loss", "utils") or name.startswith(("model.", "utils.")):
            del sys.modules[name]
    sys.path.insert(0, REF)
    import model.model as ref_model       # noqa: E402
    import loss as ref_loss               # noqa: E402
    import utils.utils as ref_utils       # noqa: E402
    import model.RAFT.corr as ref_corr    # noqa: E402
    assert ref_model.__file__.startswith(REF), ref_model.__file__
    return types.SimpleNamespace(model=ref_model, loss=ref_loss, utils=ref_utils, corr=ref_corr)


def make_args(dataset="RLV", of_scale=3):
    return argparse.Namespace(dataset=dataset, of_scale=of_scale)
