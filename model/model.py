"""Drop-in for the reference `model/model.py` (train.py:12 `from model.model import *`, predict.py:10
`from model.model import Finetunemodel`): same class names, constructor arguments, attributes and state-dict keys;
the compute runs in libzerotig_hip.so (zero-tig_amd/)."""
import importlib

import numpy as np  # noqa: F401  (train.py's save_images relies on `np` arriving through the star import, train.py:57-62)
import torch  # noqa: F401
import torch.nn as nn  # noqa: F401

_net = importlib.import_module("zero-tig_amd.network")
Network = _net.Network
Finetunemodel = _net.Finetunemodel
Enhancer = _net.Enhancer
Denoise_1 = _net.Denoise_1
Denoise_2 = _net.Denoise_2

from loss import LossFunction, TextureDifference  # noqa: E402,F401
from utils.utils import blur, pair_downsampler, warp_tensor, InputPadder  # noqa: E402,F401
from model.RAFT.raft import RAFT  # noqa: E402,F401
