import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("zero-tig_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle import zt_oracle
    return zt_oracle


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def golden():
    return load_golden


def frames(synth_mod, n, H, W, seed=2):
    return [torch.from_numpy(synth_mod.lowlight_frame(t, H, W, seed)) for t in range(n)]
