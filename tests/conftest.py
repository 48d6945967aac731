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


# ---------------------------------------------------------------------------------------------------------
# kernel back-ends for the parity tests: "emu" = the product's .hip sources compiled for the host against
# tests/hipemu (CPU tests); "hip" = the real libzerotig_hip.so on the MI355X (tests marked gpu).
# ---------------------------------------------------------------------------------------------------------
import subprocess


@pytest.fixture(scope="session")
def emu_ops():
    emu_dir = os.path.join(ROOT, "tests", "hipemu")
    r = subprocess.run(["make", "-C", emu_dir, "-j8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    pkg = importlib.import_module("zero-tig_amd")
    lib_mod = importlib.import_module("zero-tig_amd.lib")
    ops_mod = importlib.import_module("zero-tig_amd.ops")
    return ops_mod.Ops(lib_mod.Lib(os.path.join(emu_dir, "libzerotig_emu.so"))), torch.device("cpu")


@pytest.fixture(scope="session")
def hip_ops():
    lib_mod = importlib.import_module("zero-tig_amd.lib")
    ops_mod = importlib.import_module("zero-tig_amd.ops")
    return ops_mod.Ops(lib_mod.get_lib()), torch.device("cuda:0")


BACKENDS = [pytest.param("emu", id="emu"), pytest.param("hip", id="hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    name = request.param
    ops, dev = request.getfixturevalue(name + "_ops")
    return ops, dev, name
