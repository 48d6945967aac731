"""ctypes binding of libzerotig_hip.so (the C ABI declared in include/zerotig_hip.h).

The prototypes are parsed from the header itself so the Python side can never drift from the declared ABI.
There is NO fallback: if the shared library is missing, or no HIP device is visible, `get_lib()` raises.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "zerotig_hip.h")
DEFAULT_SO = os.path.join(_HERE, "libzerotig_hip.so")

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "long long": ctypes.c_longlong,
    "size_t": ctypes.c_size_t, "zt_stream_t": ctypes.c_void_p, "unsigned": ctypes.c_uint,
}


def parse_header(path=HEADER):
    """-> {function name: [ctypes argument types]} for every `int zt_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\bint\s+(zt_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2)
        types = []
        for a in [x.strip() for x in args.split(",") if x.strip()]:
            if "*" in a:
                types.append(ctypes.c_void_p)
            else:
                t = re.sub(r"\bconst\b", "", a).strip()
                t = re.sub(r"\s+\w+$", "", t).strip()       # drop the parameter name
                types.append(_CTYPES[t])
        protos[name] = types
    return protos


class Lib:
    """Loaded shared library + checked calls.  `calls` counts launches per entry point (used by tests)."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError("Zero-TIG HIP library not found: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
        self.path = path
        self.dll = ctypes.CDLL(path)
        self.protos = parse_header()
        self.fns = {}
        for name, types in self.protos.items():
            fn = getattr(self.dll, name)           # AttributeError if the ABI symbol is not exported
            fn.argtypes = types
            fn.restype = ctypes.c_int
            self.fns[name] = fn
        self.calls = {}

    def call(self, name, *args):
        """Launch one entry point.  Tensors are passed as their device pointers; everything else goes through ctypes as is."""
        T = torch.Tensor
        rc = self.fns[name](*[a.data_ptr() if isinstance(a, T) else a for a in args])
        self.calls[name] = self.calls.get(name, 0) + 1
        if rc != 0:
            raise RuntimeError("%s failed with code %d%s" % (name, rc, " (invalid argument)" if rc == 1001 else " (hipError_t)"))


_LIB = None


def get_lib():
    """The product's one and only compute backend.  Raises when it cannot run on a HIP device."""
    global _LIB
    if _LIB is None:
        if not torch.cuda.is_available():
            raise RuntimeError("zero-tig_amd needs a HIP device (MI355X / gfx950); none is visible and there is no CPU fallback")
        _LIB = Lib(os.environ.get("ZEROTIG_HIP_LIB", DEFAULT_SO))
    return _LIB


def current_stream(device):
    """Raw hipStream_t of torch's current stream on `device` (None = default stream of the test emulator's CPU tensors)."""
    if device.type == "cuda":
        idx = device.index
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device() if idx is None else idx)
    return None
