"""Deterministic synthetic weights and low-light video frames.

Everything here is a pure function of (name / coordinates, seed): a counter-based
integer hash (splitmix64 finaliser) feeds uniform and Box-Muller normal draws, so the
golden-vector tool (build container), the CPU oracle and the GPU box all see bit-identical
weights and frames without shipping any data file.

Shapes and init distributions follow the reference:
  * `enhance.*` convs  ~ N(0, 0.02), bias 0, BN gamma ~ N(1, 0.02)    (train.py:82-84, model.py:123-130)
  * `denoise_*` convs  ~ U(+-1/sqrt(fan_in)) (torch Conv2d default)     (model.py:15-44)
  * RAFT encoder convs ~ Kaiming-normal fan_out                        (extractor.py:149-156)
  * RAFT update block  ~ U(+-1/sqrt(fan_in))                           (update.py, torch default)
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(x):
    """splitmix64 finaliser on a uint64 array."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def _uniform01(key, n):
    """n float64 draws in (0,1), keyed by a 64-bit integer key."""
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(key) * np.uint64(0x100000001B3)
    bits = _mix64(_mix64(ctr) ^ np.uint64(key))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def name_key(name, seed=0):
    return (zlib.crc32(name.encode()) & 0xFFFFFFFF) ^ ((int(seed) & 0xFFFFFFFF) << 32) ^ 0x5A17


def uniform(name, shape, lo, hi, seed=0):
    n = int(np.prod(shape))
    u = _uniform01(name_key(name, seed), n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(name, shape, mean, std, seed=0):
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = _uniform01(name_key(name + "#a", seed), m)
    u2 = _uniform01(name_key(name + "#b", seed), m)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])[:n]
    return (mean + std * z).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# parameter inventory (names/shapes equal the reference Network.state_dict(), 223 keys)
# --------------------------------------------------------------------------------------

def _conv(prefix, cout, cin, kh, kw):
    return [(prefix + ".weight", (cout, cin, kh, kw)), (prefix + ".bias", (cout,))]


def _bn(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]


def _encoder(prefix, out_dim, norm):
    """BasicEncoder (extractor.py:117-165). norm in {'instance','batch'}; instance norm has no tensors."""
    items = []
    if norm == "batch":
        items += _bn(prefix + ".norm1", 64)
    items += _conv(prefix + ".conv1", 64, 3, 7, 7)
    cin = 64
    for li, (dim, stride) in enumerate([(64, 1), (96, 2), (128, 2)], start=1):
        for bi in range(2):
            p = "%s.layer%d.%d" % (prefix, li, bi)
            s = stride if bi == 0 else 1
            c0 = cin if bi == 0 else dim
            items += _conv(p + ".conv1", dim, c0, 3, 3)
            items += _conv(p + ".conv2", dim, dim, 3, 3)
            if norm == "batch":
                items += _bn(p + ".norm1", dim) + _bn(p + ".norm2", dim)
                if s != 1:
                    items += _bn(p + ".norm3", dim)
            if s != 1:
                items += _conv(p + ".downsample.0", dim, c0, 1, 1)
                if norm == "batch":
                    items += _bn(p + ".downsample.1", dim)   # alias of norm3 (extractor.py:47-48)
        cin = dim
    items += _conv(prefix + ".conv2", out_dim, 128, 1, 1)
    return items


def _update_block(prefix):
    """BasicUpdateBlock (update.py:114-125)."""
    e = prefix + ".encoder"
    g = prefix + ".gru"
    items = []
    items += _conv(e + ".convc1", 256, 324, 1, 1)
    items += _conv(e + ".convc2", 192, 256, 3, 3)
    items += _conv(e + ".convf1", 128, 2, 7, 7)
    items += _conv(e + ".convf2", 64, 128, 3, 3)
    items += _conv(e + ".conv", 126, 256, 3, 3)
    for nm in ("convz1", "convr1", "convq1"):
        items += _conv(g + "." + nm, 128, 384, 1, 5)
    for nm in ("convz2", "convr2", "convq2"):
        items += _conv(g + "." + nm, 128, 384, 5, 1)
    items += _conv(prefix + ".flow_head.conv1", 256, 128, 3, 3)
    items += _conv(prefix + ".flow_head.conv2", 2, 256, 3, 3)
    items += _conv(prefix + ".mask.0", 256, 128, 3, 3)
    items += _conv(prefix + ".mask.2", 576, 256, 1, 1)
    return items


def enhancement_inventory():
    items = []
    items += _conv("enhance.in_conv.0", 64, 9, 3, 3)
    items += _conv("enhance.conv.0", 64, 64, 3, 3) + _bn("enhance.conv.1", 64)
    for i in range(3):                      # three aliases of the one shared block (model.py:60-67)
        items += _conv("enhance.blocks.%d.0" % i, 64, 64, 3, 3) + _bn("enhance.blocks.%d.1" % i, 64)
    items += _conv("enhance.out_conv.0", 3, 64, 3, 3)
    items += _conv("denoise_1.conv1", 48, 3, 3, 3) + _conv("denoise_1.conv2", 48, 48, 3, 3)
    items += _conv("denoise_1.conv3", 3, 48, 1, 1)
    items += _conv("denoise_2.conv1", 48, 12, 3, 3) + _conv("denoise_2.conv2", 48, 48, 3, 3)
    items += _conv("denoise_2.conv3", 6, 48, 1, 1)
    return items


def raft_inventory(prefix="raft"):
    return (_encoder(prefix + ".fnet", 256, "instance") + _encoder(prefix + ".cnet", 256, "batch")
            + _update_block(prefix + ".update_block"))


def inventory():
    return enhancement_inventory() + raft_inventory()


_ALIAS = {}
for _i in range(3):
    for _s in ("0.weight", "0.bias", "1.weight", "1.bias", "1.running_mean", "1.running_var", "1.num_batches_tracked"):
        _ALIAS["enhance.blocks.%d.%s" % (_i, _s)] = "enhance.conv." + _s


def canonical(name):
    """Resolve aliased state-dict keys to the tensor that owns the storage."""
    if name in _ALIAS:
        return _ALIAS[name]
    if ".downsample.1." in name:
        return name.replace(".downsample.1.", ".norm3.")
    return name


def make_state(seed=0, raft_gain=1.0):
    """Full state dict {name: np.ndarray} with the reference's init distributions."""
    out = {}
    for name, shape in inventory():
        cn = canonical(name)
        if cn in out:
            out[name] = out[cn]
            continue
        leaf = cn.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            v = np.zeros((), np.int64)
        elif leaf == "running_mean":
            # non-trivial frozen statistics for the RAFT context encoder (eval-mode BN)
            v = normal(cn, shape, 0.0, 0.05, seed) if cn.startswith("raft.") else np.zeros(shape, np.float32)
        elif leaf == "running_var":
            v = uniform(cn, shape, 0.6, 1.4, seed) if cn.startswith("raft.") else np.ones(shape, np.float32)
        elif len(shape) == 4:
            cout, cin, kh, kw = shape
            if cn.startswith("enhance."):
                v = normal(cn, shape, 0.0, 0.02, seed)
            elif cn.startswith("raft.fnet") or cn.startswith("raft.cnet"):
                v = normal(cn, shape, 0.0, raft_gain * np.sqrt(2.0 / (cout * kh * kw)), seed)
            else:
                b = 1.0 / np.sqrt(cin * kh * kw)
                v = uniform(cn, shape, -b, b, seed)
        elif leaf == "bias":
            if cn.startswith("enhance.") :
                v = np.zeros(shape, np.float32)
            elif ".norm" in cn:
                v = np.zeros(shape, np.float32)
            else:
                # conv bias: torch default U(+-1/sqrt(fan_in)); fan_in from the sibling weight
                wshape = dict(inventory())[cn[:-4] + "weight"]
                b = 1.0 / np.sqrt(wshape[1] * wshape[2] * wshape[3])
                v = uniform(cn, shape, -b, b, seed)
        elif leaf == "weight":           # norm gamma
            v = normal(cn, shape, 1.0, 0.02, seed) if cn.startswith("enhance.") else np.ones(shape, np.float32)
        else:
            raise KeyError(cn)
        out[cn] = v
        out[name] = v
    return out


# --------------------------------------------------------------------------------------
# synthetic low-light clip
# --------------------------------------------------------------------------------------

def _lattice(ix, iy, ch, seed):
    with np.errstate(over="ignore"):
        k = (ix.astype(np.uint64) * np.uint64(73856093)) ^ (iy.astype(np.uint64) * np.uint64(19349663)) \
            ^ np.uint64((ch + 1) * 83492791) ^ np.uint64(seed * 2654435761 + 12345)
    return ((_mix64(k) >> np.uint64(11)).astype(np.float64)) / float(1 << 53)


def clean_frame(t, H, W, seed=2, cell=24.0):
    """Band-limited value-noise texture + 3 sinusoids in [0.1, 0.9]; global translation (+2,+1) px / frame."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xs = xx + 2.0 * t + 4096.0
    ys = yy + 1.0 * t + 4096.0
    out = np.empty((3, H, W), np.float64)
    for c in range(3):
        gx, gy = xs / cell, ys / cell
        x0, y0 = np.floor(gx), np.floor(gy)
        fx, fy = gx - x0, gy - y0
        fx = fx * fx * (3 - 2 * fx)
        fy = fy * fy * (3 - 2 * fy)
        x0i, y0i = x0.astype(np.int64), y0.astype(np.int64)
        v = ((1 - fx) * (1 - fy) * _lattice(x0i, y0i, c, seed) + fx * (1 - fy) * _lattice(x0i + 1, y0i, c, seed)
             + (1 - fx) * fy * _lattice(x0i, y0i + 1, c, seed) + fx * fy * _lattice(x0i + 1, y0i + 1, c, seed))
        s = (np.sin(2 * np.pi * xs / (37.0 + 5 * c)) + np.sin(2 * np.pi * ys / (53.0 - 4 * c))
             + np.sin(2 * np.pi * (xs + ys) / (91.0 + 3 * c))) / 3.0
        out[c] = 0.5 + 0.28 * (2 * v - 1) + 0.12 * s
    return np.clip(out, 0.1, 0.9)


def lowlight_frame(t, H, W, seed=2, gain=0.12, sigma=0.02):
    """Low-light observation of clean_frame(t): gain, additive Gaussian noise, clamp, 8-bit quantisation
    (the reference loader delivers PNG -> ToTensor, multi_read_data.py:127-132)."""
    clean = clean_frame(t, H, W, seed)
    noise = normal("frame%d" % t, (3, H, W), 0.0, sigma, seed).astype(np.float64)
    y = np.clip(gain * clean + noise, 0.0, 1.0)
    return (np.round(y * 255.0) / 255.0).astype(np.float32)[None]
