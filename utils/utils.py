"""Drop-in for the hot-path functions of the reference `utils/utils.py` (same names and argument meaning) on the HIP kernels,
plus the small host helpers train.py / predict.py use."""
import importlib
import os
import shutil

import numpy as np
import torch

_ops_mod = importlib.import_module("zero-tig_amd.ops")
_lib_mod = importlib.import_module("zero-tig_amd.lib")
_OPS = None


def _ops():
    global _OPS
    if _OPS is None:
        _OPS = _ops_mod.Ops(_lib_mod.get_lib())
    return _OPS


def _prep(x):
    return x.detach().contiguous().float()


def pair_downsampler(img):                                   # utils.py:15-24
    return _ops().pair_down(_prep(img))


def gauss_kernel(kernlen=21, nsig=1, channels=1):            # utils.py:29-39 (the rank-1 factor is what the kernels use)
    assert (kernlen, nsig) == (21, 1)
    t = _ops().gauss_taps()
    return torch.outer(t, t).view(1, 1, kernlen, kernlen).repeat(channels, 1, 1, 1)


def blur(x):                                                 # utils.py:52-58
    return _ops().blur21(_prep(x))


class LocalMean(torch.nn.Module):                            # utils.py:41-50
    def __init__(self, patch_size=5):
        super().__init__()
        assert patch_size == 5
        self.patch_size, self.padding = patch_size, patch_size // 2

    def forward(self, image):
        return _ops().box5_reflect(_prep(image))


def calculate_local_variance(train_noisy):                   # utils.py:66-79
    return _ops().localvar_fwd(_prep(train_noisy), want_D=False)[1]


def warp_tensor(flow, img1, img2):                           # utils.py:203-230 -> (warped, overlap)
    warped, _ = _ops().warp2(_prep(flow), _prep(img1))
    overlap = torch.empty_like(warped)
    _ops().lib.call("zt_axpby_f32", warped, _prep(img2), overlap, 0.5, 0.5, warped.numel(), _lib_mod.current_stream(warped.device))
    return warped, overlap


class InputPadder:                                           # utils.py:233-251 (host-side shape logic)
    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pad_wd = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, *inputs):
        return [torch.nn.functional.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[..., c[0]:c[1], c[2]:c[3]]


def ingest_frame(frame, device):
    """Loader item -> fp32 [1,3,H,W] on `device`: a float frame is copied; a decoded uint8 [1,H0,W0,3] frame (the loaders'
    device-ingest mode) goes through `im.resize((1920, 1080))` + `ToTensor()` of multi_read_data.py:127-132 on the device."""
    if frame.dtype == torch.uint8:
        return _ops().ingest_u8(frame.to(device, non_blocking=True))
    return frame.to(device, non_blocking=True)


def loader_workers(requested):
    """DataLoader worker count: `requested` >= 0 as given; -1 (default of this repo's scripts) = the host cores this process may
    use minus two, at most 12 -- PNG decode of a 1080p frame costs ~30-40 ms of one core, the GPU step 10 ms."""
    if requested is not None and requested >= 0:
        return requested
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(0, min(12, n - 2))


def loader_kwargs(workers):
    """DataLoader arguments of this repo's scripts.  Workers are SPAWNED, not forked: the parent has initialised the GPU by the
    time the loader starts, and a decode worker must not inherit any of that state (it only runs PIL)."""
    extra = dict(persistent_workers=True, prefetch_factor=4, multiprocessing_context="spawn") if workers > 0 else {}
    return dict(num_workers=workers, pin_memory=True, shuffle=False, **extra)


def quantize_u8(tensor, round_half_even=False):
    """[1,3,H,W] in [0,1] on the device -> uint8 [H,W,3] on the device: predict.py:57-61 `save_images` (truncation) or, with
    round_half_even, evals.py:83-84 `np.round(x * 255).astype(np.uint8)`."""
    return _ops().quantize_u8(_prep(tensor), 1 if round_half_even else 0)


def psnr(img, gt):
    """evals.py:83-85: cv2.PSNR(np.round(img * 255), np.round(gt * 255)) computed on the device (exact integer reduction)."""
    return _ops().psnr_u8(_prep(img), _prep(gt))


def count_parameters_in_MB(model):                           # utils.py:81-82
    return sum(int(np.prod(v.size())) for name, v in model.named_parameters() if "auxiliary" not in name) / 1e6


def save(model, model_path):                                 # utils.py:94-95
    torch.save(model.state_dict(), model_path)


def load(model, model_path):                                 # utils.py:98-99
    model.load_state_dict(torch.load(model_path))


def save_checkpoint(model, optimizer, path, epoch=0, step=0):
    """Resume state next to the reference's plain `state_dict` file (utils.py:94-99 has no optimizer / step state): the same 223
    keys under "model" (so `load` / the key-filtered merges of train.py:86-92 and model.py:271-277 still apply to it) plus Adam's
    moments and step count over the flat bucket and the loop position.  Under data parallelism the BatchNorm running statistics
    (per rank: the reference has no SyncBN) are averaged over the ranks first, so every rank resumes from the same file."""
    import torch.distributed as dist
    bn = model.enhance.conv[1]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        for t in (bn.running_mean, bn.running_var):
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            t.div_(dist.get_world_size())
        if dist.get_rank() != 0:
            return
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": int(epoch), "step": int(step)}, path)


def load_checkpoint(model, optimizer, path):
    """-> (epoch, step).  Accepts a resume file of `save_checkpoint` or a plain reference-style `state_dict` (then (0, 0))."""
    ck = torch.load(path, map_location="cpu")
    sd = ck["model"] if isinstance(ck, dict) and "model" in ck and "optimizer" in ck else ck
    md = model.state_dict()
    md.update({k: v for k, v in sd.items() if k in md})
    model.load_state_dict(md)
    if sd is ck:
        return 0, 0
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer"])
    return int(ck.get("epoch", 0)), int(ck.get("step", 0))


def create_exp_dir(path, scripts_to_save=None):              # utils.py:109-118
    os.makedirs(path, exist_ok=True)
    print("Experiment dir : {}".format(path))
    if scripts_to_save is not None:
        os.makedirs(os.path.join(path, "scripts"), exist_ok=True)
        for script in scripts_to_save:
            shutil.copyfile(script, os.path.join(path, "scripts", os.path.basename(script)))


def sequential_judgment(img_path, last_img_path):            # utils.py:145-160
    """New sequence iff the directory differs or the integer file stem is not last + 1 (both paths must exist)."""
    assert os.path.exists(img_path)
    assert os.path.exists(last_img_path)
    d, n = os.path.split(img_path)
    ld, ln = os.path.split(last_img_path)
    return d != ld or int(os.path.splitext(n)[0]) != int(os.path.splitext(ln)[0]) + 1
