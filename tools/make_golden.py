#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself (imported read-only from /root/reference)
on the build's deterministic synthetic weights and frames.  Build-container only; commit the outputs.

Fixtures are data (inputs are regenerated from zero-tig_amd/synth.py, only expected outputs are stored).
Cases: G1/G2 new-sequence forward+loss+grads, G3/G4 two-frame RAFT+warp sequences,
G5 underwater (is_WB) variant, G6 op-level vectors, G7 three optimizer steps, G8 BN bookkeeping,
G9 Finetunemodel (inference twin).
"""
import importlib
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
synth = importlib.import_module("zero-tig_amd.synth")
from ref_import import import_reference, make_args   # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.manual_seed(0)
torch.set_num_threads(8)


def np_(t):
    return t.detach().cpu().numpy()


def load_weights(net, seed):
    st = synth.make_state(seed)
    sd = net.state_dict()
    assert set(sd.keys()) == set(st.keys()), (set(sd.keys()) ^ set(st.keys()))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(st[k].shape), (k, sd[k].shape, st[k].shape)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()}, strict=True)
    return st


def frames(n, H, W, seed=2):
    return [torch.from_numpy(synth.lowlight_frame(t, H, W, seed)) for t in range(n)]


def trainable(net):
    return [(n, p) for n, p in net.named_parameters() if p.requires_grad]


def run_loss_with_grads(net, x, new_seq):
    net.zero_grad(set_to_none=True)
    net.is_new_seq = new_seq
    loss = net._loss(x.clone())
    loss.backward()
    return loss.detach(), {n: np_(p.grad) for n, p in trainable(net)}


def hook_raft(net, store):
    orig = net.raft.forward

    def fwd(*a, **k):
        lo, up = orig(*a, **k)
        store["flow_low"], store["flow_up"] = np_(lo), np_(up)
        store["raft_img1"], store["raft_img2"] = np_(a[0]), np_(a[1])
        return lo, up
    net.raft.forward = fwd


def case_newseq(ref, name, H, W, dataset, seed):
    net = ref.model.Network(make_args(dataset, 3))
    load_weights(net, seed)
    net.train()
    x = frames(1, H, W)[0]
    net.is_new_seq = True
    outs = net(x.clone())
    d = {"out%02d" % i: np_(o) for i, o in enumerate(outs)}
    # G8: BN bookkeeping after exactly one training-mode forward
    sd = net.state_dict()
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        d["bn_" + k] = np_(sd["enhance.conv.1." + k])
    # fresh net for loss + grads so BN running stats do not matter (they do not enter train-mode math anyway)
    net2 = ref.model.Network(make_args(dataset, 3))
    load_weights(net2, seed)
    net2.train()
    loss, grads = run_loss_with_grads(net2, x, True)
    d["loss"] = np_(loss)
    for n, g in grads.items():
        d["grad:" + n] = g
    d["meta"] = np.array([H, W, seed, 3], np.int64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "loss", float(loss), "bytes", os.path.getsize(os.path.join(OUT, name + ".npz")))


def case_sequence(ref, name, H, W, of_scale, seed, nframes=2):
    net = ref.model.Network(make_args("RLV", of_scale))
    load_weights(net, seed)
    net.train()
    store = {}
    hook_raft(net, store)
    xs = frames(nframes, H, W)
    d = {"meta": np.array([H, W, seed, of_scale], np.int64)}
    for t, x in enumerate(xs):
        loss, grads = run_loss_with_grads(net, x, t == 0)
        d["loss%d" % t] = np_(loss)
        if t == nframes - 1:
            for n, g in grads.items():
                d["grad:" + n] = g
    # last frame tensors (recompute forward is not possible without disturbing state: read attributes instead)
    d["flow_low"], d["flow_up"] = store["flow_low"], store["flow_up"]
    d["raft_img1"], d["raft_img2"] = store["raft_img1"], store["raft_img2"]
    d["wpH"], d["wps"] = np_(net.last_H3_wp), np_(net.last_s3_wp)
    d["last_H3"], d["last_s3"] = np_(net.last_H3), np_(net.last_s3)
    # tap indices of the reference warp (integer contract): floor of the source coordinates ATen derives from the
    # reference's own grid (utils.py:203-225), using ATen's CPU arithmetic ix = fma(g+1, W/2, -0.5)
    flow_up = torch.from_numpy(store["flow_up"])
    Hd, Wd = xs[0].shape[-2:]
    B, _, Hf, Wf = flow_up.shape
    gy, gx = torch.meshgrid(torch.arange(Hf, dtype=torch.float32), torch.arange(Wf, dtype=torch.float32), indexing="ij")
    mx = torch.nn.functional.interpolate(((gx[None] - flow_up[:, 0]) * (float(Hd) / Hf)).unsqueeze(1), (Hd, Wd), mode="bilinear")
    my = torch.nn.functional.interpolate(((gy[None] - flow_up[:, 1]) * (float(Wd) / Wf)).unsqueeze(1), (Hd, Wd), mode="bilinear")
    gxn, gyn = mx / ((Wd - 1) / 2) - 1, my / ((Hd - 1) / 2) - 1
    ix = ((gxn + 1).double() * (Wd / 2.0) - 0.5).float()
    iy = ((gyn + 1).double() * (Hd / 2.0) - 0.5).float()
    d["warp_x0"], d["warp_y0"] = np_(torch.floor(ix)).astype(np.int32), np_(torch.floor(iy)).astype(np.int32)
    # probe that these really are ATen's taps: sampling the images x and y reproduces ix, iy to rounding
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "losses", [float(d["loss%d" % t]) for t in range(nframes)], "flow_up absmax", float(np.abs(d["flow_up"]).max()),
          "bytes", os.path.getsize(os.path.join(OUT, name + ".npz")))


def case_ops(ref, name, seed=7):
    U, L = ref.utils, ref.loss
    g = {}
    x = torch.from_numpy(synth.uniform("ops.x", (1, 3, 40, 56), 0.0, 1.0, seed))
    y = torch.from_numpy(synth.uniform("ops.y", (1, 3, 40, 56), 0.0, 1.0, seed))
    g["x"], g["y"] = np_(x), np_(y)
    a, b = U.pair_downsampler(x)
    g["pd1"], g["pd2"] = np_(a), np_(b)
    g["blur"] = np_(U.blur(x))
    g["gauss21"] = np_(U.gauss_kernel(21, 1, 1))[0, 0]
    g["localmean"] = np_(U.LocalMean(5)(x))
    g["localvar"] = np_(U.calculate_local_variance(x))
    td = L.TextureDifference()
    # smooth image pair so the >0.975 mask has both classes
    ramp = torch.linspace(0.0, 0.5, x.shape[-1]).view(1, 1, 1, -1)
    xs = 0.5 * x + 0.5 * U.blur(x)
    ys = xs + ramp * (y - 0.5)
    g["tex_in1"], g["tex_in2"] = np_(xs), np_(ys)
    g["texmask"] = np_(td(xs, ys))
    s1, s2 = td.local_stddev(td.rgb_to_gray(xs)), td.local_stddev(td.rgb_to_gray(ys))
    g["texratio"] = np_(2 * s1 * s2 / (s1 ** 2 + s2 ** 2 + 1e-5))
    g["smooth"] = np_(L.SmoothLoss()(x * 0.2, y))
    g["ycc"] = np_(L.SmoothLoss().rgb2yCbCr(x * 0.2))
    g["tv"] = np_(L.L_TV()(y))
    # warp: flow at 24x40 (scales 40/24 vs 56/40 differ -> exercises the swapped scales)
    flow = torch.from_numpy(synth.normal("ops.flow", (1, 2, 40, 64), 0.0, 1.5, seed))
    wimg = torch.from_numpy(synth.uniform("ops.wimg", (1, 3, 72, 120), 0.0, 1.0, seed))
    g["warp_flow"], g["warp_img"] = np_(flow), np_(wimg)
    g["warp_out"] = np_(U.warp_tensor(flow, wimg, wimg)[0])
    # correlation volume + lookup
    f1 = torch.from_numpy(synth.normal("ops.f1", (1, 256, 16, 24), 0.0, 1.0, seed))
    f2 = torch.from_numpy(synth.normal("ops.f2", (1, 256, 16, 24), 0.0, 1.0, seed))
    cb = ref.corr.CorrBlock(f1, f2, radius=4)
    for i, c in enumerate(cb.corr_pyramid):
        g["corr_pyr%d" % i] = np_(c) if i else np_(c)[::7]        # level 0: every 7th source pixel
    coords = U.coords_grid(1, 16, 24, "cpu") + torch.from_numpy(synth.normal("ops.dc", (1, 2, 16, 24), 0.0, 2.5, seed))
    g["lookup_coords"] = np_(coords)
    g["lookup_out"] = np_(cb(coords))
    # equalize (build's restatement; recorded so the HIP kernel and oracle share one vector)
    import torchvision.transforms.functional as tvf
    e_in = (torch.from_numpy(synth.uniform("ops.eq", (1, 3, 30, 44), 0.0, 1.0, seed)) ** 3 * 200).to(torch.uint8)
    g["eq_in"], g["eq_out"] = np_(e_in), np_(tvf.equalize(e_in))
    # one update-block step + convex upsample with the synthetic RAFT weights
    net = ref.model.Network(make_args("RLV", 1))
    load_weights(net, 3)
    ub = net.raft.update_block
    h = torch.tanh(torch.from_numpy(synth.normal("ops.h", (1, 128, 16, 24), 0.0, 1.0, seed)))
    inp = torch.relu(torch.from_numpy(synth.normal("ops.inp", (1, 128, 16, 24), 0.0, 1.0, seed)))
    corr = cb(coords)
    fl = coords - U.coords_grid(1, 16, 24, "cpu")
    with torch.no_grad():
        h2, mask, dfl = ub(h, inp, corr, fl)
        up = net.raft.upsample_flow(fl + dfl, mask)
    g["ub_net"], g["ub_mask"], g["ub_dflow"], g["ub_up"] = np_(h2), np_(mask)[:, ::9], np_(dfl), np_(up)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **g)
    print(name, "bytes", os.path.getsize(os.path.join(OUT, name + ".npz")), "mask mean", float(g["texmask"].mean()))


def case_adam(ref, name, H, W, seed, steps=3):
    net = ref.model.Network(make_args("RLV", 1))
    load_weights(net, seed)
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=3e-4)
    xs = frames(steps, H, W)
    d = {"meta": np.array([H, W, seed, 1], np.int64)}
    for t, x in enumerate(xs):
        net.is_new_seq = (t == 0)
        opt.zero_grad()
        loss = net._loss(x.clone())
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(net.parameters(), 5)
        opt.step()
        d["loss%d" % t], d["gnorm%d" % t] = np_(loss), np_(gn)
    for n, p in trainable(net):
        d["w:" + n] = np_(p)
    sd = net.state_dict()
    for k in ("running_mean", "running_var", "num_batches_tracked"):
        d["bn_" + k] = np_(sd["enhance.conv.1." + k])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "losses", [float(d["loss%d" % t]) for t in range(steps)], "gnorm", [float(d["gnorm%d" % t]) for t in range(steps)])


def case_finetune(ref, name, H, W, seed):
    net = ref.model.Network(make_args("RLV", 1))
    load_weights(net, seed)
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "ck.pt")
        torch.save(net.state_dict(), ck)
        orig_load = torch.load
        torch.load = lambda f, map_location=None, **k: orig_load(f, map_location="cpu", **k)
        try:
            args = make_args("RLV", 1)
            args.model_pretrain = ck
            ft = ref.model.Finetunemodel(args)
        finally:
            torch.load = orig_load
    # Finetunemodel builds its own (random) RAFT: overwrite with the synthetic one
    ft.raft.load_state_dict({k[5:]: v for k, v in net.state_dict().items() if k.startswith("raft.")})
    ft.eval()
    xs = frames(2, H, W)
    d = {"meta": np.array([H, W, seed, 1], np.int64)}
    with torch.no_grad():
        for t, x in enumerate(xs):
            ft.is_new_seq = (t == 0)
            H2, H3, s3 = ft(x.clone())
            d["H2_%d" % t], d["H3_%d" % t], d["s3_%d" % t] = np_(H2), np_(H3), np_(s3)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "bytes", os.path.getsize(os.path.join(OUT, name + ".npz")))


def main():
    os.makedirs(OUT, exist_ok=True)
    ref = import_reference()
    which = sys.argv[1:] or ["g12", "g3", "g4", "g5", "g6", "g7", "g9"]
    with torch.no_grad():
        pass
    if "g12" in which:
        case_newseq(ref, "g12_newseq_rlv_48x64", 48, 64, "RLV", 1)
    if "g5" in which:
        case_newseq(ref, "g5_newseq_wb_48x64", 48, 64, "underwater", 1)
    if "g3" in which:
        case_sequence(ref, "g3_seq_128x160", 128, 160, 1, 1)
    if "g4" in which:
        case_sequence(ref, "g4_seq_132x164", 132, 164, 1, 1)
    if "g6" in which:
        case_ops(ref, "g6_ops")
    if "g7" in which:
        case_adam(ref, "g7_adam_128x160", 128, 160, 1)
    if "g9" in which:
        case_finetune(ref, "g9_finetune_128x160", 128, 160, 1)


if __name__ == "__main__":
    main()
