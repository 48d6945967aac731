"""The drop-in boundary: ABI symbols, header/ctypes agreement, reference module paths and class surface (CPU),
and the drop-in functions against the oracle (emulated backend on CPU, real library with -m gpu)."""
import argparse
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, frames, load_golden


def test_header_symbols_exported_by_hip_library():
    """libzerotig_hip.so loads here (no GPU needed to dlopen) and exports every function include/zerotig_hip.h declares."""
    lib_mod = importlib.import_module("zero-tig_amd.lib")
    so = os.path.join(ROOT, "zero-tig_amd", "libzerotig_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    lib = lib_mod.Lib(so)
    assert len(lib.protos) >= 45 and set(lib.fns) == set(lib.protos)
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(lib.protos) <= exported
    assert {s for s in exported if s.startswith("zt_")} == set(lib.protos), "exported zt_* symbols must all be declared in the header"


def test_integration_doc_names_exist_in_header():
    """INTEGRATION.md's entry-point table must not drift from include/zerotig_hip.h."""
    import re
    from importlib import import_module
    lib_mod = import_module("zero-tig_amd.lib")
    declared = set(lib_mod.parse_header())
    doc = open(os.path.join(lib_mod.ROOT, "INTEGRATION.md")).read()
    named = set(re.findall(r"`(zt_\w+)`", doc))
    assert named and named <= declared, sorted(named - declared)


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib_mod = importlib.import_module("zero-tig_amd.lib")
    with pytest.raises(RuntimeError):
        lib_mod.get_lib()
    net_mod = importlib.import_module("zero-tig_amd.network")
    net = net_mod.Network(argparse.Namespace(dataset="RLV", of_scale=3))
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 16, 16))
    # and the product never imports the oracle
    src = ""
    for f in os.listdir(os.path.join(ROOT, "zero-tig_amd")):
        if f.endswith(".py") and f != "smoke.py":
            src += open(os.path.join(ROOT, "zero-tig_amd", f)).read()
    assert "zt_oracle" not in src and "from oracle" not in src


def test_state_dict_surface(synth):
    """Same 223 keys / shapes / aliases / trainable set as the reference Network."""
    sys.path.insert(0, ROOT)
    model_mod = importlib.import_module("model.model")
    net = model_mod.Network(argparse.Namespace(dataset="RLV", of_scale=3))
    sd = net.state_dict()
    inv = dict(synth.inventory())
    assert len(sd) == 223 and set(sd) == set(inv)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(inv[k]), k
    trainable = [n for n, p in net.named_parameters() if p.requires_grad]
    assert len(trainable) == 20 and sum(p.numel() for p in net.parameters() if p.requires_grad) == 92620
    assert sd["enhance.blocks.1.0.weight"].data_ptr() == sd["enhance.conv.0.weight"].data_ptr()
    net.enhance.in_conv.apply(net.enhance_weights_init)          # train.py:82-84
    assert float(net.enhance.in_conv[0].bias.abs().max()) == 0.0
    assert hasattr(net, "is_new_seq") and hasattr(net, "update_H3") and hasattr(net, "update_cache") and net.of_scale == 3
    wb = model_mod.Network(argparse.Namespace(dataset="underwater", of_scale=3))
    assert wb.is_WB and not net.is_WB
    for name in ("LossFunction", "TextureDifference", "blur", "pair_downsampler", "warp_tensor", "InputPadder", "RAFT", "Finetunemodel"):
        assert hasattr(model_mod, name), name


def test_sequential_judgment(tmp_path):
    utils = importlib.import_module("utils.utils")
    a = tmp_path / "s1"
    b = tmp_path / "s2"
    a.mkdir(), b.mkdir()
    for d, n in ((a, "0001"), (a, "0002"), (a, "0004"), (b, "0005")):
        (d / (n + ".png")).write_bytes(b"x")
    assert not utils.sequential_judgment(str(a / "0002.png"), str(a / "0001.png"))
    assert utils.sequential_judgment(str(a / "0004.png"), str(a / "0002.png"))        # gap
    assert utils.sequential_judgment(str(b / "0005.png"), str(a / "0004.png"))        # directory change
    assert utils.sequential_judgment(str(a / "0001.png"), str(a / "0001.png"))        # first frame (loader hands itself)
    with pytest.raises(AssertionError):
        utils.sequential_judgment(str(a / "0009.png"), str(a / "0001.png"))


def test_dataloader_api(tmp_path):
    from PIL import Image
    root = tmp_path / "data"
    for sub in ("low_light_10", "low_light_20"):
        d = root / "input" / "S01" / sub
        d.mkdir(parents=True)
        for i in (2, 1, 10):
            Image.fromarray(np.full((8, 12, 3), i, np.uint8)).save(str(d / ("%05d.png" % i)))
    (root / "train_list.txt").write_text("S01\n")
    cd = importlib.import_module("dataloader.create_data")
    ds = cd.CreateDataset(argparse.Namespace(dataset="RLV", lowlight_images_path=str(root)), "train")
    assert len(ds) == 6 and ds.name() == "BVI-RLV"
    x, name, path, last = ds[0]
    assert tuple(x.shape) == (3, 1080, 1920) and name == "00001" and last == path and abs(float(x.max()) - 1 / 255) < 1e-6
    x, name, path, last = ds[1]
    assert name == "00002" and last.endswith("00001.png")


def test_dropin_loss_and_utils(backend, oracle, synth):
    """loss.LossFunction / TextureDifference / SmoothLoss / L_TV and the CorrBlock seam through the C ABI vs the oracle."""
    ops, dev, _ = backend
    sys.path.insert(0, ROOT)
    loss_mod = importlib.import_module("loss")
    g = load_golden("g12_newseq_rlv_48x64")
    H, W, seed, _ = [int(v) for v in g["meta"]]
    x = frames(synth, 1, H, W)[0]
    outs = [torch.from_numpy(g["out%02d" % i]).to(dev) for i in range(23)]
    lf = loss_mod.LossFunction(False, ops=ops)
    val = lf(x.to(dev), *outs[:21])
    ref, _ = oracle.loss_terms(x, [o.cpu() for o in outs], False)
    assert abs(float(val) - float(ref)) <= 1e-4 * abs(float(ref))
    assert abs(float(val) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    s = loss_mod.SmoothLoss(ops=ops)(outs[2], outs[3])
    assert abs(float(s) - float(oracle.smooth_loss(outs[2].cpu(), outs[3].cpu()))) <= 1e-4 * float(s)
    tv = loss_mod.L_TV(ops=ops)(outs[3])
    assert abs(float(tv) - float(oracle.tv_loss(outs[3].cpu()))) <= 1e-4 * float(tv)
    m = loss_mod.TextureDifference(ops=ops)(outs[21], outs[22])
    assert (m.cpu().numpy() != g["out18"]).mean() <= 1e-3
    corr_mod = importlib.import_module("model.RAFT.corr")
    f1 = torch.from_numpy(synth.normal("ops.f1", (1, 256, 16, 24), 0.0, 1.0, 7))
    f2 = torch.from_numpy(synth.normal("ops.f2", (1, 256, 16, 24), 0.0, 1.0, 7))
    cb = corr_mod.CorrBlock(f1.to(dev), f2.to(dev), radius=4, ops=ops)
    g6 = load_golden("g6_ops")
    look = cb(torch.from_numpy(g6["lookup_coords"]).to(dev))
    assert float((look.cpu() - torch.from_numpy(g6["lookup_out"])).abs().max()) < 2e-5


def test_bench_gpus_flag_launches_one_rank_per_gpu():
    """`python bench.py --gpus N` outside a launcher starts N ranks itself (the flag used to be ignored); under a launcher
    (WORLD_SIZE set) it must not fork again.  Probe mode: ranks report their environment instead of touching the GPU."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["ZT_BENCH_LAUNCH_PROBE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["RANK"] == "0" and lines[0]["WORLD_SIZE"] == "4" and lines[0]["MASTER_ADDR"] == "127.0.0.1"
    env.update(RANK="1", LOCAL_RANK="1", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip())["RANK"] == "1"


def test_dataloader_did_sdsd_default_layouts(tmp_path):
    """Reference multi_read_data.py:144-318 layouts: DID list file + input/<folder>/*.{jpg,png}; SDSD pair directories listed per
    subset, one low-light frame per pair; any other dataset name walks the tree (the upstream DefaultDataset is truncated)."""
    from PIL import Image
    cd = importlib.import_module("dataloader.create_data")

    def img(path, v):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        Image.fromarray(np.full((6, 8, 3), v, np.uint8)).save(path)
    did = str(tmp_path / "did")
    for i in (3, 1, 2):
        img(os.path.join(did, "input", "V1", "%d.jpg" % i), i)
    img(os.path.join(did, "input", "V2", "7.png"), 7)
    open(os.path.join(did, "train_list.txt"), "w").write("V1\nV2\n")
    ds = cd.CreateDataset(argparse.Namespace(dataset="DID", lowlight_images_path=did), "train")
    assert ds.name() == "DID" and [ds[i][1] for i in range(len(ds))] == ["1", "2", "3", "7"]
    assert tuple(ds[0][0].shape) == (3, 1080, 1920)
    sd = str(tmp_path / "sdsd")
    for pair, stem in (("pair5", 5), ("pair2", 2)):
        img(os.path.join(sd, "indoor", "indoor_png", pair, "%d.png" % stem), stem)
        img(os.path.join(sd, "indoor", "indoor_png", pair, "%d_gt.png" % stem), 200)
    img(os.path.join(sd, "outdoor", "outdoor_png", "o1", "9.png"), 9)
    open(os.path.join(sd, "sdsd_in_test.txt"), "w").write("pair5\npair2\nmissing\n")
    open(os.path.join(sd, "sdsd_out_test.txt"), "w").write("o1\n")
    ds = cd.CreateDataset(argparse.Namespace(dataset="SDSD", lowlight_images_path=sd), "test")
    assert ds.name() == "SDSD" and [ds[i][1] for i in range(len(ds))] == ["2", "5", "9"]
    uw = str(tmp_path / "uw")
    for i in (12, 11):
        img(os.path.join(uw, "clipA", "%d.png" % i), i)
    open(os.path.join(uw, ".hidden.png"), "w").write("x")
    ds = cd.CreateDataset(argparse.Namespace(dataset="underwater", lowlight_images_path=uw), "train")
    x, name, path, last = ds[1]
    assert len(ds) == 2 and name == "12" and last.endswith("11.png") and abs(float(x.max()) - 12 / 255) < 1e-6


def test_output_side_quantise_and_psnr(backend, oracle):
    """predict.py:57-61 save_images and evals.py:83-85 PSNR on the device vs their numpy definitions (bit / integer exact)."""
    ops, dev, _ = backend
    g = torch.Generator().manual_seed(5)
    H, W = 37, 53                                   # HW not a multiple of 4: ragged tail
    a = torch.rand(1, 3, H, W, generator=g)
    a[0, 0, 0, :6] = torch.tensor([0.0, 1.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 254.5 / 255])      # ties -> half-to-even
    b = torch.clamp(a + 0.02 * torch.randn(1, 3, H, W, generator=g), 0, 1)
    ref0 = np.clip(np.transpose(a[0].numpy(), (1, 2, 0)) * 255.0, 0, 255.0).astype("uint8")
    ref1 = np.round(np.transpose(a[0].numpy(), (1, 2, 0)) * 255).astype(np.uint8)
    assert np.array_equal(ops.quantize_u8(a.to(dev), 0).cpu().numpy(), ref0)
    assert np.array_equal(ops.quantize_u8(a.to(dev), 1).cpu().numpy(), ref1)
    q = lambda t: np.round(t.numpy().astype(np.float32) * np.float32(255)).astype(np.int64)
    sq = int(((q(a) - q(b)) ** 2).sum())                                       # cv2.PSNR's sum, in exact integers
    assert ops.psnr_u8(a.to(dev), b.to(dev)) == 10.0 * np.log10(255.0 ** 2 * a.numel() / sq)
    assert abs(ops.psnr_u8(a.to(dev), b.to(dev)) - oracle.psnr_u8(a, b)) < 1e-6   # the oracle averages in fp32
    assert ops.psnr_u8(a.to(dev), a.to(dev)) == float("inf")


def test_resume_checkpoint_roundtrip(backend, synth, tmp_path):
    """model (223 reference keys) + Adam moments + step survive a save / load; training continues bit-identically."""
    ops, dev, _ = backend
    optim = importlib.import_module("zero-tig_amd.optim")
    net_mod = importlib.import_module("zero-tig_amd.network")
    sys.path.insert(0, ROOT)
    utils = importlib.import_module("utils.utils")
    xs = [f.to(dev) for f in frames(synth, 3, 48, 64)]

    def make():
        net = net_mod.Network(argparse.Namespace(dataset="RLV", of_scale=3), ops=ops)
        st = synth.make_state(1)
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
        net = net.to(dev).train()
        return net, optim.ClipAdam(net)
    net, opt = make()
    ts = optim.TrainStep(net, opt, use_graph=False)
    ts(xs[0], True)
    ts(xs[1], True)
    ck = str(tmp_path / "resume.pt")
    utils.save_checkpoint(net, opt, ck, epoch=1, step=2)
    assert set(torch.load(ck)["model"].keys()) == set(net.state_dict().keys()) and len(net.state_dict()) == 223
    l_ref = float(ts(xs[2], True))
    net2, opt2 = make()
    assert utils.load_checkpoint(net2, opt2, ck) == (1, 2) and opt2.t == 2
    l_res = float(optim.TrainStep(net2, opt2, use_graph=False)(xs[2], True))
    assert l_res == l_ref and torch.equal(opt.fp.flat, opt2.fp.flat) and torch.equal(opt.m, opt2.m)
    plain = str(tmp_path / "plain.pt")
    utils.save(net, plain)
    assert utils.load_checkpoint(net2, None, plain) == (0, 0)
