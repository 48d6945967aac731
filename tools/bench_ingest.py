#!/usr/bin/env python3
"""GPU box: end-to-end training throughput FROM IMAGE FILES next to bench.py's synthetic HBM-resident number.
Writes N synthetic 1080p low-light frames as PNGs in the BVI-RLV layout under /tmp, then runs train.py (decode workers ->
uint8 over PCIe -> device ToTensor -> hipGraph step) for 2 epochs and reports the frames/s train.py logs for the second one; with
--host_ingest the reference's host-side resize + ToTensor (fp32 over PCIe) for comparison.  Usage: python tools/bench_ingest.py [N] [workers]"""
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import importlib
    from PIL import Image
    synth = importlib.import_module("zero-tig_amd.synth")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    workers = sys.argv[2] if len(sys.argv) > 2 else "-1"
    root = "/tmp/zt_ingest_bench/RLV"
    d = os.path.join(root, "input", "S01", "low_light_10")
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    for t in range(n):
        f = os.path.join(d, "%05d.png" % (t + 1))
        if not os.path.exists(f):
            a = np.asarray(synth.lowlight_frame(t, 1080, 1920), dtype=np.float32)[0]
            Image.fromarray((np.transpose(a, (1, 2, 0)) * 255.0 + 0.5).astype(np.uint8)).save(f, compress_level=1)
    for lst, body in (("train_list.txt", "S01\n"), ("test_list.txt", "")):
        open(os.path.join(root, lst), "w").write(body)
    # the per-epoch test loop needs at least one frame: a second, one-frame scene
    d2 = os.path.join(root, "input", "S02", "low_light_10")
    os.makedirs(d2, exist_ok=True)
    if not os.path.exists(os.path.join(d2, "00001.png")):
        Image.open(os.path.join(d, "00001.png")).save(os.path.join(d2, "00001.png"), compress_level=1)
    open(os.path.join(root, "test_list.txt"), "w").write("S02\n")
    print("[bench_ingest] %d PNG frames (%.1f MB each) written in %.1f s" % (n, os.path.getsize(os.path.join(d, "00001.png")) / 1e6, time.time() - t0),
          file=sys.stderr, flush=True)
    out = {}
    for mode, extra in (("device_ingest", []), ("host_ingest", ["--host_ingest"])):
        r = subprocess.run([sys.executable, "train.py", "--lowlight_images_path", root, "--epochs", "2", "--num_workers", workers,
                            "--save", "/tmp/zt_ingest_bench/exp_" + mode] + extra, cwd=ROOT, capture_output=True, text=True, timeout=1500,
                           env=dict(os.environ, PYTHONPATH=ROOT))
        if r.returncode != 0:
            print(r.stdout[-2000:], r.stderr[-3000:], file=sys.stderr)
            raise SystemExit(1)
        fps = [float(m.group(1)) for m in re.finditer(r"throughput: .* = ([0-9.]+) frames/s", r.stdout)]
        wk = re.search(r"decode \((\d+) workers\)", r.stdout)
        out[mode] = {"frames_per_s_epoch0_incl_warmup": fps[0], "frames_per_s_epoch1": fps[1], "workers": int(wk.group(1)) if wk else None}
        print("[bench_ingest] %s: %s" % (mode, out[mode]), file=sys.stderr, flush=True)
    print(json.dumps({"what": "train.py end to end from %d 1080p PNG files (decode -> PCIe -> step), frames/s of the second epoch" % n, **out}))


if __name__ == "__main__":
    main()
