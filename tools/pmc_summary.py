#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc counter_collection CSVs to per-kernel averages and merge the FETCH_SIZE / WRITE_SIZE passes.

FETCH_SIZE / WRITE_SIZE are KB per dispatch.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE of wide 16-B/lane
streaming reads counts half the bytes, so it is doubled for the kernels that read 16 B per lane (the bf16 conv / wgrad / norm
kernels); 4-B/lane kernels are left as counted (uncalibrated)."""
import csv
import json
import re
import sys

WIDE = ("conv_rs_bf16", "conv_ws_bf16", "conv_mfma_bf16", "wgrad_mfma_bf16", "wgrad64_dma_bf16", "corr_pyramid_bf16", "thin1x1_bwd_bf16", "conv1x1_thin",
        "stats_bf16x8", "bn_bwd", "norm_apply", "chan_stats", "relu_mask")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name).strip()


def reduce(path, ctr):
    acc = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != ctr:
                continue
            k = short(row["Kernel_Name"])
            v = float(row["Counter_Value"])
            a = acc.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += v
            a[2] = max(a[2], v)
    return {k: {"dispatches": a[0], ctr + "_KB_avg": a[1] / a[0], ctr + "_KB_max": a[2]} for k, a in acc.items()}


def main():
    if sys.argv[1] == "--reduce":
        print(json.dumps(reduce(sys.argv[2], sys.argv[3])))
        return
    fe, wr = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    out = {"_how": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- python3 bench.py --steps 2 "
                   "--warmup 2 --cpu-baseline none (tools/run_pmc.sh); KB per dispatch averaged over every launch of the kernel; "
                   "gfx950 correction: FETCH_SIZE x2 for the 16-B/lane kernels, as counted for the others",
           "kernels": {}}
    for k, v in fe.items():
        w = wr.get(k, {})
        corr = 2.0 if any(t in k for t in WIDE) else 1.0
        e = dict(v)
        e.update({kk: vv for kk, vv in w.items() if kk != "dispatches"})
        e["fetch_correction"] = corr
        e["hbm_bytes_per_launch_avg"] = (v["FETCH_SIZE_KB_avg"] * corr + w.get("WRITE_SIZE_KB_avg", 0.0)) * 1024.0
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
