#!/usr/bin/env python3
"""Headline benchmark: 1080p self-supervised Zero-TIG training step (frames/s) on N MI355X of one node.

One "step" = one steady-state frame of the reference loop (train.py:119-131): forward incl. bilinear downscale +
histogram equalisation + RAFT(12 iterations) + fused backward warp, LossFunction, hand-written backward, one
flat-bucket gradient all-reduce over RCCL (N > 1), clip_grad_norm_(5) + Adam.  Frames are synthetic
(zero-tig_amd/synth.py) and already resident in HBM when the timed region starts.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel, measured live with HIP
events on the launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores, rank 0, N == 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--of_scale", type=int, default=3)
    ap.add_argument("--dataset", type=str, default="RLV")
    ap.add_argument("--cpu-baseline", type=str, default="540p", choices=["540p", "1080p", "none"])
    ap.add_argument("--frames", type=int, default=6, help="distinct synthetic frames kept in HBM (cycled)")
    ap.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"],
                    help="bf16: activations+weights bf16 in HBM, fp32 accumulate (BASELINE config 3); fp32: exact-fp32 parity mode")
    return ap.parse_args()


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def host_cores():
    """Usable host cores: affinity mask capped by the cgroup CPU quota (the GPU box grants a share of a big host)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(kind, of_scale, dataset):
    """The oracle (a torch-CPU restatement of the reference, pinned to reference-generated goldens) timed on the host cores.
    Sample: one steady-state training step (frame 1 of a clip; frame 0 primes the recurrent cache, untimed)."""
    from oracle import zt_oracle
    synth = importlib.import_module("zero-tig_amd.synth")
    H, W = (540, 960) if kind == "540p" else (1080, 1920)
    cores = host_cores()
    torch.set_num_threads(cores)
    log("cpu_baseline: oracle on %d host threads at %dx%d (cache-priming step)" % (cores, H, W))
    tr = zt_oracle.OracleTrainer(zt_oracle.to_torch_state(synth.make_state(1)), is_WB=(dataset == "underwater"), of_scale=of_scale)
    x0 = torch.from_numpy(synth.lowlight_frame(0, H, W))
    x1 = torch.from_numpy(synth.lowlight_frame(1, H, W))
    tr.step(x0, True)
    log("cpu_baseline: timed steady-state step")
    t0 = time.perf_counter()
    tr.step(x1, False)
    dt = time.perf_counter() - t0
    scale = (H * W) / (1080.0 * 1920.0)
    return {"value": (1.0 / dt) * scale, "unit": "frames/s (1080p-equivalent)", "cores": cores, "kind": "port",
            "sample": "1 steady-state training step (fwd+RAFT+warp+loss+bwd+clip+Adam, fp32) of the CPU oracle at %dx%d in %.2f s, "
                      "after 1 untimed cache-priming step; scaled by pixel count to 1080p" % (H, W, dt),
            "seconds_per_step_sample": dt}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    synth = importlib.import_module("zero-tig_amd.synth")
    net_mod = importlib.import_module("zero-tig_amd.network")
    optim = importlib.import_module("zero-tig_amd.optim")
    args = argparse.Namespace(dataset=a.dataset, of_scale=a.of_scale)
    net = net_mod.Network(args, precision=a.precision)
    st = synth.make_state(1)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in st.items()})
    net = net.to(dev)
    net.train()
    opt = optim.ClipAdam(net, lr=1e-4, betas=(0.9, 0.999), weight_decay=3e-4, max_norm=5.0)

    H, W = a.height, a.width
    nfr = max(2, min(a.frames, a.steps + a.warmup + 1))
    # each rank owns its own clip (seed 2 + 1000*rank): frame-level data parallelism with per-rank recurrent cache
    frames = []
    for t in range(nfr):
        frames.append(torch.from_numpy(synth.lowlight_frame(t, H, W, seed=2 + 1000 * rank)).to(dev))
        if rank == 0:
            log("synthetic frame %d/%d resident in HBM" % (t + 1, nfr))

    def step(i):
        net.is_new_seq = (i == 0)
        opt.zero_grad()
        loss = net._loss(frames[i % nfr])
        loss.backward()
        opt.step()
        return loss

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    if rank == 0:
        log("warm-up")
    for _ in range(max(1, a.warmup)):           # at least one step: frame 0 primes the cache (new sequence, no RAFT)
        step(it)
        it += 1
    # live roofline instrumentation of the dominant kernel (Enhancer 64->64 3x3 conv: fwd + dgrad launches)
    prof = {"match": (3, 3, 1, 64, 64, H, W), "events": []}
    net._ops.profile = prof
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step(it)
        it += 1
    sync()
    dt = time.perf_counter() - t0
    net._ops.profile = None
    last_loss = float(loss.detach())
    if rank == 0:
        log("timed region done: %.2f ms/step" % (1e3 * dt / a.steps))
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    value = world * a.steps / dt

    roof = None
    if prof["events"]:
        ms = [s.elapsed_time(e) for s, e in prof["events"]]
        avg_ms = sum(ms) / len(ms)
        flops = 2.0 * 9 * 64 * 64 * H * W
        tf = flops / (avg_ms * 1e-3) / 1e12
        if a.precision == "bf16":
            # bf16: 288 FLOP/B sits at the ridge (2500 TF / 8 TB/s = 312); priced against HBM.  Per launch the kernel reads the
            # 64-ch bf16 input once and writes the 64-ch output once (the three dgrad launches also read the residual df).
            px = float(H) * W
            alg = (3 * (2 * px * 64 * 2) + 3 * (3 * px * 64 * 2)) / 6.0
            traffic = None
            try:        # measured HBM traffic of the same kernel from the committed PMC passes (profiles/, see its _how)
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_l_pmc_summary.json")))
                if (H, W) == (1080, 1920):
                    traffic = pm["kernels"]["conv_rs_bf16_kernel<2, 2, true, 2, 0>"]["hbm_bytes_per_launch_avg"]
            except Exception:
                pass
            gbs = alg / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "conv_rs_bf16_kernel<2,2,true,2,0,*> (Enhancer 64->64 3x3: 3 fwd + 3 dgrad launches per step)",
                    "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": traffic, "launches": len(ms),
                    "avg_ms": avg_ms, "algorithmic_bytes_per_launch": alg,
                    "mfma_view": {"achieved_TFLOPs": tf, "peak_TFLOPs": PEAK_BF16_MFMA_TFLOPS, "frac": tf / PEAK_BF16_MFMA_TFLOPS,
                                  "algorithmic_flops_per_launch": flops}}
        else:
            roof = {"bound": "mfma", "kernel": "conv_mfma_f32_kernel<3,3,1,4> (Enhancer 64->64 3x3, fwd+dgrad)", "achieved": tf,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                    "launches": len(ms), "avg_ms": avg_ms, "algorithmic_flops_per_launch": flops}

    cpu = None
    if rank == 0 and world == 1 and a.cpu_baseline != "none":
        cpu = cpu_baseline(a.cpu_baseline, a.of_scale, a.dataset)

    if rank == 0:
        out = {"metric": "1080p self-supervised training frames/sec", "value": value, "unit": "frames/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": ("bf16" if a.precision == "bf16" else "f32"), "data": "synthetic",
               "config": {"workload": "%dx%d BVI-RLV-style self-supervised training step (enhance+RAFT flow+warp+loss+backward+clip+Adam), "
                                      "batch 1 frame per GPU, of_scale=%d, dataset=%s" % (H, W, a.of_scale, a.dataset),
                          "parallelism": "dp%d (one contiguous clip per rank, one 370 KB flat-bucket all-reduce per step)" % world,
                          "global_batch": world},
               "roofline": roof, "cpu_baseline": cpu, "final_loss": last_loss}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
