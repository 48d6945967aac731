#!/usr/bin/env python3
"""Headline benchmark: 1080p self-supervised Zero-TIG training step (frames/s) on N MI355X of one node.

One "step" = one steady-state frame of the reference loop (train.py:119-131): forward incl. bilinear downscale +
histogram equalisation + RAFT(12 iterations) + fused backward warp, LossFunction, hand-written backward, one
flat-bucket gradient all-reduce over RCCL (N > 1), clip_grad_norm_(5) + Adam.  Frames are synthetic
(zero-tig_amd/synth.py) and already resident in HBM when the timed region starts (`value`); the same K steps are timed
a second time with the reference's per-step host->device copy of the frame (train.py:125) inside the step (`with_h2d`).

Launch: `python bench.py --gpus N ...` starts N ranks itself (one process per GPU, RCCL) when it is not already running
under torch.distributed.run / torchrun (WORLD_SIZE unset); under a launcher it reads RANK / LOCAL_RANK / WORLD_SIZE.

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (dominant kernel), `roofline_extra` (the warp and
the correlation volume, BASELINE.json's HBM targets) -- all timed live with HIP events on the launch stream in a separate
pass after the timed region -- and `cpu_baseline` (the CPU oracle timed on this box's host cores, rank 0, N == 1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--of_scale", type=int, default=3)
    ap.add_argument("--dataset", type=str, default="RLV")
    ap.add_argument("--cpu-baseline", type=str, default="1080p", choices=["540p", "1080p", "none"])
    ap.add_argument("--frames", type=int, default=6, help="distinct synthetic frames kept in HBM (cycled)")
    ap.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"],
                    help="bf16: activations+weights bf16 in HBM, fp32 accumulate (BASELINE config 3); fp32: exact-fp32 parity mode")
    ap.add_argument("--graph", type=int, default=1, help="1: replay the step from a captured hipGraph (default); 0: eager launches")
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


def launch_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one process per GPU) BEFORE this process touches the GPU
    and pass rank 0's JSON line through.  Children see WORLD_SIZE and therefore never recurse."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        # HSA_ENABLE_IPC_MODE_LEGACY=0: this pool's host driver only supports dmabuf IPC; with the legacy mode RCCL's intra-node
        # transport setup fails in hipIpcGetMemHandle ("invalid argument").  The image exports it already; the children get it
        # explicitly so that a caller's scrubbed environment cannot lose it.
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = None if r == 0 else subprocess.DEVNULL          # rank 0 prints the line on our stdout; stderr is shared
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                r = p.poll()
                if r is None:
                    continue
                procs.remove(p)
                if r != 0:                                   # one rank died: the others would wait in a collective forever
                    rc = r
                    for q in procs:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            p.kill()
    return rc


def cpu_baseline(kind, of_scale, dataset):
    """the oracle (a torch-CPU restatement of the reference, pinned to reference-generated goldens) timed on the host
    cores on the same synthetic clip: 1 cache-priming new-sequence step, 1 warm-up + 3 timed steady-state steps, median."""
    import torch
    from oracle import zt_oracle
    synth = importlib.import_module("zero-tig_amd.synth")
    H, W = (540, 960) if kind == "540p" else (1080, 1920)
    cores = host_cores()
    torch.set_num_threads(cores)
    tr = zt_oracle.OracleTrainer(zt_oracle.to_torch_state(synth.make_state(1)), is_WB=(dataset == "underwater"), of_scale=of_scale)
    times = []
    for t in range(5):
        x = torch.from_numpy(synth.lowlight_frame(t, H, W))
        log("cpu_baseline: oracle step %d/5 on %d host threads at %dx%d (%s)" %
            (t + 1, cores, H, W, "new sequence, primes the cache" if t == 0 else ("warm-up" if t == 1 else "timed")))
        t0 = time.perf_counter()
        tr.step(x, t == 0)
        dt = time.perf_counter() - t0
        if t >= 2:
            times.append(dt)
    med = sorted(times)[1]
    scale = (H * W) / (1080.0 * 1920.0)
    return {"value": (1.0 / med) * scale, "unit": "frames/s" if scale == 1.0 else "frames/s (1080p-equivalent)", "cores": cores,
            "kind": "port",
            "sample": "median of 3 steady-state training steps (fwd+RAFT+warp+loss+bwd+clip+Adam, fp32) of the CPU oracle at %dx%d "
                      "(%.2f / %.2f / %.2f s) after 1 cache-priming + 1 warm-up step%s"
                      % (H, W, times[0], times[1], times[2], "" if scale == 1.0 else "; scaled by pixel count to 1080p"),
            "seconds_per_step_median": med}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))
    if os.environ.get("ZT_BENCH_LAUNCH_PROBE"):       # tests/test_dropin.py: what a rank sees, without touching the GPU
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}), flush=True)
        return
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    use_dist = world > 1 or "RANK" in os.environ        # under a launcher (also with ONE rank) the RCCL path runs: init + all-reduce
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

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
    # host side of `with_h2d`: what the loader hands over in device-ingest mode -- the DECODED frame, uint8 [1,H,W,3] (the synthetic
    # frames are quantised to k/255 like an 8-bit PNG, so the device's ToTensor reproduces the float frame bit for bit: checked)
    net._plan()
    host_frames, frames = [], []
    for t in range(nfr):
        f32 = torch.from_numpy(synth.lowlight_frame(t, H, W, seed=2 + 1000 * rank))
        u8 = torch.round(f32[0].permute(1, 2, 0) * 255.0).to(torch.uint8).contiguous()[None].pin_memory()
        host_frames.append(u8)
        frames.append(f32.to(dev))
        if t == 0:
            assert torch.equal(net._ops.ingest_u8(u8.to(dev), size=(W, H)), frames[0]), "device ToTensor must reproduce the float frame"
        if rank == 0:
            log("synthetic frame %d/%d resident in HBM" % (t + 1, nfr))

    stepper = optim.TrainStep(net, opt, use_graph=bool(a.graph), ingest_size=(W, H))

    def step(i, from_host=False):
        src = host_frames if from_host else frames
        return stepper(src[i % nfr], is_new_seq=(i == 0))

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    it = 0
    if rank == 0:
        log("warm-up (%s)" % ("hipGraph capture + replay" if a.graph else "eager launches"))
    for _ in range(max(2, a.warmup)):     # >= 2 steps: frame 0 primes the cache (new sequence, no RAFT), frame 1 is the first steady-state step
        step(it)
        it += 1

    def timed(from_host):
        nonlocal it
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
        # from_host: the frames come from pinned host memory, the next frame's copy running on a copy stream under the current step.
        # Two untimed steps first: the prefetcher allocates its two device buffers and the ingest kernel's table on first use
        # (hipMalloc synchronises), which is start-up cost of the loop, not of a step.
        feed = optim.FramePrefetcher((host_frames[(it + k) % nfr] for k in range(a.steps + 2)), dev) if from_host else None
        if from_host:
            for _ in range(2):
                stepper(next(feed), is_new_seq=False)
                it += 1
        sync()
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(a.steps):
            loss = stepper(next(feed), is_new_seq=False) if from_host else step(it)
            ev[k + 1].record()
            it += 1
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax)
        per = sorted(ev[k].elapsed_time(ev[k + 1]) for k in range(a.steps))
        return dt, per[len(per) // 2], float(loss.detach())

    dt, med_ms, last_loss = timed(False)
    if rank == 0:
        log("timed region done: %.2f ms/step (median of the per-step HIP-event times %.2f ms)" % (1e3 * dt / a.steps, med_ms))
    dt_h, med_h, _ = timed(True)
    if rank == 0:
        log("with the per-step H2D copy of the frame: %.2f ms/step" % (1e3 * dt_h / a.steps))
    value = world * a.steps / dt

    # ---- live roofline instrumentation, separate untimed pass (eager launches, HIP events around the matched kernels)
    px = float(H) * W
    h8, w8 = ((H // a.of_scale + 7) // 8), ((W // a.of_scale + 7) // 8)
    npx = h8 * w8
    prof = {"match": {(3, 3, 1, 64, 64, H, W): "conv64", (1, 1, 1, 256, npx, h8, w8): "corr", ("wgrad", 3, 64, 64, H, W): "wgrad64"}, "events": {}}
    net._ops.profile = prof
    eager = optim.TrainStep(net, opt, use_graph=False)
    for _ in range(3):
        eager(frames[it % nfr], is_new_seq=False)
        it += 1
    torch.cuda.synchronize()
    net._ops.profile = None

    def avg_ms(name):
        """mean of the per-launch HIP-event times of one kernel family (an event pair around a short kernel also times the pair
        itself: `event_pair_overhead_ms` is reported next to the sub-50-us kernels; rocprofv3's averages are in profiles/)"""
        ev = prof["events"].get(name, [])
        return (sum(s.elapsed_time(e) for s, e in ev) / len(ev), len(ev)) if ev else (None, 0)

    # an empty HIP-event pair on the launch stream: what the bracket itself adds to a short kernel's reading
    pairs = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    ev_overhead_ms = sorted(a.elapsed_time(b) for a, b in pairs)[len(pairs) // 2]

    roof, extra = None, {}
    ms, nl = avg_ms("conv64")
    if ms is not None:
        flops = 2.0 * 9 * 64 * 64 * px
        tf = flops / (ms * 1e-3) / 1e12
        if a.precision == "bf16":
            # bf16: 288 FLOP/B sits at the ridge (2500 TF / 8 TB/s = 312); priced against HBM.  Per launch the kernel reads the
            # 64-ch bf16 input once and writes the 64-ch output once (the three dgrad launches also read the residual df).
            # With the BatchNorm-backward sums fused in (zero-tig_amd/engine.py: default at this size), two of the three dgrad
            # launches also read the previous block's pre-activation z once -- the read of the separate statistics pass they replace.
            fused_bn = os.environ.get("ZT_FUSED_BN_BWD", "1") == "1" and \
                ((W + 31) // 32) * ((H + 7) // 8) >= int(os.environ.get("ZT_STATS_FUSE_MIN_TILES", "1024"))
            alg = (3 * (2 * px * 64 * 2) + 3 * (3 * px * 64 * 2) + (2 * px * 64 * 2 if fused_bn else 0)) / 6.0
            gbs = alg / (ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": "conv_rs_bf16_kernel<2,2,true,2,0,*> (Enhancer 64->64 3x3: 3 fwd + 3 dgrad launches per step)",
                    "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                    "launches": nl, "avg_ms": ms, "algorithmic_bytes_per_launch": alg, "bn_backward_sums_fused": fused_bn,
                    "mfma_view": {"achieved_TFLOPs": tf, "peak_TFLOPs": PEAK_BF16_MFMA_TFLOPS, "frac": tf / PEAK_BF16_MFMA_TFLOPS,
                                  "algorithmic_flops_per_launch": flops}}
            try:        # NOT measured in this run: HBM traffic of the same kernels from the committed PMC passes (profiles/)
                src = "profiles/r03_pmc_summary.json"
                pm = json.load(open(os.path.join(ROOT, src)))["kernels"]
                ks = [v for k, v in pm.items() if k.startswith("conv_rs_bf16_kernel<2, 2, true, 2, 0,")]
                if (H, W) == (1080, 1920) and ks:
                    nd = sum(v["dispatches"] for v in ks)
                    roof["traffic_from_profile"] = {"bytes_per_launch": sum(v["hbm_bytes_per_launch_avg"] * v["dispatches"] for v in ks) / nd,
                                                    "source": src}
            except Exception:
                pass
        else:
            roof = {"bound": "mfma", "kernel": "conv_mfma_f32_kernel<3,3,1,4> (Enhancer 64->64 3x3, fwd+dgrad)", "achieved": tf,
                    "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                    "launches": nl, "avg_ms": ms, "algorithmic_flops_per_launch": flops}
    ms, nl = avg_ms("warp2")
    if ms is not None:          # utils.py:203-230 x2 fused: read 2 images + flow, write 2 images (fp32 planar)
        alg = 4 * px * 3 * 4 + (H // a.of_scale) * (W // a.of_scale) * 2 * 4.0
        extra["warp2_kernel"] = {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "avg_ms": ms, "launches": nl,
                                 "algorithmic_bytes_per_launch": alg, "event_pair_overhead_ms": ev_overhead_ms}
    ms, nl = avg_ms("wgrad64")
    if ms is not None:          # autograd of model.py:60-67's conv: reads the 64-ch input and the 64-ch output gradient once
        alg = 2 * px * 64 * 2
        extra["wgrad64_dma_bf16_kernel"] = {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                            "frac": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "avg_ms": ms, "launches": nl,
                                            "algorithmic_bytes_per_launch": alg,
                                            "mfma_frac": 2.0 * 9 * 64 * 64 * px / (ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS}
    ms, nl = avg_ms("corr")
    if ms is not None:          # corr.py:13-27, 52-60: the fp32 all-pairs volume and (bf16 mode: same launch) its 3 pooled levels written once, both feature maps read
        fc = os.environ.get("ZT_FUSED_CORR", "auto")
        fused = a.precision == "bf16" and (fc == "1" or (fc == "auto" and npx >= 8000))     # zero-tig_amd/raft.py: one launch for volume + pyramid
        pyr = sum((h8 >> l) * (w8 >> l) for l in (1, 2, 3)) * float(npx) * 4 if fused else 0.0
        alg = float(npx) * npx * 4 + pyr + 2 * npx * 256 * (2 if a.precision == "bf16" else 4)
        extra["corr_volume"] = {"bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": alg / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "avg_ms": ms, "launches": nl,
                                "algorithmic_bytes_per_launch": alg, "event_pair_overhead_ms": ev_overhead_ms}

    cpu = None
    if rank == 0 and world == 1 and a.cpu_baseline != "none":
        cpu = cpu_baseline(a.cpu_baseline, a.of_scale, a.dataset)

    if rank == 0:
        label = "1080p" if (H, W) == (1080, 1920) else "%dx%d" % (H, W)
        out = {"metric": "%s self-supervised training frames/sec" % label, "value": value, "unit": "frames/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "ms_per_step_median": med_ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": ("bf16" if a.precision == "bf16" else "f32"), "data": "synthetic",
               "config": {"workload": "%dx%d BVI-RLV-style self-supervised training step (enhance+RAFT flow+warp+loss+backward+clip+Adam), "
                                      "batch 1 frame per GPU, of_scale=%d, dataset=%s" % (H, W, a.of_scale, a.dataset),
                          "parallelism": "dp%d (one contiguous clip per rank, one 370 KB flat-bucket all-reduce per step)" % world,
                          "global_batch": world, "launch": "hipGraph replay" if a.graph else "eager",
                          "collective": ("flat 370 KB gradient bucket all-reduced through torch.distributed backend '%s' (RCCL) every step, world=%d"
                                         % (dist.get_backend(), world)) if use_dist else "none (single process, no launcher)"},
               "with_h2d": {"value": world * a.steps / dt_h, "ms_per_step": 1e3 * dt_h / a.steps, "ms_per_step_median": med_h,
                            "note": "same K steps with every frame coming from pinned host memory as the loaders deliver it in device-ingest mode "
                                    "(decoded uint8 HWC, 6.2 MB at 1080p; the reference moves 24.9 MB of fp32, train.py:125): the next frame's copy is "
                                    "issued on a copy stream while the current step runs (optim.FramePrefetcher, as train.py does) and ToTensor "
                                    "(+ the PIL-exact resize when the file is not 1920x1080) runs on the device (zt_ingest.hip)"},
               "roofline": roof, "roofline_extra": extra, "cpu_baseline": cpu, "final_loss": last_loss}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
