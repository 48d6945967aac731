#!/bin/bash
# GPU box: pc-kernel tests, per-layer timings, full bench with and without the producer/consumer kernel
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "conv_bf16" > gpurun_out/pc_tests.log 2>&1
tail -2 gpurun_out/pc_tests.log
timeout -k 10 300 python tools/bench_conv.py > gpurun_out/pc_conv.log 2>&1
tail -30 gpurun_out/pc_conv.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --cpu-baseline none > gpurun_out/pc_bench.log 2>gpurun_out/pc_bench.err
cat gpurun_out/pc_bench.log
ZT_CONV_RS=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --cpu-baseline none > gpurun_out/pc_bench_off.log 2>gpurun_out/pc_bench_off.err
cat gpurun_out/pc_bench_off.log
