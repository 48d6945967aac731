#!/bin/bash
# GPU box: wgrad tests, then the full bench for a few slab counts (ZT_WGRAD_BLOCKS tuning hook)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "wgrad or norm or bn" > gpurun_out/wg_tests.log 2>&1
tail -2 gpurun_out/wg_tests.log
for nb in 512 256; do
  ZT_WGRAD_BLOCKS=$nb timeout -k 10 400 python bench.py --steps 20 --warmup 3 --cpu-baseline none > gpurun_out/wg_bench_$nb.log 2>gpurun_out/wg_bench_$nb.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/wg_bench_$nb.log').read().strip().splitlines()[-1]); print('blocks $nb', d['ms_per_step'], d['value'])"
done
