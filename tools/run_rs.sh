#!/bin/bash
# GPU box: conv tests, per-layer timings of the bf16 conv variants, then the full bench with the register-stationary kernel in its
# default form (two 4-wave workgroups per CU), in the one-workgroup 8-row form (ZT_CONV_RS4=0) and switched off (ZT_CONV_RS=0)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels.py -x -q -m gpu -k "conv_bf16" > gpurun_out/rs_tests.log 2>&1
tail -2 gpurun_out/rs_tests.log
timeout -k 10 300 python tools/bench_conv.py > gpurun_out/rs_conv.log 2>&1
grep " us" gpurun_out/rs_conv.log
for cfg in "ZT_CONV_RS=1" "ZT_CONV_RS4=0" "ZT_CONV_RS=0"; do
  env $cfg timeout -k 10 400 python bench.py --steps 20 --warmup 3 --cpu-baseline none > gpurun_out/rs_bench.json 2> gpurun_out/rs_bench.err
  python -c "import json; d=json.loads(open('gpurun_out/rs_bench.json').read().strip().splitlines()[-1]); print('$cfg', round(d['ms_per_step'], 3), 'ms/step', round(d['value'], 1), 'frames/s')"
done
