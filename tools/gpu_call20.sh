set -x
timeout -k 10 600 python -m pytest tests/test_kernels.py tests/test_engine.py -x -q -m gpu -k "conv_bf16 or bn_stats or dgrad_bn_sums or rs_pipeline or bf16 or golden" > gpurun_out/r03t_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03t_tests.log; tail -3 gpurun_out/r03t_tests.log
grep -q "rc=0" gpurun_out/r03t_tests.log || exit 1
timeout -k 10 300 python tools/bench_conv.py 2>/dev/null | grep "epi" > gpurun_out/r03t_conv.txt 2>&1; cat gpurun_out/r03t_conv.txt
for v in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03t_bench_$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03t_bench_$v.json').read().strip().splitlines()[-1]); print('run $v', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_ms'], d['roofline']['frac'])"; done 2>&1 | tee gpurun_out/r03t_bench.txt
