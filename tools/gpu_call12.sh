set -x
python -m pytest tests/test_kernels.py -x -q -m gpu -k "rs_pipeline or conv_bf16 or bn_stats" > gpurun_out/r03l_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03l_tests.log; tail -2 gpurun_out/r03l_tests.log
ZT_BENCH_ABL="64,64" python tools/bench_conv.py 2>&1 | grep "c64" | tee gpurun_out/r03l_conv_rs_ablation.txt
ZT_BENCH_ABL="48,48" python tools/bench_conv.py 2>&1 | grep "c48" | tee -a gpurun_out/r03l_conv_rs_ablation.txt
for v in 1 2; do python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03l_bench$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03l_bench$v.json').read().strip().splitlines()[-1]); print('run $v', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'])"; done 2>&1 | tee gpurun_out/r03l_bench.txt
