set -x
python -m pytest tests/test_kernels.py -x -q -m gpu -k "conv_pair or raft or wgrad or corr_volume" > gpurun_out/r03g_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03g_tests.log; tail -2 gpurun_out/r03g_tests.log
for v in 1 0 1 0; do ZT_RAFT_PAIR=$v python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03g_bench_pair$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03g_bench_pair$v.json').read().strip().splitlines()[-1]); print('PAIR=$v', d['ms_per_step'], d['ms_per_step_median'], 'h2d', d['with_h2d']['ms_per_step'], d['with_h2d']['ms_per_step_median'])"; done 2>&1 | tee gpurun_out/r03g_bench_ab.txt
python tools/bench_raft.py 2>&1 | tail -5 | tee gpurun_out/r03g_bench_raft.txt
