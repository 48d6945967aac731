set -x
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_kernels.py -x -q -m gpu -k "wgrad or raft or corr_volume" > gpurun_out/r03f_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03f_tests.log; tail -2 gpurun_out/r03f_tests.log
python tools/bench_wgrad64.py 2>&1 | grep wgrad | tee gpurun_out/r03f_wgrad64_ab.txt
for v in 1 0 1 0; do ZT_FUSED_CORR=$v python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03f_bench_corr$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03f_bench_corr$v.json').read().strip().splitlines()[-1]); print('FUSED_CORR=$v', d['ms_per_step'], d['ms_per_step_median'], d['roofline_extra']['corr_volume'])"; done 2>&1 | tee gpurun_out/r03f_bench_ab.txt
for v in 1 0; do ZT_FUSED_CORR=$v python bench.py --height 2160 --width 3840 --dataset underwater --steps 8 --warmup 2 --cpu-baseline none 2>/dev/null > gpurun_out/r03f_bench4k_corr$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03f_bench4k_corr$v.json').read().strip().splitlines()[-1]); print('4K FUSED_CORR=$v', d['ms_per_step'], d['ms_per_step_median'], d['roofline_extra'])"; done 2>&1 | tee -a gpurun_out/r03f_bench_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03f_prof1080 -o b -- python3 $R/bench.py --steps 16 --warmup 3 --cpu-baseline none > $R/gpurun_out/r03f_prof1080_line.json 2>/dev/null
cd $R
rm -f gpurun_out/r03f_prof1080/*kernel_trace.csv
grep -i "corr_pyramid\|wgrad64" gpurun_out/r03f_prof1080/b_kernel_stats.csv | cut -c1-200
