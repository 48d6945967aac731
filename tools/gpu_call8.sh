set -x
R=$GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > gpurun_out/r03h_tests_full.log 2>&1; echo "rc=$?" >> gpurun_out/r03h_tests_full.log; tail -3 gpurun_out/r03h_tests_full.log
python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03h_bench.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03h_bench.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'], 'h2d', d['with_h2d']['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03h_prof1080 -o b -- python3 $R/bench.py --steps 16 --warmup 3 --cpu-baseline none > $R/gpurun_out/r03h_prof1080_line.json 2>/dev/null
cd $R
rm -f gpurun_out/r03h_prof1080/*kernel_trace.csv
grep -i "conv_rs_bf16_kernel<2, 2, true, 2, 0\|loss_s2" gpurun_out/r03h_prof1080/b_kernel_stats.csv | cut -c1-220
