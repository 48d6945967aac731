set -x
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_kernels.py -x -q -m gpu -k "wgrad or raft" > gpurun_out/r03e_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03e_tests.log; tail -2 gpurun_out/r03e_tests.log
python tools/bench_wgrad64.py 2>&1 | grep wgrad | tee gpurun_out/r03e_wgrad64_ab.txt
bash tools/pmc_kernel.sh "wgrad64_dma" tools/bench_wgrad64.py 2>&1 | grep -v "simple_timer" > gpurun_out/r03e_wgrad64_pmc.txt; cut -c1-330 gpurun_out/r03e_wgrad64_pmc.txt
for v in 1 0; do ZT_RAFT_PAIR=$v python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03e_bench_pair$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03e_bench_pair$v.json').read().strip().splitlines()[-1]); print('PAIR=$v', d['ms_per_step'], d['ms_per_step_median'])"; done 2>&1 | tee gpurun_out/r03e_bench_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03e_prof1080 -o b -- python3 $R/bench.py --steps 16 --warmup 3 --cpu-baseline none > $R/gpurun_out/r03e_prof1080_line.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03e_prof4k -o b -- python3 $R/bench.py --height 2160 --width 3840 --dataset underwater --steps 6 --warmup 2 --cpu-baseline none > $R/gpurun_out/r03e_prof4k_line.json 2>/dev/null
cd $R
rm -f gpurun_out/r03e_prof1080/*kernel_trace.csv gpurun_out/r03e_prof4k/*kernel_trace.csv
tail -1 gpurun_out/r03e_prof4k_line.json | cut -c1-300
