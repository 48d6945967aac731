set -x
python -m pytest tests/test_kernels.py tests/test_engine.py -x -q -m gpu -k "wgrad or bf16 or golden" > gpurun_out/r03n_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03n_tests.log; tail -2 gpurun_out/r03n_tests.log
ZT_BENCH_CH=48 python tools/bench_wgrad64.py 2>&1 | grep wgrad | tee gpurun_out/r03n_wgrad48_ab.txt
ZT_BENCH_CH=48 python tools/bench_wgrad64.py 540 960 2>&1 | grep wgrad | tee -a gpurun_out/r03n_wgrad48_ab.txt
for v in 1 0 1 0; do ZT_WGRAD_DMA=$v python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03n_bench_dma$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03n_bench_dma$v.json').read().strip().splitlines()[-1]); print('WGRAD_DMA=$v', d['ms_per_step'], d['ms_per_step_median'])"; done 2>&1 | tee gpurun_out/r03n_bench_ab.txt
