set -x
python -m pytest tests/test_kernels.py tests/test_engine.py -x -q -m gpu -k "wgrad or bf16 or golden" > gpurun_out/r03o_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03o_tests.log; tail -2 gpurun_out/r03o_tests.log
for v in 1 0; do ZT_WGRAD_DMA=$v python tools/bench_wgrad.py 2>&1 | grep wgrad | sed "s/^/DMA=$v /"; done | tee gpurun_out/r03o_wgrad_all.txt
for v in 1 0 1 0; do ZT_WGRAD_DMA=$v python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03o_bench_dma$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03o_bench_dma$v.json').read().strip().splitlines()[-1]); print('WGRAD_DMA=$v', d['ms_per_step'], d['ms_per_step_median'])"; done 2>&1 | tee gpurun_out/r03o_bench_ab.txt
