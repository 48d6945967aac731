set -x
python -m pytest tests/test_kernels.py tests/test_engine.py -x -q -s -m gpu -k "raft or stem or bn_fold or bf16_sequence or bf16_mode or conv_bf16" > gpurun_out/r03q_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r03q_tests.log; grep -E "rel-L2|passed|failed|rc=" gpurun_out/r03q_tests.log
for v in 1 0; do ZT_RAFT_FOLD_BN=$v python tools/bench_raft.py 2>/dev/null | tail -1; done | tee gpurun_out/r03q_bench_raft.txt
for v in 1 0 1 0; do ZT_RAFT_FOLD_BN=$v python bench.py --steps 20 --warmup 3 --cpu-baseline none 2>/dev/null > gpurun_out/r03q_bench_fold$v.json; python -c "import json,sys; d=json.loads(open('gpurun_out/r03q_bench_fold$v.json').read().strip().splitlines()[-1]); print('RAFT_FOLD_BN=$v', d['ms_per_step'], d['ms_per_step_median'])"; done 2>&1 | tee gpurun_out/r03q_bench_ab.txt
