#!/bin/bash
# GPU box: the round's judged artifacts (arg 1 = tag): default bench line (with the CPU baseline), rocprofv3 kernel stats of the
# bench, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only), and the other configurations' bench lines.
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
step() {
  local lim=$1 log=$2; shift 2
  timeout -k 10 $lim "$@" > $log 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc  $*"
  if [ $rc -ge 124 ]; then echo "killed at its limit: stopping"; tail -5 $log; exit $rc; fi
  return $rc
}
step 500 gpurun_out/${tag}_bench_default.err python bench.py
tail -1 gpurun_out/${tag}_bench_default.err > gpurun_out/${tag}_bench1080p_bf16_unprofiled_bench_line.json; cut -c1-300 gpurun_out/${tag}_bench1080p_bf16_unprofiled_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_prof
step 500 $R/gpurun_out/${tag}_prof.err rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o out -- python3 $R/bench.py --steps 5 --warmup 3 --cpu-baseline none
grep "\"metric\"" $R/gpurun_out/${tag}_prof.err | tail -1 > $R/gpurun_out/${tag}_bench1080p_bf16_bench_line.json
f=$(find $R/gpurun_out/${tag}_prof -name "*kernel_stats.csv" | sort | tail -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_bench1080p_bf16_kernel_stats.csv && rm -rf $R/gpurun_out/${tag}_prof
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/${tag}_$ctr
  step 500 $R/gpurun_out/${tag}_$ctr.err rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$ctr -o out -- python3 $R/bench.py --graph 0 --steps 2 --warmup 2 --cpu-baseline none
  f=$(find $R/gpurun_out/${tag}_$ctr -name "*counter_collection.csv" | sort | tail -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py --reduce $f $ctr > $R/gpurun_out/${tag}_$ctr.reduced.json
  rm -rf $R/gpurun_out/${tag}_$ctr
done
python3 $R/tools/pmc_summary.py --merge $R/gpurun_out/${tag}_FETCH_SIZE.reduced.json $R/gpurun_out/${tag}_WRITE_SIZE.reduced.json > $R/gpurun_out/${tag}_pmc_summary.json
cd $R
step 300 gpurun_out/${tag}_fp32.err python bench.py --precision fp32 --cpu-baseline none --steps 10
tail -1 gpurun_out/${tag}_fp32.err > gpurun_out/${tag}_bench1080p_fp32_unprofiled_bench_line.json
step 300 gpurun_out/${tag}_540.err python bench.py --height 540 --width 960 --cpu-baseline none
tail -1 gpurun_out/${tag}_540.err > gpurun_out/${tag}_bench540p_bf16_bench_line.json
step 400 gpurun_out/${tag}_4k.err python bench.py --height 2160 --width 3840 --dataset underwater --cpu-baseline none --steps 10 --frames 3
tail -1 gpurun_out/${tag}_4k.err > gpurun_out/${tag}_bench4k_underwater_bf16_bench_line.json
step 300 gpurun_out/${tag}_eager.err python bench.py --graph 0 --cpu-baseline none
tail -1 gpurun_out/${tag}_eager.err > gpurun_out/${tag}_bench1080p_bf16_eager_bench_line.json
for f in gpurun_out/${tag}_bench*bench_line.json; do echo "$f: $(cut -c1-200 $f)"; done
exit 0
