#!/bin/bash
# GPU box: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes of the 4K underwater configuration (BASELINE config 5, one rank)
tag=${1:-r03_4k}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
ARGS="--height 2160 --width 3840 --dataset underwater --cpu-baseline none --frames 3"
step() {
  local lim=$1 log=$2; shift 2
  timeout -k 10 $lim "$@" > $log 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc  $*"
  if [ $rc -ge 124 ]; then echo "killed at its limit: stopping"; tail -5 $log; exit $rc; fi
  return $rc
}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_prof
step 500 $R/gpurun_out/${tag}_prof.err rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o out -- python3 $R/bench.py $ARGS --steps 5 --warmup 2 || exit 1
grep "\"metric\"" $R/gpurun_out/${tag}_prof.err | tail -1 > $R/gpurun_out/${tag}_bench_line.json
f=$(find $R/gpurun_out/${tag}_prof -name "*kernel_stats.csv" | sort | tail -1)
[ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats.csv && rm -rf $R/gpurun_out/${tag}_prof
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/${tag}_$ctr
  step 500 $R/gpurun_out/${tag}_$ctr.err rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$ctr -o out -- python3 $R/bench.py $ARGS --graph 0 --steps 2 --warmup 2 || exit 1
  f=$(find $R/gpurun_out/${tag}_$ctr -name "*counter_collection.csv" | sort | tail -1)
  [ -n "$f" ] && python3 $R/tools/pmc_summary.py --reduce $f $ctr > $R/gpurun_out/${tag}_$ctr.reduced.json
  rm -rf $R/gpurun_out/${tag}_$ctr
done
python3 $R/tools/pmc_summary.py --merge $R/gpurun_out/${tag}_FETCH_SIZE.reduced.json $R/gpurun_out/${tag}_WRITE_SIZE.reduced.json > $R/gpurun_out/${tag}_pmc_summary.json
head -c 400 $R/gpurun_out/${tag}_bench_line.json; echo
exit 0
