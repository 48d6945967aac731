#!/bin/bash
# GPU box: -m gpu suite, bench (hipGraph), then rocprofv3 kernel stats of the same bench command (arg 1 = tag).
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
step() {   # step <seconds> <logfile> <cmd...>
  local lim=$1 log=$2; shift 2
  timeout -k 10 $lim "$@" > $log 2>&1
  local rc=$?
  echo "[$(date +%T)] rc=$rc  $*  (log $log)"
  if [ $rc -ge 124 ]; then echo "killed at its limit: stopping"; tail -5 $log; exit $rc; fi
  return $rc
}
if [ -z "$SKIP_TESTS" ]; then
  step 900 gpurun_out/${tag}_pytest.log python -m pytest tests -q -m gpu -x --durations=5 ${PYTEST_K:+-k "$PYTEST_K"}; tail -12 gpurun_out/${tag}_pytest.log
fi
step 400 gpurun_out/${tag}_bench.err python bench.py --cpu-baseline none --steps 20
tail -1 gpurun_out/${tag}_bench.err > gpurun_out/${tag}_bench_line.json; cut -c1-400 gpurun_out/${tag}_bench_line.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${tag}_prof
step 500 $R/gpurun_out/${tag}_prof.err rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_prof -o out -- python3 $R/bench.py --steps 3 --warmup 3 --cpu-baseline none
cd $R
f=$(find gpurun_out/${tag}_prof -name "*kernel_stats.csv" | sort | tail -1)
[ -n "$f" ] && cp $f gpurun_out/${tag}_kernel_stats.csv && rm -rf gpurun_out/${tag}_prof && head -45 gpurun_out/${tag}_kernel_stats.csv | cut -c1-160
exit 0
