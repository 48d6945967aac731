#!/bin/bash
# GPU box: the -m gpu suite, smoke(), and the bench in eager and hipGraph mode (arg 1 = output tag).  A step that is killed at
# its time limit ends the call (no further GPU step is started after a hang).
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
step 900 gpurun_out/${tag}_pytest.log python -m pytest tests -q -m gpu -x --durations=8 ${PYTEST_K:+-k "$PYTEST_K"}; tail -25 gpurun_out/${tag}_pytest.log
step 300 gpurun_out/${tag}_smoke.log python -c "import __graft_entry__ as g; g.smoke()"; tail -4 gpurun_out/${tag}_smoke.log
step 400 gpurun_out/${tag}_bench_eager.err python bench.py --graph 0 --cpu-baseline none --steps 20; tail -3 gpurun_out/${tag}_bench_eager.err | cut -c1-1500
step 400 gpurun_out/${tag}_bench_graph.err python bench.py --graph 1 --cpu-baseline none --steps 20; tail -3 gpurun_out/${tag}_bench_graph.err | cut -c1-1500
exit 0
