#!/bin/bash
# GPU box: HBM traffic of the bench's kernels from two separate PMC passes (never combined with tracing domains other than
# --kernel-trace).  arg 1 = output tag.  Summarise afterwards with tools/pmc_summary.py.
set -e
tag=${1:-pmc}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/${tag}_$ctr
  timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_$ctr -o out -- python3 $R/bench.py --steps 2 --warmup 2 --cpu-baseline none > $R/gpurun_out/${tag}_$ctr.json 2> $R/gpurun_out/${tag}_$ctr.err
  f=$(find $R/gpurun_out/${tag}_$ctr -name "*counter_collection.csv" | sort | tail -1)
  python3 $R/tools/pmc_summary.py --reduce $f $ctr > $R/gpurun_out/${tag}_$ctr.reduced.json
  rm -rf $R/gpurun_out/${tag}_$ctr
  echo "$ctr done"
done
python3 $R/tools/pmc_summary.py --merge $R/gpurun_out/${tag}_FETCH_SIZE.reduced.json $R/gpurun_out/${tag}_WRITE_SIZE.reduced.json > $R/gpurun_out/${tag}_summary.json
head -c 1500 $R/gpurun_out/${tag}_summary.json
