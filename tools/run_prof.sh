#!/bin/bash
# GPU box: rocprofv3 kernel stats of the default bench (arg 1 = output tag)
set -e
tag=${1:-prof}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
rm -rf $R/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -o out -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-baseline none > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
cd $R
f=$(find gpurun_out/$tag -name "*kernel_stats.csv" | sort | tail -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
head -40 gpurun_out/${tag}_kernel_stats.csv | cut -c1-150
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --cpu-baseline none > gpurun_out/${tag}_unprofiled.json 2>/dev/null
cat gpurun_out/${tag}_unprofiled.json | cut -c1-200
