#!/bin/bash
# GPU box: small-map conv timings for the tile-shape knobs
cd $GRAFT_REPO_ROOT
for cfg in "0 0" "2 2" "1 4" "2 4" "1 1" "2 1"; do
  set -- $cfg
  echo "== ZT_TILED_MT=$1 ZT_TILED_NT=$2"
  ZT_TILED_MT=$1 ZT_TILED_NT=$2 timeout -k 10 120 python tools/bench_small.py 2>&1 | grep -v amdgpu.ids | cut -c1-60,72-82,96-106,118-128 || exit 1
done
