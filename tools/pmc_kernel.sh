#!/bin/bash
# GPU box: SQ/TCC counters of the kernels whose name matches $1 while running the python tool $2 (separate --pmc passes).
# FETCH_SIZE (3 TCC slots) and WRITE_SIZE (2) do not fit the 4 TCC slots of one pass (MI355X_MICROARCH.md, rocprofv3 PMC slots):
# asked for together rocprofv3 aborts (signal 6; gpurun_out/r02l, r02m), so they are two passes.  FETCH_SIZE reads 1/2 of a wide
# coalesced stream on gfx950: tools/pmc_summary.py doubles it.
R=$GRAFT_REPO_ROOT
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmck_$i
  timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmck_$i -o out -- python3 $R/"$@" > $R/gpurun_out/pmck_$i.log 2>&1
  f=$(find $R/gpurun_out/pmck_$i -name "*counter_collection.csv" | head -1)
  echo "== $ctrs"
  if [ -n "$f" ]; then python3 - "$f" "$pat" <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if not re.search(sys.argv[2], k): continue
    k = re.sub(r"\(anonymous namespace\)::", "", k)
    k = re.sub(r"^void ", "", k).split("(")[0][:70]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("  ", k, {c: round(sum(v[1:]) / max(1, len(v) - 1)) for c, v in d.items()}, "n=%d" % len(next(iter(d.values()))))
PY
  else tail -3 $R/gpurun_out/pmck_$i.log; fi
  rm -rf $R/gpurun_out/pmck_$i $R/gpurun_out/pmck_$i.log
done
