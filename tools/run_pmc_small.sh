#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/r02_counters_avail.txt 2>&1
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcs_$i
  timeout -k 10 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_$i -o out -- python3 $R/tools/pmc_small.py > $R/gpurun_out/pmcs_$i.log 2>&1
  f=$(find $R/gpurun_out/pmcs_$i -name "*counter_collection.csv" | head -1)
  echo "== $ctrs"
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "conv_mfma_bf16" not in k: continue
    acc[k[k.index("conv_mfma_bf16"):][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("  ", k, {c: round(sum(v[1:]) / max(1, len(v) - 1)) for c, v in d.items()})
PY
  else tail -3 $R/gpurun_out/pmcs_$i.log; fi
  rm -rf $R/gpurun_out/pmcs_$i
done
