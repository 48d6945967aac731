#!/bin/bash
# GPU box: per-dispatch kernel trace of ONE eager RAFT call (encoders + 1 refinement iteration), in launch order.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_raft
cat > /tmp/one_raft.py <<'PY'
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
ops_mod = importlib.import_module("zero-tig_amd.ops"); lib_mod = importlib.import_module("zero-tig_amd.lib")
raft_mod = importlib.import_module("zero-tig_amd.raft"); synth = importlib.import_module("zero-tig_amd.synth")
ops = ops_mod.Ops(lib_mod.get_lib()); dev = torch.device("cuda:0")
st = synth.make_state(1)
W = {k: torch.from_numpy(np.array(v)).to(dev) for k, v in st.items() if k.startswith("raft.")}
os.environ["ZT_RAFT_STREAMS"] = "1"
plan = raft_mod.RaftPlan(ops, W, dev, precision="bf16")
x2 = (torch.randn(2, 360, 640, 8, device=dev) * 0.5).bfloat16()
for _ in range(3):
    plan.run(x2, iters=1)
torch.cuda.synchronize()
PY
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_raft -o out -- python3 /tmp/one_raft.py > $R/gpurun_out/trace_raft.log 2>&1
f=$(find $R/gpurun_out/trace_raft -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 3
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
tot = 0
for r in last:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k).split("(")[0][:58]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); tot += d
    print("%8.1f us  +%7.1f  %-58s grid %s" % (d / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3, k, r.get("Grid_Size", "") + "/" + r.get("Workgroup_Size", "")))
print("sum of kernel durations %.1f us over %d launches; span %.1f us" % (tot / 1e3, n, (int(last[-1]["End_Timestamp"]) - t0) / 1e3))
PY
rm -rf $R/gpurun_out/trace_raft
