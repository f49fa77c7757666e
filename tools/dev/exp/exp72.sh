#!/bin/bash
# exp72: the two launches of a headline solve, one by one (rocprofv3 kernel trace of the bench): how long phase 1 (12 iterations for everybody) and phase 2
# (the parked trajectories resumed longest-first) take.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kt72; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt72 -o kt -- python bench.py --steps 6 --warmup 2 --cpu-budget 0 --no-other-configs > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/kt72/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "hs_solve_fused_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("fused-kernel dispatches (ms):", " ".join(f"{x:.2f}" for x in d))
PY
