#!/bin/bash
# exp32: timeline of a two-phase step (rocprofv3 kernel trace): phase 1, ordering kernels, phase 2, gaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp32
MYRIAD_PARK_ITER=${K1:-12} timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/exp32/kt -o kt -- python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-other-configs > gpurun_out/exp32/bench.json 2> gpurun_out/exp32/err.txt
f=$(find gpurun_out/exp32/kt -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows]
# print the last ~40 kernels with gaps
prev_end = None
for s, e, n in ev[-45:]:
  gap = (s - prev_end) / 1e3 if prev_end else 0.0
  print("%9.1f us  gap %8.1f us  %s" % ((e - s) / 1e3, gap, n))
  prev_end = e
PY
rm -rf gpurun_out/exp32/kt
