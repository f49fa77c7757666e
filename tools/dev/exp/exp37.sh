#!/bin/bash
# exp37: what the device does around the sporadic 6 ms the host spends in the download enqueue (two-phase launch)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp37
MYRIAD_PARK_ITER=12 MYRIAD_BENCH_TRACE=1 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/exp37/kt -o kt -- python bench.py --steps 30 --warmup 2 --cpu-budget 0 --no-other-configs > gpurun_out/exp37/bench.json 2> gpurun_out/exp37/err.txt
grep "step\|slow" gpurun_out/exp37/err.txt | cut -c1-200
ls gpurun_out/exp37/kt/*/ 2>/dev/null | head
python - <<'PY'
import csv, glob
kf = glob.glob("gpurun_out/exp37/kt/**/*kernel_trace.csv", recursive=True)[0]
mf = glob.glob("gpurun_out/exp37/kt/**/*memory_copy_trace.csv", recursive=True)[0]
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:50]) for r in csv.DictReader(open(kf))]
rows = list(csv.DictReader(open(mf)))
print(rows[0].keys())
M = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M %s %s bytes" % (r.get("Direction", r.get("Name", "")), r.get("Bytes", r.get("Size", "")))) for r in rows]
ev = sorted(K + M)
slow = [i for i, e in enumerate(ev) if e[2].startswith("M") and e[1] - e[0] > 2e6]
print("copies > 2 ms:", len(slow), " all copies:", len(M))
big = sorted(((e[1] - e[0]) / 1e3, e[2]) for e in ev if e[2].startswith("M"))[-8:]
print("longest copies (us):", big)
for i in slow[:2]:
  for s, e, n in ev[max(0, i - 12): i + 8]:
    print("%14.1f  %10.1f us  %s" % ((s - ev[i][0]) / 1e3, (e - s) / 1e3, n))
  print("----")
PY
rm -rf gpurun_out/exp37/kt
