#!/bin/bash
# exp79: first run of the two-level sweep (round 6): agreement with the one-wavefront kernel, the small-batch rates, the network kernel.
O=gpurun_out/exp79; mkdir -p $O
timeout 900 python tools/dev/twolevel/agree.py > $O/agree.txt 2>&1; tail -40 $O/agree.txt
for B in 128 256 512; do timeout 300 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/batch_sweep.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/exp79/batch_sweep.jsonl"):
  d = json.loads(l); print("B", d["config"]["global_batch"], round(d["value"]), "solves/s kernel ms", d["solver_kernel"]["avg_ms"], "conv", d["converged_fraction"], d["iterations"])
PY
timeout 600 python tools/dev/node_bench.py 128 256 1024 2>&1 | grep -v Warning | tail -5 | tee $O/node.txt
