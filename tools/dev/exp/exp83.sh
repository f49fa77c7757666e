#!/bin/bash
# exp83: lane-parallel join + the fold written by the hessian pass: agreement and rates
O=gpurun_out/exp83; mkdir -p $O
timeout 600 python tools/dev/twolevel/agree.py CARTPOLE:100:512 CARTPOLE:25:64 CARTPOLE:5:8 CARTPOLE:2:4 CARTPOLE:1:2 2>&1 | grep -v instance | grep waves | tee $O/agree.txt
rm -f $O/batch_sweep.jsonl
for B in 128 256 512; do timeout 300 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/batch_sweep.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/exp83/batch_sweep.jsonl"):
  d = json.loads(l); print("B", d["config"]["global_batch"], round(d["value"]), "solves/s kernel ms", d["solver_kernel"]["avg_ms"], "conv", d["converged_fraction"], d["iterations"])
PY
timeout 600 python tools/dev/node_bench.py 128 256 1024 2>&1 | grep config | tee $O/node.txt
