#!/bin/bash
# exp84: W = 4 for closed-form small batches, trapezoidal two-level sweep: agreement and rates; the whole GPU suite
O=gpurun_out/exp84; mkdir -p $O
AGREE_WAVES=1,2,4 timeout 600 python tools/dev/twolevel/agree.py CARTPOLE:100:256 CARTPOLE:25:64 CARTPOLE:5:8 CARTPOLE:3:4 CARTPOLE:2:4 VANDERPOL:40:32 TIMBERHARVEST:6:8 2>&1 | grep -v instance | grep waves | tee $O/agree.txt
rm -f $O/batch_sweep.jsonl
for B in 128 256 512; do timeout 300 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/batch_sweep.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/exp84/batch_sweep.jsonl"):
  d = json.loads(l); print("B", d["config"]["global_batch"], round(d["value"]), "solves/s kernel ms", d["solver_kernel"]["avg_ms"], "waves", d["solver_kernel"].get("waves_per_trajectory"), "conv", d["converged_fraction"], d["iterations"])
PY
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_smoke.py 2>&1 | tail -15 | tee $O/pytest.txt
