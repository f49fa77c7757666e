#!/bin/bash
# exp85: adaptive terminal weights of the interfaces: agreement with the plain recursion across systems of very different cost scales; rates
O=gpurun_out/exp85; mkdir -p $O
timeout 900 python tools/dev/twolevel/agree.py CARTPOLE:100:512 CARTPOLE:25:64 CARTPOLE:5:8 CARTPOLE:2:4 VANDERPOL:40:32 TUMOUR:50:8 TUMOUR:20:3 CANCERTREATMENT:100:8 BIOREACTOR:50:8 PENDULUM:20:4 MOUNTAINCAR:60:16 GLUCOSE:50:8 HIVTREATMENT:50:8 SEIR:50:4 BACTERIA:50:8 MOULDFUNGICIDE:50:8 HARVEST:50:8 SIMPLECASE:30:8 TIMBERHARVEST:6:8 2>&1 | grep -v instance | grep waves | tee $O/agree.txt
rm -f $O/batch_sweep.jsonl
for B in 256 512; do timeout 300 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/batch_sweep.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/exp85/batch_sweep.jsonl"):
  d = json.loads(l); print("B", d["config"]["global_batch"], round(d["value"]), "solves/s kernel ms", d["solver_kernel"]["avg_ms"], "conv", d["converged_fraction"], d["iterations"])
PY
timeout 600 python tools/dev/node_bench.py 128 1024 2>&1 | grep config | tee $O/node.txt
timeout 900 python -m pytest tests/test_gpu_poison.py tests/test_gpu_reference_fixtures.py tests/test_gpu_solve.py tests/test_gpu_node.py -q -x -k "not shooting_batch" 2>&1 | tail -8
