#!/bin/bash
# exp81: two-level sweep after the interface-delta fix: agreement, small-batch rates, network kernel
O=gpurun_out/exp81; mkdir -p $O
for mi in 2 5 1000; do echo "max_iter $mi"; AGREE_MAX_ITER=$mi timeout 300 python tools/dev/twolevel/agree.py CARTPOLE:100:64 CARTPOLE:10:16 2>&1 | grep -v -i Warn | tail -8; done 2>&1 | tee $O/agree_iters.txt
timeout 900 python tools/dev/twolevel/agree.py > $O/agree.txt 2>&1; grep -v "instance" $O/agree.txt | tail -20
rm -f $O/batch_sweep.jsonl
for B in 128 256 512; do timeout 300 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/batch_sweep.jsonl; done
python - <<'PY'
import json
for l in open("gpurun_out/exp81/batch_sweep.jsonl"):
  d = json.loads(l); print("B", d["config"]["global_batch"], round(d["value"]), "solves/s kernel ms", d["solver_kernel"]["avg_ms"], "conv", d["converged_fraction"], d["iterations"])
PY
timeout 600 python tools/dev/node_bench.py 128 256 1024 2>&1 | grep config | tee $O/node.txt
