#!/bin/bash
# exp88: the join inlined (no callee-saved register traffic): small-batch rates, agreement
O=gpurun_out/exp88; mkdir -p $O
for B in 128 256 512; do timeout 300 python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B', d['config']['global_batch'], round(d['value']), 'solves/s kernel', d['solver_kernel']['avg_ms'], d['iterations'])"; done | tee $O/rates.txt
timeout 300 python tools/dev/node_bench.py 128 256 300 512 1024 2>&1 | grep config | cut -c1-170 | tee $O/node.txt
timeout 300 python tools/dev/twolevel/agree.py CARTPOLE:100:512 CARTPOLE:5:8 2>&1 | grep waves | cut -c1-250
