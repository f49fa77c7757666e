#!/bin/bash
# exp47: phase-1 length against the batch size (solves per resident wavefront)
cd $GRAFT_REPO_ROOT
run() { env MYRIAD_PARK_ITER=$2 python bench.py --batch $1 --steps $3 --warmup 2 --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('B=$1 K1=$2', round(d['value']), round(d['solver_kernel']['avg_ms'],3))"; }
for B in 1280 1536; do for k in 0 8 12; do run $B $k 20; done; done
for B in 2048 3072; do for k in 0 8 10 12; do run $B $k 20; done; done
for B in 8192; do for k in 0 10 12 14 16; do run $B $k 8; done; done
