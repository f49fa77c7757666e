#!/bin/bash
# exp31: the two-phase launch (park after k1 iterations, resume longest-first) against whole solves, headline workload
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp31
python -m pytest tests/test_gpu_poison.py -q -x -k "two_phase or headline" 2>&1 | grep -E "passed|failed|Error|assert" | head -20
run() { echo "== $*"; env "$@" python bench.py ${BARGS} --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['converged_fraction'], round(d['ms_per_step'],3), round(d['solver_kernel']['avg_ms'],3), d['iterations'])"; }
for k in 0 -1 6 8 10 12 14; do run MYRIAD_PARK_ITER=$k; done
BARGS="--batch 2048" run MYRIAD_PARK_ITER=0; BARGS="--batch 2048" run MYRIAD_PARK_ITER=10
BARGS="--batch 8192 --steps 6" run MYRIAD_PARK_ITER=0; BARGS="--batch 8192 --steps 6" run MYRIAD_PARK_ITER=10
BARGS="--batch 16384 --steps 4 --warmup 1" run MYRIAD_PARK_ITER=0; BARGS="--batch 16384 --steps 4 --warmup 1" run MYRIAD_PARK_ITER=10
