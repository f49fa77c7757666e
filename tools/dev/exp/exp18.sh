#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --batch 16384 --steps 4 --warmup 1 --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['converged_fraction'], round(d['solver_kernel']['avg_ms'],2), d['iterations'])"; }
run A=1
run MYRIAD_MU_INIT=0.001
run MYRIAD_MU_INIT=0.003
run MYRIAD_MU_INIT=0.01 MYRIAD_KAPPA_MU=0.1 MYRIAD_KAPPA_EPS=30
run MYRIAD_MU_INIT=0.003 MYRIAD_KAPPA_MU=0.1 MYRIAD_KAPPA_EPS=30
run MYRIAD_MU_INIT=0.001 MYRIAD_KAPPA_MU=0.1 MYRIAD_KAPPA_EPS=30
run MYRIAD_MU_INIT=0.01 MYRIAD_KAPPA_EPS=30
run MYRIAD_MU_INIT=0.001 MYRIAD_KAPPA_EPS=30
run A=1
