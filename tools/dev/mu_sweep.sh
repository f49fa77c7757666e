#!/bin/bash
# dev tool: barrier-schedule ablation on the bench workload (env knobs read by make_opts)
run() { echo "== $*"; env "$@" python bench.py --steps 3 --warmup 1 --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['converged_fraction'], d['solver_kernel']['avg_ms'], d['iterations'])"; }
run A=1
run MYRIAD_MU_INIT=0.01
run MYRIAD_MU_INIT=0.001
run MYRIAD_KAPPA_MU=0.1
run MYRIAD_KAPPA_MU=0.05
run MYRIAD_THETA_MU=2.0
run MYRIAD_KAPPA_EPS=30
run MYRIAD_KAPPA_EPS=100
run MYRIAD_KAPPA_MU=0.1 MYRIAD_KAPPA_EPS=30
run MYRIAD_MU_INIT=0.01 MYRIAD_KAPPA_MU=0.1 MYRIAD_KAPPA_EPS=30
run MYRIAD_MU_INIT=0.03
run MYRIAD_MU_INIT=0.3
run MYRIAD_MU_INIT=1.0
run MYRIAD_THETA_MU=1.7
run MYRIAD_KAPPA_MU=0.3
run MYRIAD_KAPPA_EPS=5
run MYRIAD_KAPPA_EPS=20 MYRIAD_THETA_MU=1.7
run MYRIAD_NONMONO=0
run MYRIAD_NONMONO=5
run MYRIAD_DELTA_WARM=0
