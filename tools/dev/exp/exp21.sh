#!/bin/bash
# exp21: skipping the doomed probe of a smaller inertia shift (pivot-margin predictor), fused kernel, CARTPOLE
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --mu-init 0 --steps 10 --warmup 2 --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['converged_fraction'], round(d['solver_kernel']['avg_ms'],2), d['iterations'])"; }
for l in libpp00 libpp10 libpp05 libpp00 libpp10 libpp05; do run MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/$l.so; done
run MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/libpp10.so MYRIAD_MU_INIT=0.003
