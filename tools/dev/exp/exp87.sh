#!/bin/bash
# exp87: fewer operand moves in the sweep stage (MYR_SWEEP_CARRY 1: zero halves inherited; 2: + no C tuple for the midpoint product): headline rate, same results?
O=gpurun_out/exp87; mkdir -p $O
for v in ${EXP87_VARIANTS:-default carry1 carry2}; do
  if [ $v = default ]; then unset MYRIAD_HIP_LIB; else export MYRIAD_HIP_LIB=$PWD/xv/lib$v.so; fi
  for rep in 1 2; do timeout 300 python bench.py --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), 'solves/s; solver kernel', d['solver_kernel']['avg_ms'], 'ms', d['iterations'], d['converged_fraction'])"; done
  MYRIAD_FUSED_WAVES=1 python - <<'PY'
import os, sys, hashlib, numpy as np
sys.path.insert(0, os.getcwd())
from myriad_amd import _lib
from bench import build_workload
x0, z0, lb, ub, T = build_workload(512, 100, 2019)
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, T, max_batch=512)
r = eng.solve(z0, lb, ub)
print("   W=1 B=512: z#", hashlib.sha1(np.ascontiguousarray(r["z"]).tobytes()).hexdigest()[:12], "iters sum", int(r["iters"].sum()), "status", np.bincount(r["status"]).tolist(), "cost sum %.15g" % r["cost"].sum())
PY
done 2>&1 | tee $O/carry.txt
