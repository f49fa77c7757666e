#!/bin/bash
# exp48: the two-phase launch under the trapezoidal scheme (README:83's literal config), phase-1 length
cd $GRAFT_REPO_ROOT
for k in 0 5 6 8 10; do
MYRIAD_PARK_ITER=$k python - <<'PY'
import os, numpy as np, bench
from myriad_amd import _lib
N, B = 100, 4096
x0, z0h, lbh, ubh, T = bench.build_workload(B, N, 2019)
# the trapezoidal problem of the same start states, through the host API
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=N, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
opt.solve_batch(x0s=x0)
ts = []
for _ in range(5):
  opt.engine.kernel_time_reset(); r = opt.solve_batch(x0s=x0); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms)
it = r["iters"]
print("PARK_ITER", os.environ["MYRIAD_PARK_ITER"], "kernel ms", np.round(ts, 3), "converged", (r["status"] == 0).mean(), "iters median %g p99 %g max %g" % (np.median(it), np.percentile(it, 99), it.max()))
PY
done
