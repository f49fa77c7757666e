#!/bin/bash
# exp49: README:83's literal config (tools/bench_configs.py's draw of start states) under the two-phase launch
cd $GRAFT_REPO_ROOT
for k in 0 5 6 7 8; do
MYRIAD_PARK_ITER=$k python - <<'PY'
import os, numpy as np
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, intervals=100, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()); opt.devices = [0]
for seed in (2019, 7):
  B = 4096; x0 = np.clip(0.1 * np.random.default_rng(seed).standard_normal((B, 4)), -2, 2)
  opt.solve_batch(x0s=x0); ts = []
  for _ in range(5):
    opt.engine.kernel_time_reset(); r = opt.solve_batch(x0s=x0); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms * max(1, n))
  it = r["iters"]
  print("PARK_ITER", os.environ["MYRIAD_PARK_ITER"], "seed", seed, "kernel ms median %.3f" % np.median(ts), "converged", (r["status"] == 0).mean(), "iters median %g p99 %g max %g" % (np.median(it), np.percentile(it, 99), it.max()))
PY
done
