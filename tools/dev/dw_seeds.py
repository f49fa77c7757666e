import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
for sd in (2019, 2020, 2021, 2022, 7):
  rng = np.random.default_rng(sd); x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((8192, 2)), -4, 4)
  opt.solve_batch(x0s=x0); opt.engine.kernel_time_reset(); r = opt.solve_batch(x0s=x0); ms, n = opt.engine.kernel_time(_lib.K_SOLVE)
  print("DW", os.environ.get("MYRIAD_DELTA_WARM"), "seed", sd, "kernel ms %.2f" % ms, "solves/s %.0f" % (8192 / ms * 1e3), "conv", (r["status"] == 0).mean(), "its med/p99/max", np.percentile(r["iters"], [50, 99, 100]), "launches", n)
