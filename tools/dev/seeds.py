"""dev tool: convergence of the headline workload (CARTPOLE HS N=100, B=4096) and of config 3 (VANDERPOL shooting 1x50,
B=8192) over several seeds."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_workload
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
seeds = [int(a) for a in sys.argv[1:]] or [2019, 2020, 2021, 2022, 2023]
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, 2.0, max_batch=4096)
for sd in seeds:
  x0, z0, lb, ub, T = build_workload(4096, 100, sd)
  r = eng.solve(z0, lb, ub)
  print("HS seed", sd, "converged", (r["status"] == 0).mean(), "status counts", np.bincount(r["status"]), "iters med/p99/max", np.percentile(r["iters"], [50, 99, 100]))
eng.close()
hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
for sd in seeds:
  rng = np.random.default_rng(sd)
  x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((8192, 2)), -4, 4)
  r = opt.solve_batch(x0s=x0)
  print("config 3 seed", sd, "converged", (r["status"] == 0).mean(), "status counts", np.bincount(r["status"]), "iters med/p99/max", np.percentile(r["iters"], [50, 99, 100]))
