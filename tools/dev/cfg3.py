import os, sys, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
rng = np.random.default_rng(2019)
hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()); B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4)
opt.solve_batch(x0s=x0); opt.engine.kernel_time_reset()
res = opt.solve_batch(x0s=x0)
ms, n = opt.engine.kernel_time(_lib.K_SOLVE)
print("LPW", os.environ.get("MYRIAD_SOLVE_LPW"), "B", B, "converged", (res['status'] == 0).mean(), "kernel ms %.2f" % ms, "solves/s %.0f" % (B / ms * 1e3), "iters med/p99/max", np.percentile(res['iters'], [50, 99, 100]))
