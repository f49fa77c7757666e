# dev tool: README:83 literal config (CARTPOLE trapezoidal N=100, B=4096) on the GPU: status histogram, iterations, kernel time
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(int(os.environ.get("SEED", "5")))
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, intervals=100, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
res = opt.solve_batch(x0s=x0)
ms, n = opt.engine.kernel_time(_lib.K_SOLVE)
print("status hist", np.bincount(res['status']), "iters pct", np.percentile(res['iters'], [50, 99, 100]), "kernel ms", ms / max(n, 1))
bad = np.nonzero(res['status'] != 0)[0]
print("bad", bad[:20], res['iters'][bad][:20], "kkt", res['kkt'][bad][:3] if len(bad) else None)
