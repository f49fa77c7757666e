import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
rng = np.random.default_rng(2019)
hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()); B = 8192
x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4)
res = opt.solve_batch(x0s=x0)
np.savez(sys.argv[1], iters=res['iters'], status=res['status'], cost=res['cost'])
print(sys.argv[1], "converged", (res['status'] == 0).mean(), "iters med/max", np.median(res['iters']), res['iters'].max())
