import os, sys, numpy as np
sys.path.insert(0, "/root/repo") if os.path.isdir("/root/repo") else None
sys.path.insert(0, os.getcwd())
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
rng = np.random.default_rng(2019)
hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((8192, 2)), -4, 4)[int(os.environ.get("TRAJ", "0")):int(os.environ.get("TRAJ", "0")) + 1]
res = opt.solve_batch(x0s=x0)
print("status", res['status'], "iters", res['iters'], "cost", res['cost'])
print({k: (v if np.ndim(v) < 2 else v.shape) for k, v in res.items() if k not in ('x', 'u', 'xs_and_us', 'lambda')})
