import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
name = sys.argv[1]; mi = int(sys.argv[2]); N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=N, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
z0, lb, ub = opt.batch_inputs(np.tile(opt.system.x_0, (1, 1)), opt.system.device_params())
o = opt.engine.default_opts(); o.max_iter = mi
r = opt.engine.solve(z0, lb, ub, params=opt.system.device_params(), opts=o)
print(name, "max_iter", mi, "N", N, r["status"], r["iters"], r["cost"], r["kkt"], flush=True)
