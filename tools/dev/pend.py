import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
name = sys.argv[1] if len(sys.argv) > 1 else "PENDULUM"
hp = HParams(system=getattr(SystemType, name), optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=int(os.environ.get("N", "50")), nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
r = opt.solve_batch(x0s=np.array([opt.system.x_0], float))
print(name, "mu", os.environ.get("MYRIAD_MU_INIT"), "status", r["status"], "iters", r["iters"], "cost", r["cost"], "kkt", r["kkt"])
