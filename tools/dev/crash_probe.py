import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
name = sys.argv[1]
hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=20, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
print("solving", name, flush=True)
r = opt.solve_batch()
print(name, r["status"], r["iters"], r["cost"], flush=True)
