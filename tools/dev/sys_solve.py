# dev: solve one system through the host API at a few sizes and print status
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
name = sys.argv[1]
for N in [int(a) for a in sys.argv[2:]]:
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=N, nlpsolver=NLPSolverType.SQP)
  r = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()).solve_batch()
  print(name, "N", N, "status", r['status'], "iters", r['iters'], "kkt", r['kkt'], "cost", r['cost'], flush=True)
