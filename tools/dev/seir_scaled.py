# dev: does EPIDEMICSEIRN converge when posed in units of 1000 individuals (uniform state scaling == parameter change)?
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType, EpidemicSEIRN
from myriad_amd.trajectory_optimizers import get_optimizer
for S in (1.0, 10.0, 100.0, 1000.0):
  hp = HParams(system=SystemType.EPIDEMICSEIRN, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=20, nlpsolver=NLPSolverType.SQP)
  s = EpidemicSEIRN(A=.1 * S, c=.0001 * S, x_0=(1000. / S, 100. / S, 50. / S, 15. / S))
  opt = get_optimizer(hp, Config(verbose=False, plot=False), s)
  r = opt.solve_batch()
  print("S", S, "status", r['status'], "iters", r['iters'], "kkt", r['kkt'], "cost", r['cost'])
