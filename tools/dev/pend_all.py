import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
CFG = Config(verbose=False, plot=False)
cases = [dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=60),
         dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=40, integration_method=IntegrationMethod.RK4),
         dict(optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=60, integration_method=IntegrationMethod.HEUN),
         dict(optimizer=OptimizerType.SHOOTING, intervals=6, controls_per_interval=10, integration_method=IntegrationMethod.HEUN),
         dict(optimizer=OptimizerType.SHOOTING, intervals=3, controls_per_interval=10, integration_method=IntegrationMethod.RK4)]
for kw in cases:
  hp = HParams(system=SystemType.PENDULUM, nlpsolver=NLPSolverType.SQP, **kw)
  opt = get_optimizer(hp, CFG, hp.system())
  r = opt.solve_batch()
  print({k: getattr(v, "name", v) for k, v in kw.items()}, "status", r["status"], "iters", r["iters"], "cost", r["cost"], "max|c|", np.abs(opt.constraints(r["xs_and_us"][0])).max())
