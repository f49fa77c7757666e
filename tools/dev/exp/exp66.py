"""exp66: what the twin solves of the elastic phase cost on the problems that need it (MYRIAD_DEBUG_ELASTIC prints status / iterations per rho)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_DEBUG_ELASTIC"] = "1"
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
for name, rule, N in (("PENDULUM", "HERMITE_SIMPSON", 20), ("PENDULUM", "HERMITE_SIMPSON", 50), ("PENDULUM", "TRAPEZOIDAL", 40), ("PENDULUM", "TRAPEZOIDAL", 100),
                      ("MOUNTAINCAR", "TRAPEZOIDAL", 40), ("MOUNTAINCAR", "HERMITE_SIMPSON", 40), ("ROCKETLANDING", "HERMITE_SIMPSON", 20)):
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=N, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  r = opt.solve_batch()
  print(name, rule, N, "status", r["status"], "iters", r["iters"], "attempts", r["attempts"], "restored", r["restored"], "cost", r["cost"], flush=True)
