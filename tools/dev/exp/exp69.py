import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_DEBUG_ELASTIC"] = "1"
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
for rule, N in (("HERMITE_SIMPSON", 50), ("HERMITE_SIMPSON", 100), ("TRAPEZOIDAL", 100)):
  hp = HParams(system=SystemType.ROCKETLANDING, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=N, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  t = time.time(); r = opt.solve_batch(); dt = time.time() - t
  print("ROCKETLANDING", rule, N, "status", r["status"], "iters", r["iters"], "attempts", r["attempts"], "cost", r["cost"], "kkt", r["kkt"][0], f"{dt:.2f} s", flush=True)
