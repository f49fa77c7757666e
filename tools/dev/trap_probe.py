# dev tool: trapezoidal solves of systems with more than one control, wave vs lane kernel
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
import numpy as np
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, IntegrationMethod, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
for name in ("BEARPOPULATIONS", "PENDULUM"):
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=30, max_iter=1000)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  r = opt.solve_batch()
  print(name, "TRAP", os.environ.get("MYRIAD_SOLVE_MODE"), r["status"], r["iters"], r["cost"], r["kkt"], flush=True)
  if name == "PENDULUM":
    z0, lb, ub = opt.batch_inputs(np.tile(opt.system.x_0, (1, 1)), None)
    tw = opt._twin_engine() if os.environ.get("X") else None
    os.environ["MYRIAD_ELASTIC"] = "1"
    tw = opt._twin_engine()
    (rows_x, ns), (rows_u, nu) = opt._x_shape, opt._u_shape
    nx = rows_x * ns
    def widen(a, fill):
      U = a[:, nx:].reshape(1, rows_u, nu)
      return np.concatenate([a[:, :nx], np.concatenate([U, np.full((1, rows_u, ns), fill)], axis=2).reshape(1, -1)], axis=1)
    o = tw.default_opts(); o.max_iter = 300
    for rho in (1.0, 100.0):
      rr = tw.solve(widen(z0, 0.0), widen(lb, -np.inf), widen(ub, np.inf), params=np.concatenate([opt.system.device_params(), [rho]]), opts=o)
      s = rr["z"][:, nx:].reshape(rows_u, 3)[:, 1:]
      print("  twin rho", rho, rr["status"], rr["iters"], rr["cost"], rr["kkt"], "slack rows", np.round(s[:4].ravel(), 3), "...", np.round(s[-3:].ravel(), 3), flush=True)
