# dev: does a homotopy on the control bounds get PENDULUM (HS, N=20/50) from the reference guess to a feasible optimum?
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
for name, N in (("PENDULUM", 20), ("PENDULUM", 50), ("ROCKETLANDING", 20)):
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, nlpsolver=NLPSolverType.SQP, intervals=N)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  eng = opt.engine
  z0, lb, ub = opt.batch_inputs(hp.system().x_0[None], opt.system.device_params())
  o = eng.default_opts(); o.max_iter = 300
  r = eng.solve(z0, lb, ub, params=opt.system.device_params(), opts=o)
  print(name, N, "plain: status", r["status"], "iters", r["iters"], "cost", r["cost"], "feas", r["kkt"][:, 0])
  nx = opt.x_guess.size
  for sched in ((4, 2, 1.4, 1.0), (8, 4, 2, 1.4, 1.0), (3, 1.0), (2.5, 1.7, 1.3, 1.1, 1.0)):
    z = z0.copy(); tot = 0
    for a in sched:
      l2, u2 = lb.copy(), ub.copy()
      l2[:, nx:] *= a; u2[:, nx:] *= a
      z = np.minimum(np.maximum(z, l2), u2)
      r = eng.solve(z, l2, u2, params=opt.system.device_params(), opts=o)
      tot += int(r["iters"][0]); z = r["z"]
      if r["status"][0] != 0: break
    print("  schedule", sched, "-> status", r["status"], "at factor", a, "total iters", tot, "cost", r["cost"], "feas", r["kkt"][:, 0])
