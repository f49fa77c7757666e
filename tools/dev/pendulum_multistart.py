# dev: does a perturbed guess get PENDULUM out of its locally infeasible one-swing strategy?
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
for opt_t, quad, kw in ((OptimizerType.COLLOCATION, QuadratureRule.HERMITE_SIMPSON, dict(intervals=50)), (OptimizerType.SHOOTING, QuadratureRule.TRAPEZOIDAL, dict(intervals=50, controls_per_interval=1))):
  hp = HParams(system=SystemType.PENDULUM, optimizer=opt_t, quadrature_rule=quad, nlpsolver=NLPSolverType.SQP, **kw)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  B = 256
  rng = np.random.default_rng(0)
  g = np.tile(opt.guess, (B, 1))
  nx = opt.x_guess.size
  for scale in (0.5, 2.0):
    gg = g.copy(); gg[:, nx:] += scale * rng.uniform(-1, 1, (B, g.shape[1] - nx))
    gg[0] = opt.guess
    r = opt.solve_batch(x0s=np.tile(hp.system().x_0, (B, 1)), guess=None)
    eng = opt.engine
    z0, lb, ub = opt.batch_inputs(np.tile(hp.system().x_0, (B, 1)), opt.system.device_params())
    o = eng.default_opts(); o.max_iter = 300
    res = eng.solve(np.clip(gg, lb + 1e-6 * (ub > lb), ub - 1e-6 * (ub > lb)), lb, ub, params=opt.system.device_params(), opts=o)
    ok = res["status"] == 0
    print(opt_t.name, quad.name, "noise", scale, "converged", ok.mean(), "costs", np.sort(res["cost"][ok])[:5], "median its", np.median(res["iters"][ok]) if ok.any() else None, flush=True)
