"""dev experiment: PENDULUM swing-up (HS, N given) from control guesses u = umax sin(w t) / umax sign(sin(w t)) with the states of
their rollouts, instead of the reference's straight-line guess (which jams at an infeasible point)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
N = int(os.environ.get("N", "50"))
hp = HParams(system=SystemType.PENDULUM, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=N, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
s = opt.system; K = 2 * N + 1; t = np.linspace(0, s.T, K)
ws = np.linspace(0.5, 4.5, 33); umax = s.bounds[2][1]
us = np.concatenate([umax * np.sin(ws[:, None] * t[None]), umax * np.sign(np.sin(ws[:, None] * t[None])) * 0.95], 0)
B = us.shape[0]
x0 = np.tile(s.x_0, (B, 1))
xs, _ = opt.engine.rollout(x0, us[:, :, None], K - 1, params=s.device_params())
xs = np.clip(xs, np.array(s.bounds)[:2, 0] * 0.999, np.array(s.bounds)[:2, 1] * 0.999)
guess = np.concatenate([xs.reshape(B, -1), np.clip(us, -0.999 * umax, 0.999 * umax)], 1)
r = opt.solve_batch(x0s=x0, guess=guess)
print("converged", (r["status"] == 0).sum(), "of", B, "costs", np.sort(r["cost"][r["status"] == 0])[:8], "iters", r["iters"][r["status"] == 0][:8])
print("status", np.bincount(r["status"]), "best feas", r["kkt"][:, 0].min())
