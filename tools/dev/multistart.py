"""dev experiment: second starts from excitation guesses (u = centre + 0.95 amp sin(2 pi c t / T), states = rollout) for systems whose
reference guess jams."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
name, N = sys.argv[1], int(sys.argv[2])
hp = HParams(system=getattr(SystemType, name), optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=N, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
s = opt.system; K = 2 * N + 1; t = np.linspace(0, s.T, K)
b = np.array(s.bounds, float); ns = len(s.x_0); nu = b.shape[0] - ns
lo, hi = b[ns:, 0], b[ns:, 1]
fin = np.isfinite(lo) & np.isfinite(hi)
centre = np.where(fin, 0.5 * (lo + hi), 0.0); amp = np.where(fin, 0.5 * (hi - lo), 1.0)
cycles = [1, 2, 3, 5, 8, 13]
us = np.stack([centre[None, :] + 0.95 * amp[None, :] * np.sin(2 * np.pi * c * t / s.T)[:, None] for c in cycles])     # [B, K, nu]
B = len(cycles)
x0 = np.tile(s.x_0, (B, 1))
xs, _ = opt.engine.rollout(x0, us, K - 1, params=s.device_params())
xl, xh = b[:ns, 0], b[:ns, 1]
w = np.where(np.isfinite(xh - xl), xh - xl, 1.0)
xs = np.clip(xs, np.where(np.isfinite(xl), xl + 1e-3 * w, -np.inf), np.where(np.isfinite(xh), xh - 1e-3 * w, np.inf))
xs = np.nan_to_num(xs, nan=0.0, posinf=1e3, neginf=-1e3)
guess = np.concatenate([xs.reshape(B, -1), us.reshape(B, -1)], 1)
r0 = opt.solve_batch()
r = opt.solve_batch(x0s=x0, guess=guess)
print(name, N, "reference guess: status", r0["status"], "cost", r0["cost"], "| excitation guesses (cycles", cycles, "): status", r["status"], "iters", r["iters"], "cost", np.round(r["cost"], 6))
