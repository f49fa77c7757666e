#!/usr/bin/env python3
"""Throughput of the SURVEY.md 8(f) rows on one GPU (informative): HS-collocation solves of the (f4) systems with start
states perturbed by the reference's rule (x0 + 0.1 |x0| N(0,I), clipped to the bounds), and the batched FBSM (f3) on a
CANCERTREATMENT parameter sweep."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
CFG = Config(verbose=False, plot=False)
rng = np.random.default_rng(2019)
B, N = 4096, 50
for name in ["BIOREACTOR", "GLUCOSE", "MOULDFUNGICIDE", "SIMPLECASEWITHBOUNDS", "HIVTREATMENT", "EPIDEMICSEIRN", "SEIR", "BEARPOPULATIONS", "MOUNTAINCAR"]:
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=N, nlpsolver=NLPSolverType.SQP)
  s = hp.system(); opt = get_optimizer(hp, CFG, s)
  ns = s.x_0.shape[0]
  x0 = s.x_0 * (1.0 + 0.1 * rng.standard_normal((B, ns)))
  x0 = np.clip(x0, np.where(np.isfinite(s.bounds[:ns, 0]), s.bounds[:ns, 0] + 1e-3, -np.inf), np.where(np.isfinite(s.bounds[:ns, 1]), s.bounds[:ns, 1] - 1e-3, np.inf))
  opt.solve_batch(x0s=x0)
  opt.engine.kernel_time_reset()
  res = opt.solve_batch(x0s=x0)
  ms, n = opt.engine.kernel_time(_lib.K_SOLVE)
  print(json.dumps(dict(row="f4", system=name, transcription="HERMITE_SIMPSON", N=N, B=B, converged=float((res['status'] == 0).mean()), kernel_ms=ms,
                        solves_per_s_kernel=B / ms * 1e3, its_median=float(np.median(res['iters'])), its_max=int(res['iters'].max()))), flush=True)
# f3: FBSM parameter sweep
hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.FBSM, fbsm_intervals=1000)
opt = get_optimizer(hp, CFG, hp.system()); Bf = 8192
P = np.stack([rng.uniform(0.1, 0.5, Bf), rng.uniform(1, 5, Bf), rng.uniform(0.2, 0.8, Bf)], 1); x0 = rng.uniform(0.5, 0.99, (Bf, 1))
CAP = 200     # the reference's while-loop has no cap; instances whose fixed-point iteration oscillates stop here
opt.solve_batch(x0s=x0[:64], params=P[:64], max_sweeps=CAP)
opt.engine.kernel_time_reset()
r = opt.solve_batch(x0s=x0, params=P, max_sweeps=CAP)
ms, n = opt.engine.kernel_time(_lib.K_FBSM)
print(json.dumps(dict(row="f3", system="CANCERTREATMENT", method="FBSM N=1000", B=Bf, kernel_ms=ms, instances_per_s=Bf / ms * 1e3, sweeps_median=float(np.median(r['sweeps'])),
                      stopped_by_rule=float((r['sweeps'] < CAP).mean()), sweep_cap=CAP, rk4_steps_per_s=float(2000.0 * r['sweeps'].sum() / ms * 1e3))))
