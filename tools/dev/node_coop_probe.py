#!/usr/bin/env python3
"""Network dynamics (config 5): the fused kernel (four wavefronts share a trajectory; the default) against round 2's wavefront kernel
(MYRIAD_SOLVE_MODE=wave1, one wavefront per trajectory / its own cooperative mode), by N and B -- the question tools/dev/w2_probe.py
asked of the two-wavefront fused kernel of the closed-form systems."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
bad = n = 0
for N in (10, 20, 50, 100):
  for B in (1, 3, 12, 128, 300):
    hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=N, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
    out = {}
    rng = np.random.default_rng(N * 1000 + B)
    x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
    for coop, mode in (("1", "wave"), ("0", "wave1")):
      os.environ["MYRIAD_SOLVE_MODE"] = mode
      opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
      out[coop] = opt.solve_batch(x0s=x0, params=opt.system.device_params(), max_iter=300)
    a, b = out["0"], out["1"]; n += 1
    same = np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.allclose(a["cost"], b["cost"], rtol=1e-9)
    if not same:
      bad += 1
      d = np.nonzero((a["status"] != b["status"]) | (a["iters"] != b["iters"]) | ~np.isclose(a["cost"], b["cost"], rtol=1e-9))[0]
      print(f"MISMATCH N={N} B={B}: instances {d[:6]} round-2 kernel status {a['status'][d[:4]]} iters {a['iters'][d[:4]]} cost {a['cost'][d[:4]]} | fused status {b['status'][d[:4]]} iters {b['iters'][d[:4]]} cost {b['cost'][d[:4]]}")
print(f"compared {n} cases, {bad} mismatches")
