#!/usr/bin/env python3
"""The two-wavefront fused kernel (W = 2, small batches) against the one-wavefront form, system by system: same status, iteration
count and optimum?  (a latent defect is suspected.)  On a GPU box: python tools/dev/w2_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
bad = 0; n = 0
for st in SystemType:
  if st.name in ("INVASIVEPLANT", "ROCKETLANDING"):
    continue
  for rule in ("HERMITE_SIMPSON", "TRAPEZOIDAL"):
    for N in (6, 20, 50, 100):
      for B in (1, 3):
        try:
          hp = HParams(system=st, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=N, nlpsolver=NLPSolverType.SQP)
          out = {}
          for W in ("1", "2"):
            os.environ["MYRIAD_FUSED_WAVES"] = W
            opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
            x0 = np.tile(opt.system.x_0, (B, 1)) * (1.0 + 0.01 * np.arange(B)[:, None])
            out[W] = opt.solve_batch(x0s=x0, max_iter=300)
        except Exception as e:
          print(f"{st.name} {rule} N={N} B={B}: {type(e).__name__} {str(e)[:80]}"); continue
        a, b = out["1"], out["2"]; n += 1
        same = np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"]) and np.allclose(a["cost"], b["cost"], rtol=1e-9, atol=1e-12, equal_nan=True)
        if not same:
          bad += 1
          print(f"MISMATCH {st.name} {rule} N={N} B={B}: W=1 status {a['status']} iters {a['iters']} cost {a['cost']} | W=2 status {b['status']} iters {b['iters']} cost {b['cost']}")
print(f"compared {n} cases, {bad} mismatches")
