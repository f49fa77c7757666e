#!/usr/bin/env python3
"""ROCKETLANDING (NS = 6, NU = 2) through the wavefront kernel's general sweep and the lane kernel, iteration by iteration
(max_iter = 1, 2, ...): where do the two paths part?  On a GPU box: python tools/dev/rocket_probe.py [HERMITE_SIMPSON|TRAPEZOIDAL]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"; os.environ["MYRIAD_LANE_UNVERIFIED"] = "1"
rule = sys.argv[1] if len(sys.argv) > 1 else "HERMITE_SIMPSON"
sysname = sys.argv[2] if len(sys.argv) > 2 else "ROCKETLANDING"
hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=20, nlpsolver=NLPSolverType.SQP)
for it in (0, 1, 2, 3, 4, 6, 8, 12, 20, 40):
  out = {}
  for mode in ("wave", "lane"):
    os.environ["MYRIAD_SOLVE_MODE"] = mode
    out[mode] = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()).solve_batch(max_iter=it)
  w, l = out["wave"], out["lane"]
  d = np.abs(w["xs_and_us"] - l["xs_and_us"]) / np.maximum(1.0, np.abs(l["xs_and_us"]))
  print(f"max_iter {it:3d}: wave status {w['status'][0]} it {w['iters'][0]} cost {w['cost'][0]:.6g} kkt {w['kkt'][0]} | lane status {l['status'][0]} it {l['iters'][0]} cost {l['cost'][0]:.6g} kkt {l['kkt'][0]} | max rel dz {np.nanmax(d):.3e} nan {np.isnan(l['xs_and_us']).sum()}/{np.isnan(w['xs_and_us']).sum()}")
