#!/usr/bin/env python3
"""ROCKETLANDING Hermite-Simpson on the LANE kernel: cost of the first iterate (max_iter = 0; the wavefront kernel and the host twin
say 2.637376) by batch size and active lanes per wavefront -- does the wrong value (1.80221 at B = 1) depend on the lanes switched off?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"; os.environ["MYRIAD_SOLVE_MODE"] = "lane"; os.environ["MYRIAD_LANE_UNVERIFIED"] = "1"
hp = HParams(system=SystemType.ROCKETLANDING, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=20, nlpsolver=NLPSolverType.SQP)
for vs in ("1", "0"):
  os.environ["MYRIAD_VAR_SCALE"] = vs
  for lpw in (1, 16, 64):
    os.environ["MYRIAD_SOLVE_LPW"] = str(lpw)
    for B in (1, 16, 64, 65):
      opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
      r = opt.solve_batch(x0s=np.tile(opt.system.x_0, (B, 1)), max_iter=0)
      r1 = opt.solve_batch(x0s=np.tile(opt.system.x_0, (B, 1)), max_iter=1)
      print(f"var_scale {vs} lpw {lpw:2d} B {B:2d}: cost it0 {np.unique(np.round(r['cost'], 6))}  it1 {np.unique(np.round(r1['cost'], 6))} feas {np.unique(np.round(r1['kkt'][:, 0], 6))}")
