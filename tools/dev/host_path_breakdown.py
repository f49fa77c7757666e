#!/usr/bin/env python3
"""Where the wall clock of solve_batch goes on the headline workload (host buffers in and out): input packing, the copies inside
myr_solve (pageable vs pinned host arrays), the solver kernel, result handling.  On a GPU box: python tools/dev/host_path_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib

hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=100, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()); B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
rng = np.random.default_rng(2019)
x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
p = opt.system.device_params()
opt.solve_batch(x0s=x0)
eng = opt.engine
o = eng.default_opts(); o.max_iter = 1000

def t(f, n=5):
  f(); t0 = time.perf_counter()
  for _ in range(n): r = f()
  return (time.perf_counter() - t0) / n * 1e3, r

ms_all, _ = t(lambda: opt.solve_batch(x0s=x0))
ms_in, (z0, lb, ub) = t(lambda: opt.batch_inputs(x0, p))
eng.kernel_time_reset()
ms_solve, res = t(lambda: eng.solve(z0, lb, ub, p, o))
kms, _ = eng.kernel_time(_lib.K_SOLVE)
ms_dev, _ = t(lambda: opt.device_solve(z0, lb, ub, p, o))
print(f"B={B}: solve_batch {ms_all:.2f} ms | batch_inputs {ms_in:.2f} | Engine.solve {ms_solve:.2f} (kernel {kms:.2f}) | device_solve {ms_dev:.2f}")
# the same myr_solve call with pinned host arrays (torch's pinned allocator as the source of page-locked memory)
try:
  import torch
  def pinned(a):
    tt = torch.empty(a.shape, dtype=torch.float64 if a.dtype == np.float64 else torch.int32, pin_memory=True)
    v = tt.numpy(); v[...] = a; return tt, v
  keep = []
  zp = pinned(z0); lp = pinned(lb); up = pinned(ub); keep += [zp, lp, up]
  lam = pinned(np.empty((B, eng.m))); cost = pinned(np.empty(B)); kkt = pinned(np.empty((B, 3)))
  st = pinned(np.empty(B, np.int32)); it = pinned(np.empty(B, np.int32))
  import ctypes as C
  def call():
    zp[1][...] = z0
    _lib._chk(eng.lib.myr_solve(eng._h, B, _lib._addr(zp[1]), _lib._addr(lp[1]), _lib._addr(up[1]), None, 0, C.byref(o), _lib._addr(lam[1]),
                                _lib._addr(cost[1]), _lib._addr(st[1]), _lib._addr(it[1]), _lib._addr(kkt[1]), _lib.MEM_HOST), "myr_solve")
  ms_pin, _ = t(call)
  print(f"      myr_solve with pinned host arrays {ms_pin:.2f} ms (incl. {z0.nbytes/1e6:.0f} MB host memcpy of z0 into the pinned array)")
except Exception as e:
  print("pinned variant failed:", repr(e))
