#!/usr/bin/env python3
"""Throughput of the other BASELINE configs on one GPU (informative; bench.py is the headline config 2)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
CFG = Config(verbose=False, plot=False)

def _pinned(shape, dtype):
  """numpy view of page-locked host memory (torch's allocator) -- plain numpy when torch is not importable"""
  try:
    import torch
    t = torch.empty(shape, dtype={np.float64: torch.float64, np.int32: torch.int32}[dtype]).pin_memory()
    _KEEP.append(t)
    return t.numpy()
  except Exception:
    return np.empty(shape, dtype=dtype)
_KEEP = []

def timed(opt, reps=10, **kw):
  """`reps` calls of solve_batch on ONE device (opt.devices = [0]: the figures are single-GPU figures whatever the box holds), results
  into caller-provided pinned buffers (Engine.result_buffers: no allocation, no page faults in the timed calls); returns the last
  result, the MEDIAN and the MINIMUM wall clock of a call, and the library's average solve-kernel time."""
  opt.devices = [0]
  B = (kw.get("x0s") if kw.get("x0s") is not None else kw["params"]).shape[0]
  e = opt.engine
  e.result_buffers = {"z": _pinned((B, e.n), np.float64), "lam": _pinned((B, e.m), np.float64), "cost": _pinned((B,), np.float64),
                      "status": _pinned((B,), np.int32), "iters": _pinned((B,), np.int32), "kkt": _pinned((B, 3), np.float64)}
  opt.solve_batch(**kw); opt.solve_batch(**kw)        # warm-up (handle scratch, first-touch of the buffers)
  e.kernel_time_reset()
  ts = []
  for _ in range(reps):
    t0 = time.perf_counter()
    res = opt.solve_batch(**kw)
    ts.append(time.perf_counter() - t0)
  ms, n = e.kernel_time(_lib.K_SOLVE)
  launches_per_call = max(1, round(n / reps))
  res = {k: np.array(v) for k, v in res.items()}   # (the buffers are reused by the next config of the same batch size)
  e.result_buffers = None
  return res, float(np.median(ts)), float(np.min(ts)), ms * launches_per_call


def _line(config, B, res, med, mn, ms):
  return dict(config=config, B=B, converged=float((res['status'] == 0).mean()), kernel_ms=ms, wall_ms=1e3 * med, wall_ms_min=1e3 * mn, reps=10,
              solves_per_s_wall=B / med, solves_per_s_kernel=B / ms * 1e3, its_median=float(np.median(res['iters'])), attempts_max=int(res['attempts'].max()),
              device="one GPU (devices=[0]); results into pinned caller buffers; wall = median of 10 calls")



def measure():
  """One line per config: converged fraction, solve-kernel time (HIP events inside the library) and the wall clock of
  solve_batch (host buffers in and out: PCIe-inclusive; `attempts_max` > 1 means the wall clock includes second starts of instances
  the first launch left without a KKT point).  bench.py attaches the list to its JSON line as `other_configs`."""
  rng = np.random.default_rng(2019)
  out = []
  # config 2 (the headline workload) through the host-buffer API: what a caller of solve_batch sees, PCIe copies and host packing included
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=100, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system()); B = 4096
  x0 = np.clip(0.1 * np.random.default_rng(2018).standard_normal((B, 4)), -2, 2)     # (a generator of its own: configs 3-5 keep the draws of the earlier rounds)
  res, med, mn, ms = timed(opt, x0s=x0)
  out.append(_line("2 CARTPOLE HS N=100 through solve_batch (host buffers in and out)", B, res, med, mn, ms))
  # config 3: VANDERPOL shooting 1x50, 8192 per GPU
  hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system()); B = 8192
  x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4)
  res, med, mn, ms = timed(opt, x0s=x0)
  out.append(_line("3 VANDERPOL shooting 1x50 Heun", B, res, med, mn, ms))
  # config 4: CANCERTREATMENT shooting 1x100, 2048 per GPU, parameter sweep
  hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.SHOOTING, max_iter=500, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system()); B = 2048
  params = np.stack([rng.uniform(0.1, 0.5, B), rng.uniform(1, 5, B), rng.uniform(0.2, 0.8, B)], axis=1)
  res, med, mn, ms = timed(opt, x0s=rng.uniform(0.5, 0.99, (B, 1)), params=params)
  out.append(_line("4 CANCERTREATMENT shooting 1x100 Heun sweep", B, res, med, mn, ms))
  # config 5: CARTPOLE + NODE HS N=100, 128 per GPU (1024 over 8) and 1024 on one GPU
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
  for B in (128, 1024):
    x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
    res, med, mn, ms = timed(opt, reps=10, x0s=x0, params=opt.system.device_params())
    out.append(_line("5 CARTPOLE+NODE(64,64) HS N=100", B, res, med, mn, ms))
  # README:83 literal: CARTPOLE trapezoidal N=100
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, intervals=100, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system()); B = 4096
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  res, med, mn, ms = timed(opt, x0s=x0)
  out.append(_line("README:83 CARTPOLE trapezoidal N=100", B, res, med, mn, ms))
  return out


if __name__ == "__main__":
  for o in measure(): print(json.dumps(o))
