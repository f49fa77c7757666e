"""GPU lane kernels against the host build of the SAME solver source (tests/hostsim, test-only): per instance the two
must take the same path -- same status, same iteration count up to rounding-induced differences, same optimum.  This is
the guard against code-generation problems of the very large lane kernels: a build that lost fixed
lane positions of a trapezoidal batch passed every per-instance parity test but not this one."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = Config(verbose=False, plot=False)
A = lambda a: a.ctypes.data


@pytest.fixture(scope="module")
def sim():
  subprocess.run(["bash", os.path.join(HERE, "hostsim", "build.sh")], check=True)
  lib = C.CDLL(os.path.join(HERE, "hostsim", "libhostsim.so"))
  dp = C.c_void_p
  lib.hostsim_solve_trap.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
  lib.hostsim_solve_shoot.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
  return lib


def _compare(res, st, it, cost, tag):
  assert (res['status'] == st).all(), (tag, np.nonzero(res['status'] != st)[0][:20])
  ok = st == 0
  # fused multiply-adds and transcendental rounding differ between the two builds: a few instances may take one
  # iteration more or less, none a different path
  d = np.abs(res['iters'].astype(int) - it.astype(int))
  assert (d <= 2).mean() >= 0.99 and (d == 0).mean() >= 0.9, (tag, np.bincount(d))
  np.testing.assert_allclose(res['cost'][ok], cost[ok], rtol=1e-6, err_msg=tag)


def test_trapezoid_batch_matches_host_twin(sim):
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, intervals=100, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(5)
  B, N = 512, 100
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  z0, lb, ub = opt.batch_inputs(x0, opt.system.device_params())
  res = opt.solve_batch(x0s=x0)
  z = np.ascontiguousarray(z0.copy()); lb = np.ascontiguousarray(lb); ub = np.ascontiguousarray(ub)
  lam = np.zeros((B, N * 4)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
  os.environ["DWARM"] = "1"                                       # the device enables the warm-started inertia correction for collocation
  try:
    sim.hostsim_solve_trap(0, N, opt.system.T, B, A(z), A(lb), A(ub), None, 0, hp.max_iter, A(lam), A(cost), A(st), A(it), A(kkt))
  finally:
    del os.environ["DWARM"]
  _compare(res, st, it, cost, "trapezoid")


def test_shooting_batch_matches_host_twin(sim):
  hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(2019)
  B = 1024
  x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4)
  z0, lb, ub = opt.batch_inputs(x0, opt.system.device_params())
  res = opt.solve_batch(x0s=x0)
  z = np.ascontiguousarray(z0.copy()); lb = np.ascontiguousarray(lb); ub = np.ascontiguousarray(ub)
  lam = np.zeros((B, 2)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
  sim.hostsim_solve_shoot(1, 1, 50, 1, opt.system.T, B, A(z), A(lb), A(ub), None, 0, hp.max_iter, A(lam), A(cost), A(st), A(it), A(kkt))
  _compare(res, st, it, cost, "shooting")


def test_wavefront_hs_solver_matches_host_twin_of_the_lane_form():
  """The wavefront-per-trajectory Hermite-Simpson kernel (hs_solver_wave.h) and the lane form (hs_solver.h, here its host
  build) are two implementations of one algorithm: on the headline workload they must agree instance by instance."""
  subprocess.run(["bash", os.path.join(HERE, "hostsim", "build.sh")], check=True)
  lib = C.CDLL(os.path.join(HERE, "hostsim", "libhostsim.so"))
  dp = C.c_void_p
  lib.hostsim_solve.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp]
  import sys
  sys.path.insert(0, os.path.dirname(HERE))
  from bench import build_workload
  from myriad_amd import _lib
  B, N = 512, 100
  x0, z0, lb, ub, T = build_workload(B, N, 2019)
  eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
  res = eng.solve(z0, lb, ub)
  eng.close()
  z = z0.copy()
  lam = np.zeros((B, 2 * N * 4)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
  os.environ["DWARM"] = "1"
  try:
    lib.hostsim_solve(0, N, T, B, A(z), A(lb), A(ub), None, 0, 1000, 1e-8, 1e-6, 1e-7, 0.1, A(lam), A(cost), A(st), A(it), A(kkt))
  finally:
    del os.environ["DWARM"]
  assert (res['status'] == 0).all() and (st == 0).all()
  d = np.abs(res['iters'].astype(int) - it.astype(int))
  assert (d <= 2).mean() >= 0.97, np.bincount(d)
  same = np.isclose(res['cost'], cost, rtol=1e-6)
  assert same.mean() >= 0.99, (~same).sum()          # non-convex: a different basin for a handful at most
