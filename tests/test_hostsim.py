"""CPU tests of the solver core (myriad_amd/csrc/hs_solver.h) through its TEST-ONLY host build (tests/hostsim):
one Newton/SQP step against a dense KKT solve built from the oracle, full solves against the golden trajectories.
The same templates are what the HIP kernel instantiates per lane; the GPU-side parity tests are in test_gpu_solve.py."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import myriad_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def sim():
  subprocess.run(["bash", os.path.join(HERE, "hostsim", "build.sh")], check=True)
  lib = C.CDLL(os.path.join(HERE, "hostsim", "libhostsim.so"))
  dp = C.c_void_p
  lib.hostsim_step.argtypes = [C.c_int, C.c_int, C.c_double] + [dp] * 6 + [C.c_double] + [dp] * 4
  lib.hostsim_solve.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, C.c_double,
                                C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp]
  lib.hostsim_rollout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp]
  lib.hostsim_rollout.restype = C.c_double
  return lib


A = lambda a: a.ctypes.data
SID = {"CARTPOLE": 0, "VANDERPOL": 1, "CANCERTREATMENT": 2, "SIMPLECASE": 3}


@pytest.mark.parametrize("name,N,seed", [("CARTPOLE", 6, 1), ("CARTPOLE", 12, 2), ("SIMPLECASE", 4, 5)])
def test_riccati_step_equals_dense_kkt_solve(sim, name, N, seed):
  """dz from the stage-wise recursion (terminal multipliers, mu-linear right-hand side, adjoint lambda) equals the
  solution of the full dense primal-dual KKT system assembled from the oracle's f, c, J and torch's exact Hessian."""
  s = O.SYSTEMS[name](); tr = O.hermite_simpson(s, N); cb = O.Callbacks(tr)
  rng = np.random.default_rng(seed)
  n, ns, K = tr.guess.size, s.ns, 2 * N + 1
  m = 2 * N * ns
  lb, ub = tr.bounds[:, 0].copy(), tr.bounds[:, 1].copy()
  free = lb < ub
  z = tr.guess + 0.05 * rng.standard_normal(n); z[~free] = lb[~free]
  hasL, hasU = free & np.isfinite(lb), free & np.isfinite(ub)
  zL = np.where(hasL, rng.uniform(0.5, 1.5, n), 0.0); zU = np.where(hasU, rng.uniform(0.5, 1.5, n), 0.0)
  nuT = 0.3 * rng.standard_normal(ns); mu = 0.1
  lam, dz, nu, info = np.zeros(m), np.zeros(n), np.zeros(ns), np.zeros(16)
  zz = z.copy()
  assert sim.hostsim_step(SID[name], N, s.T, A(zz), A(lb), A(ub), A(zL), A(zU), A(nuT), mu, A(lam), A(dz), A(nu), A(info)) == 0
  f, g, c, J = cb.fun(z), cb.grad(z), cb.cons(z), cb.jac(z)
  assert info[0] == pytest.approx(f, rel=1e-13, abs=1e-15) and info[1] == pytest.approx(np.abs(c).sum(), rel=1e-13)
  # adjoint multipliers: x-row stationarity holds by construction
  r = g - zL + zU + J.T @ lam
  tp = ~free[(K - 1) * ns:K * ns]
  r[(K - 1) * ns:K * ns] += np.where(tp, nuT, 0)
  assert np.abs(r[ns:K * ns]).max() < 1e-11
  assert info[3] == pytest.approx(np.abs(r[K * ns:]).max(), rel=1e-9)
  if info[8] != 0:
    pytest.skip("inertia correction active at this point")
  W = torch.func.hessian(lambda zt, lt: tr.objective(zt) + (lt * tr.constraints(zt)).sum())(torch.as_tensor(z), torch.as_tensor(lam)).numpy()
  sl, su = np.where(hasL, z - lb, 1.0), np.where(hasU, ub - z, 1.0)
  Sig = np.where(hasL, zL / sl, 0) + np.where(hasU, zU / su, 0)
  gb = g - np.where(hasL, mu / sl, 0) + np.where(hasU, mu / su, 0)
  fi = np.where(free)[0]
  H = W[np.ix_(fi, fi)] + np.diag(Sig[fi])
  KKT = np.block([[H, J[:, fi].T], [J[:, fi], np.zeros((m, m))]])
  sol = np.linalg.solve(KKT, -np.concatenate([gb[fi], c]))
  dzd = np.zeros(n); dzd[fi] = sol[:fi.size]
  assert np.abs(dz - dzd).max() <= 1e-8 * max(1.0, np.abs(dzd).max())
  assert info[7] == pytest.approx(gb @ dzd, rel=1e-7)


def _solve(sim, name, N, T, z0, lb, ub, max_iter=1000):
  B, n = z0.shape
  m = 2 * N * O.SYSTEMS[name]().ns
  z = z0.copy(); lam = np.zeros((B, m)); cost = np.zeros(B)
  st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
  lb = np.ascontiguousarray(lb); ub = np.ascontiguousarray(ub)
  sim.hostsim_solve(SID[name], N, T, B, A(z), A(lb), A(ub), None, 0, max_iter, 1e-8, 1e-6, 1e-7, 0.1, A(lam), A(cost), A(st), A(it), A(kkt))
  return z, lam, cost, st, it, kkt


def test_solver_core_matches_golden_trajectories(sim, golden_dir):
  files = sorted(glob.glob(os.path.join(golden_dir, "solve_hs_cartpole_N*.npz")))
  assert files
  same_total = n_total = 0
  for path in files:
    d = np.load(path)
    N = int(d["N"])
    if N > 25:
      continue
    z, lam, cost, st, it, kkt = _solve(sim, "CARTPOLE", N, 2.0, d["z0"], d["lb"], d["ub"])
    assert (st == 0).all()
    same = np.isclose(cost, d["cost"], rtol=1e-9)
    assert (cost[~same] < d["cost"][~same]).all()           # other basin only if better (non-convex swing-up)
    assert np.abs(z[same] - d["z"][same]).max() < 1e-6
    same_total += int(same.sum()); n_total += same.size
  assert same_total >= n_total - 1


def test_solver_core_other_systems_converge_and_agree_with_slsqp(sim):
  """HS transcription of the other three hot-path systems (free terminal state / infinite bounds / log dynamics)."""
  for name, N in [("VANDERPOL", 20), ("CANCERTREATMENT", 20), ("SIMPLECASE", 10)]:
    s = O.SYSTEMS[name](); tr = O.hermite_simpson(s, N)
    z, lam, cost, st, it, kkt = _solve(sim, name, N, s.T, tr.guess[None], tr.bounds[None, :, 0], tr.bounds[None, :, 1])
    assert st[0] == 0, (name, st, kkt)
    cb = O.Callbacks(tr)
    assert np.abs(cb.cons(z[0])).max() <= 1e-8
    r = O.solve(tr, "SLSQP", extra_options={"ftol": 1e-13}, cb=cb)
    assert cost[0] <= r["cost"] + 1e-7 * max(1.0, abs(r["cost"])), (name, cost[0], r["cost"])
    assert cost[0] == pytest.approx(r["cost"], rel=1e-5)


@pytest.mark.parametrize("method,mid", [("EULER", 0), ("HEUN", 1), ("MIDPOINT", 2), ("RK4", 3)])
def test_rollout_core_matches_oracle(sim, method, mid):
  """csrc/rollout.h (what myr_rollout runs per lane) vs the oracle's get_state_trajectory_and_cost (utils.py:258-298)."""
  rng = np.random.default_rng(3)
  for name in ("CARTPOLE", "VANDERPOL", "CANCERTREATMENT", "SIMPLECASE"):
    s = O.SYSTEMS[name]()
    S = 17
    rows = (2 if method == "RK4" else 1) * S + 1
    us = 0.3 * rng.standard_normal((rows, 1))
    if name == "CANCERTREATMENT":
      us = np.abs(us)
    xs = np.zeros((S + 1, s.ns))
    c = sim.hostsim_rollout(SID[name], mid, S, s.T / S / 4, rows, A(np.ascontiguousarray(s.x_0)), A(us), None, A(xs))
    class Sh(type(s)):
      pass
    s2 = O.SYSTEMS[name](); s2.T = s.T / 4
    oxs, oc = O.get_state_trajectory_and_cost(s2, S, method, s.x_0, us)
    np.testing.assert_allclose(xs, oxs, rtol=1e-12, atol=1e-13)
    assert c == pytest.approx(oc, rel=1e-12, abs=1e-14)


def _os_lib(sim):
  dp = C.c_void_p
  sim.hostsim_solve_trap.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
  sim.hostsim_solve_shoot.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
  return sim


@pytest.mark.parametrize("name,N", [("CARTPOLE", 25), ("VANDERPOL", 30), ("CANCERTREATMENT", 30), ("SIMPLECASE", 20)])
def test_trapezoid_core_matches_oracle_slsqp(sim, name, N):
  """csrc/os_solver.h TrapCore (README.md:83's transcription) vs the oracle's SLSQP path."""
  sim = _os_lib(sim)
  s = O.SYSTEMS[name](); tr = O.trapezoidal(s, N); cb = O.Callbacks(tr)
  z = tr.guess[None].copy(); lb = np.ascontiguousarray(tr.bounds[None, :, 0]); ub = np.ascontiguousarray(tr.bounds[None, :, 1])
  lam = np.zeros((1, N * s.ns)); cost = np.zeros(1); st = np.zeros(1, np.int32); it = np.zeros(1, np.int32); kkt = np.zeros((1, 3))
  sim.hostsim_solve_trap(SID[name], N, s.T, 1, A(z), A(lb), A(ub), None, 0, 500, A(lam), A(cost), A(st), A(it), A(kkt))
  assert st[0] == 0
  assert np.abs(cb.cons(z[0])).max() <= 1e-8 and cb.fun(z[0]) == pytest.approx(cost[0], rel=1e-12)
  r = O.solve(tr, "SLSQP", extra_options={"ftol": 1e-13}, cb=cb)
  assert cost[0] == pytest.approx(r["cost"], rel=1e-7) and np.abs(z[0] - r["xs_and_us"]).max() < 1e-4
  rr = cb.grad(z[0]) + cb.jac(z[0]).T @ lam[0]
  inact = (lb[0] < ub[0]) & (z[0] - lb[0] > 1e-3) & (ub[0] - z[0] > 1e-3)
  assert np.abs(rr[inact]).max() < 1e-5


@pytest.mark.parametrize("name,I,cpi,method", [("SIMPLECASE", 10, 100, "HEUN"), ("VANDERPOL", 1, 50, "HEUN"),
                                               ("CANCERTREATMENT", 1, 100, "HEUN"), ("CARTPOLE", 10, 5, "HEUN"),
                                               ("SIMPLECASE", 4, 10, "EULER")])
def test_shooting_core_matches_oracle_slsqp(sim, name, I, cpi, method):
  """csrc/os_solver.h ShootCore (lifted step-level Riccati) on the BASELINE shooting shapes (configs 1, 3, 4) vs the
  oracle's SLSQP path; the survey's App. C costs are reproduced."""
  sim = _os_lib(sim)
  s = O.SYSTEMS[name](); tr = O.shooting(s, I, cpi, method); cb = O.Callbacks(tr)
  z = tr.guess[None].copy(); lb = np.ascontiguousarray(tr.bounds[None, :, 0]); ub = np.ascontiguousarray(tr.bounds[None, :, 1])
  lam = np.zeros((1, I * s.ns)); cost = np.zeros(1); st = np.zeros(1, np.int32); it = np.zeros(1, np.int32); kkt = np.zeros((1, 3))
  sim.hostsim_solve_shoot(SID[name], I, cpi, {"EULER": 0, "HEUN": 1}[method], s.T, 1, A(z), A(lb), A(ub), None, 0, 500,
                          A(lam), A(cost), A(st), A(it), A(kkt))
  assert st[0] == 0
  assert np.abs(cb.cons(z[0])).max() <= 1e-8 and cb.fun(z[0]) == pytest.approx(cost[0], rel=1e-11)
  r = O.solve(tr, "SLSQP", max_iter=500, extra_options={"ftol": 1e-13}, cb=cb)
  assert cost[0] == pytest.approx(r["cost"], rel=1e-6)
  survey = {("SIMPLECASE", 10): -1.3543305221, ("VANDERPOL", 1): 2.8731963348, ("CANCERTREATMENT", 1): 20.5735535185}
  if (name, I) in survey and method == "HEUN":
    assert cost[0] == pytest.approx(survey[(name, I)], rel=1e-4) and cost[0] <= survey[(name, I)] + 1e-9   # SLSQP at default ftol=1e-6 stops slightly short
