"""GPU parity tests of the batched SQP (myr_solve through the C-ABI) against the oracle's golden solutions,
the oracle's callbacks, and size-independent optimality properties at the BASELINE size."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(N, B=64):
  from myriad_amd import _lib
  return _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, 2.0, max_batch=B)


def test_solve_matches_golden_trajectories(golden_dir):
  """z* from the oracle's tightened SLSQP path (tests/golden/make_solve_golden.py): cost to 1e-9 relative,
  trajectory to 1e-6 absolute (SURVEY.md 8(c) tolerances), feasibility re-checked by the oracle callbacks."""
  from oracle import myriad_oracle as O
  files = sorted(glob.glob(os.path.join(golden_dir, "solve_hs_cartpole_N*.npz")))
  assert files
  n_same = n_all = 0
  for path in files:
    d = np.load(path)
    N = int(d["N"])
    eng = _engine(N)
    res = eng.solve(d["z0"], d["lb"], d["ub"])
    assert (res["status"] == 0).all(), (path, res["status"], res["kkt"])
    # Swing-up is non-convex: parity is per basin (SURVEY.md 7 "hard parts").  Where both solvers land in the same
    # local optimum the trajectory must match; where they do not, ours must be the better one (and still pass the
    # oracle-side KKT checks below).
    # The solver stops at max|c| <= tol_feas = 1e-8, and f moves by lam . c to first order in that residual (|lam| ~ 1e2
    # here, so up to ~1e-8 relative): the comparison uses the Lagrangian value f + lam . c, which is second-order in it.
    corr = np.empty_like(res["cost"])
    for b in range(d["z"].shape[0]):
      s = O.CartPole(); s.x_0 = d["x0"][b]
      corr[b] = res["cost"][b] + res["lam"][b] @ O.Callbacks(O.hermite_simpson(s, N)).cons(res["z"][b])
    same = np.isclose(corr, d["cost"], rtol=1e-9, atol=0)
    n_same += int(same.sum()); n_all += same.size
    assert (res["cost"][~same] < d["cost"][~same]).all(), (path, res["cost"], d["cost"])
    assert np.abs(res["z"][same] - d["z"][same]).max() < 1e-6, (path, np.abs(res["z"] - d["z"]).max(axis=1))
    # multipliers against the polished fixture (|KKT| <= 1e-12 there), sign convention of the reference's mult_g
    # (nlp_solvers/__init__.py:82-86: L = f + lam . c); the solver stops at a scaled stationarity of 1e-6
    assert "lam" in d.files and float(d["kkt"].max()) <= 1e-12, path
    lam_err = np.abs(res["lam"][same] - d["lam"][same]).max(axis=1) / np.maximum(1.0, np.abs(d["lam"][same]).max(axis=1))
    assert lam_err.max() < 1e-5, (path, lam_err)
    for b in range(d["z"].shape[0]):
      s = O.CartPole(); s.x_0 = d["x0"][b]
      cb = O.Callbacks(O.hermite_simpson(s, N))
      assert np.abs(cb.cons(res["z"][b])).max() <= 1e-8
      assert cb.fun(res["z"][b]) == pytest.approx(res["cost"][b], rel=1e-12)
      # stationarity with the returned multipliers on every free variable (pinned rows carry their own multiplier)
      free = d["lb"][b] < d["ub"][b]
      r = cb.grad(res["z"][b]) + cb.jac(res["z"][b]).T @ res["lam"][b]
      inactive = free & (res["z"][b] - d["lb"][b] > 1e-3) & (d["ub"][b] - res["z"][b] > 1e-3)
      assert np.abs(r[inactive]).max() < 1e-5
    eng.close()
  assert n_same >= n_all - 1, (n_same, n_all)


def test_solve_random_batch_properties():
  """Ragged batch of random x0 at N=25: converged, pinned variables exact, bounds respected, feasibility and
  objective confirmed through the independent eval kernel."""
  from oracle import myriad_oracle as O
  N, B = 25, 77
  s = O.CartPole()
  x0 = O.random_x0(s, B, seed=5)
  tr = O.hermite_simpson(s, N)
  K = 2 * N + 1
  z0 = np.stack([np.concatenate([np.linspace(x0[b], s.x_T, K).ravel(), np.zeros(K)]) for b in range(B)])
  lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
  lb[:, :4] = x0; ub[:, :4] = x0
  eng = _engine(N, B)
  res = eng.solve(z0, lb, ub)
  assert (res["status"] == 0).all()
  z = res["z"]
  assert np.array_equal(z[:, :4], x0) and np.array_equal(z[:, (K - 1) * 4:K * 4], np.tile(s.x_T, (B, 1)))
  assert (z >= lb).all() and (z <= ub).all()
  ev = eng.eval(z)
  assert np.abs(ev["c"]).max() <= 1e-8
  np.testing.assert_allclose(ev["f"], res["cost"], rtol=1e-12)
  assert res["iters"].max() < 200
  # solving again from the solution converges immediately to the same point (idempotence)
  res2 = eng.solve(z, lb, ub)
  assert (res2["status"] == 0).all()
  np.testing.assert_allclose(res2["cost"], res["cost"], rtol=1e-7)


def test_park_iter_option_of_the_abi_changes_the_schedule_not_the_result(monkeypatch):
  """myr_solve_opts.park_iter through the C-ABI (not the environment): a batch too small for the library to choose the two-phase launch, forced into
  it with park_iter = 3 and 9, against whole solves (park_iter = -1): the same bits, and a zero-initialised field (the library decides) as well."""
  from oracle import myriad_oracle as O
  monkeypatch.setenv("MYRIAD_FUSED_WAVES", "1"); monkeypatch.delenv("MYRIAD_PARK_ITER", raising=False)
  N, B = 25, 77
  s = O.CartPole()
  x0 = O.random_x0(s, B, seed=11)
  tr = O.hermite_simpson(s, N)
  K = 2 * N + 1
  z0 = np.stack([np.concatenate([np.linspace(x0[b], s.x_T, K).ravel(), np.zeros(K)]) for b in range(B)])
  lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
  lb[:, :4] = x0; ub[:, :4] = x0
  eng = _engine(N, B)
  out = {}
  for k in (-1, 0, 3, 9):
    o = eng.default_opts(); o.restoration = 0; o.park_iter = k
    r = eng.solve(z0, lb, ub, opts=o)
    assert (r["status"] == 0).all()
    out[k] = b"".join(np.ascontiguousarray(r[q]).tobytes() for q in ("z", "lam", "cost", "kkt", "status", "iters"))
  assert out[0] == out[-1] and out[3] == out[-1] and out[9] == out[-1]


def test_solve_nonconvergence_is_reported_not_raised():
  """max_iter too small: status = MAXITER per instance, no exception (reference: solution['success'] is only printed,
  nlp_solvers/__init__.py:64)."""
  from oracle import myriad_oracle as O
  N = 10
  s = O.CartPole(); tr = O.hermite_simpson(s, N)
  eng = _engine(N)
  o = eng.default_opts(); o.max_iter = 2; o.restoration = 0          # ONE attempt from the caller's point (the reference's call)
  res = eng.solve(tr.guess[None], tr.bounds[None, :, 0], tr.bounds[None, :, 1], opts=o)
  assert res["status"][0] == 1 and res["iters"][0] == 2 and res["attempts"][0] == 1
  o.restoration = 2                                                   # second starts only (bit 2): three excitation guesses follow
  res = eng.solve(tr.guess[None], tr.bounds[None, :, 0], tr.bounds[None, :, 1], opts=o)
  assert res["status"][0] == 1 and res["iters"][0] == 8 and res["attempts"][0] == 4 and res["restored"][0] == 0
  assert eng.solve(np.zeros((0, eng.n)), np.zeros((0, eng.n)), np.zeros((0, eng.n)))["z"].shape == (0, eng.n)


def test_solve_full_size_baseline_config():
  """BASELINE config 2: N=100, B=4096 random x0.  All instances converge; feasibility via the eval kernel;
  golden N=100 default-x0 cost (SciPy SLSQP 87.96437982986194 / trust-constr 87.96437633178395) within their
  own stopping tolerance of ours."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from bench import build_workload
  N, B = 100, 4096
  x0, z0, lb, ub, T = build_workload(B, N, 2019)
  x0[0] = 0.0; z0[0, :] = 0.0
  K = 2 * N + 1
  z0[0, :K * 4] = np.linspace(np.zeros(4), np.array([1., np.pi, 0., 0.]), K).ravel()
  lb[0, :4] = 0.0; ub[0, :4] = 0.0
  eng = _engine(N, B)
  res = eng.solve(z0, lb, ub)
  assert (res["status"] == 0).mean() >= 0.999, np.bincount(res["status"])
  ev = eng.eval(res["z"], want=("c", "f"))
  ok = res["status"] == 0
  assert np.abs(ev["c"][ok]).max() <= 1e-8
  assert (res["z"] >= lb).all() and (res["z"] <= ub).all()
  assert res["cost"][0] == pytest.approx(87.964376, rel=1e-7)
  assert abs(res["cost"][0] - 87.96437982986194) < 1e-5      # SLSQP at default ftol=1e-6
  assert abs(res["cost"][0] - 87.96437633178395) < 1e-6      # trust-constr


def test_wave_and_lane_kernels_agree(monkeypatch):
  """The two mappings of the same algorithm (one trajectory per wavefront / per lane) must produce the same iterates
  up to round-off: same iteration counts on almost every instance, same costs."""
  from oracle import myriad_oracle as O
  N, B = 25, 40
  s = O.CartPole()
  x0 = O.random_x0(s, B, seed=11)
  tr = O.hermite_simpson(s, N)
  K = 2 * N + 1
  z0 = np.stack([np.concatenate([np.linspace(x0[b], s.x_T, K).ravel(), np.zeros(K)]) for b in range(B)])
  lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
  lb[:, :4] = x0; ub[:, :4] = x0
  out = {}
  for mode in ("wave", "lane"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    eng = _engine(N, B)
    out[mode] = eng.solve(z0, lb, ub)
    eng.close()
  assert (out["wave"]["status"] == 0).all() and (out["lane"]["status"] == 0).all()
  np.testing.assert_allclose(out["wave"]["cost"], out["lane"]["cost"], rtol=1e-9)
  assert (out["wave"]["iters"] == out["lane"]["iters"]).mean() >= 0.8
  assert np.abs(out["wave"]["z"] - out["lane"]["z"]).max() < 1e-6
  assert np.abs(out["wave"]["lam"] - out["lane"]["lam"]).max() < 1e-4 * max(1.0, np.abs(out["lane"]["lam"]).max())


def test_fused_kernel_wavefront_forms_and_round2_kernel_agree(monkeypatch):
  """hs_solver_fused.h with one and with two wavefronts per trajectory (the small-batch form: blocks of 64 stages dealt to the
  two wavefronts, scan carries and neighbour records through LDS, the second inertia candidate swept speculatively by wavefront 1)
  and round 2's kernel (MYRIAD_SOLVE_MODE=wave1) run one algorithm: same optima, same iteration counts up to the order of the sums."""
  from bench import build_workload
  from myriad_amd import _lib
  B, N = 96, 100
  x0, z0, lb, ub, T = build_workload(B, N, 11)
  out = {}
  for tag, env in (("w1", {"MYRIAD_FUSED_WAVES": "1"}), ("w2", {"MYRIAD_FUSED_WAVES": "2"}), ("r2", {"MYRIAD_SOLVE_MODE": "wave1"})):
    for k in ("MYRIAD_FUSED_WAVES", "MYRIAD_SOLVE_MODE"):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
    out[tag] = eng.solve(z0, lb, ub)
    eng.close()
  for tag in ("w2", "r2"):
    a, b = out["w1"], out[tag]
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-9)
    if tag == "w2":     # W = 1 and W = 2 take bit-identical steps; only the merit sums differ in their last bits: the same iterations
      assert np.array_equal(a["iters"], b["iters"]), (tag, np.bincount(np.abs(a["iters"] - b["iters"])))
    else:
      assert (a["iters"] == b["iters"]).mean() >= 0.9, (tag, np.bincount(np.abs(a["iters"] - b["iters"])))
    same = a["iters"] == b["iters"]
    assert np.abs(a["z"][same] - b["z"][same]).max() < 1e-6
    assert np.abs(a["lam"][same] - b["lam"][same]).max() < 1e-4 * max(1.0, np.abs(a["lam"]).max())


def test_long_horizon_falls_back_to_the_lane_solver():
  """N = 600 intervals: the wavefront solver's LDS working set (~200 KB) exceeds a CU, so the lane-per-trajectory form
  runs instead of an error; same optimum as a coarser grid to discretisation accuracy."""
  from myriad_amd.config import Config, HParams, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  cfg = Config(verbose=False, plot=False)
  sols = {}
  for N in (60, 600):
    hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=N)
    sols[N] = get_optimizer(hp, cfg, hp.system()).solve()
  assert sols[600]['x'].shape == (1201, 2)
  assert abs(sols[600]['cost'] - sols[60]['cost']) < 1e-4 * max(1.0, abs(sols[60]['cost']))


def _shoot_opt(system, intervals, cpi, method):
  from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = HParams(system=getattr(SystemType, system), optimizer=OptimizerType.SHOOTING, intervals=intervals, controls_per_interval=cpi,
               integration_method=getattr(IntegrationMethod, method), nlpsolver=NLPSolverType.SQP)
  return get_optimizer(hp, Config(verbose=False, plot=False), hp.system())


@pytest.mark.parametrize("system,intervals,cpi,method", [
    ("VANDERPOL", 1, 50, "HEUN"),        # config 3: single shooting, Riccati recursion on the matrix cores
    ("VANDERPOL", 5, 10, "HEUN"),        # multiple shooting: node states with own terms inside the sweep
    ("VANDERPOL", 1, 40, "EULER"),
    ("VANDERPOL", 2, 20, "MIDPOINT"),
    ("VANDERPOL", 1, 20, "RK4"),         # two control rows per step: the one-lane recursion (os_riccati_stage)
    ("CARTPOLE", 4, 10, "HEUN"),         # NS = 4
    ("CARTPOLE", 20, 10, "HEUN"),        # 200 steps: 105 KB of LDS per trajectory, one workgroup per CU
    ("CANCERTREATMENT", 1, 100, "HEUN"),
    ("BEARPOPULATIONS", 1, 40, "HEUN"),  # two controls: the one-lane recursion
])
def test_shooting_wave_and_lane_kernels_agree(monkeypatch, system, intervals, cpi, method):
  """shoot_solver_wave.h (one trajectory per wavefront, iterate in LDS) and ShootCore in the lane kernel are two mappings
  of one algorithm: same status, same optimum, and the same iteration count on almost every instance (sums are associated
  differently, so a few instances may take another number of iterations)."""
  rng = np.random.default_rng(7)
  B = 96
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0")     # kernels are compared: no host-side rescue of a failed device solve
  x_0 = np.array(_shoot_opt(system, intervals, cpi, method).system.x_0, float)
  if system == "VANDERPOL":
    x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4)
  else:
    x0 = x_0 * (1 + 0.05 * rng.standard_normal((B, len(x_0)))) + (0.02 * rng.standard_normal((B, len(x_0))) if system == "CARTPOLE" else 0.0)
  out = {}
  for mode in ("wave", "lane"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    out[mode] = _shoot_opt(system, intervals, cpi, method).solve_batch(x0s=x0)
  w, l = out["wave"], out["lane"]
  # same outcome, except where the wavefront kernel's second start (another initial barrier parameter) rescued a solve
  # that runs into the iteration limit in both kernels (VANDERPOL, 40 Euler steps: 4 of 96)
  assert ((w["status"] == l["status"]) | (w["status"] == 0)).all() and (w["status"] == 0).mean() >= 0.95
  ok = (w["status"] == 0) & (l["status"] == 0)
  np.testing.assert_allclose(w["cost"][ok], l["cost"][ok], rtol=1e-7)
  assert (w["iters"][ok] == l["iters"][ok]).mean() >= 0.8, np.bincount(np.abs(w["iters"][ok] - l["iters"][ok]))
  # both stop at the KKT tolerances (1e-6 / 1e-7), one of them possibly an iteration later: the points agree to that order
  assert np.abs(w["xs_and_us"][ok] - l["xs_and_us"][ok]).max() < 1e-3
  same = w["iters"] == l["iters"]
  assert np.abs(w["xs_and_us"][ok & same] - l["xs_and_us"][ok & same]).max() < 1e-5
  assert np.abs(w["lambda"][ok] - l["lambda"][ok]).max() < 1e-3 * max(1.0, np.abs(l["lambda"][ok]).max())


def test_shooting_wave_kernel_restarts_a_stalled_solve(monkeypatch):
  """Instance 3985 of the config-3 batch stalls in the line search after 58 iterations from mu = 0.1 with the wavefront
  kernel's summation order (a non-descent direction at a point with 38 regularised pivots); the kernel starts it again from
  the caller's point with mu x 3 and reports the iterations of both attempts."""
  rng = np.random.default_rng(2019)
  x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((8192, 2)), -4, 4)[3985:3986]
  monkeypatch.setenv("MYRIAD_SOLVE_MODE", "wave")
  w = _shoot_opt("VANDERPOL", 1, 50, "HEUN").solve_batch(x0s=x0)
  monkeypatch.setenv("MYRIAD_SOLVE_MODE", "lane")
  l = _shoot_opt("VANDERPOL", 1, 50, "HEUN").solve_batch(x0s=x0)
  assert w["status"][0] == 0 and l["status"][0] == 0
  assert w["cost"][0] <= l["cost"][0] * (1 + 1e-6)
  monkeypatch.setenv("MYRIAD_SOLVE_MODE", "wave")
  opt = _shoot_opt("VANDERPOL", 1, 50, "HEUN")
  assert np.abs(opt.constraints(w["xs_and_us"][0])).max() <= 1e-8
  # myr_solve_opts.restarts is a runtime option: 0 = one attempt bounded by max_iter (what the lane kernels do), and
  # `iters` <= max_iter then holds; the default (-1 -> 2 for this kernel) is what rescued the solve above
  eng = opt.engine
  z0, lb, ub = opt.batch_inputs(x0, opt.system.device_params())
  o = eng.default_opts(); o.max_iter = 150
  assert o.restarts == -1
  o.restarts = 0
  r0 = eng.solve(z0, lb, ub, params=opt.system.device_params(), opts=o)
  assert r0["iters"][0] <= 150                                    # one attempt, bounded by max_iter (converged or not)
  o.restarts = 2
  r2 = eng.solve(z0, lb, ub, params=opt.system.device_params(), opts=o)
  assert r2["status"][0] == 0                                     # first attempt cut at 100, then a second start


@pytest.mark.parametrize("system,transcription,kw", [
    ("CARTPOLE", "hs", dict(intervals=20)),                       # x_T given: linspace(x0, x_T) per instance
    ("VANDERPOL", "hs", dict(intervals=16)),                      # no x_T: ones * 0.1 whatever x0 is (g1 = 0)
    ("CARTPOLE", "trap", dict(intervals=30)),
    ("CARTPOLE", "shoot", dict(intervals=10, controls_per_interval=2)),
])
def test_solve_x0_matches_solve(system, transcription, kw, monkeypatch):
  """myr_solve_x0 (start states + guess rule + bound templates, expanded on the device) against myr_solve on the arrays
  batch_inputs builds on the host: the SAME inputs reach the solver (the expansion rounds like numpy), so every output is bit
  for bit the same -- host buffers, device buffers, and through solve_batch with MYRIAD_SOLVE_X0 on / off."""
  import ctypes as C
  import torch
  from myriad_amd import _lib
  from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  if transcription == "shoot":
    hp = HParams(system=SystemType[system], optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, nlpsolver=NLPSolverType.SQP, **kw)
  else:
    hp = HParams(system=SystemType[system], optimizer=OptimizerType.COLLOCATION, nlpsolver=NLPSolverType.SQP,
                 quadrature_rule=QuadratureRule.HERMITE_SIMPSON if transcription == "hs" else QuadratureRule.TRAPEZOIDAL, **kw)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  rule = opt.x0_rule()
  assert rule is not None
  B = 96
  rng = np.random.default_rng(5)
  x0s = opt.system.x_0[None] + 0.1 * rng.standard_normal((B, opt.system.x_0.shape[0]))
  p = opt.system.device_params()
  eng = opt.engine
  z0, lb, ub = opt.batch_inputs(x0s, p)
  ref = eng.solve(z0, lb, ub, params=p)
  assert (ref["status"] == 0).mean() > 0.9
  got = eng.solve_x0(x0s, rule[0], rule[1], opt.bounds[:, 0], opt.bounds[:, 1], params=p)
  for k in ("z", "lam", "cost", "status", "iters", "kkt"):
    assert np.array_equal(ref[k], got[k]), k
  # device buffers: every pointer a device pointer, z is output only
  dev = torch.device("cuda", 0)
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  dx0, dg0, dg1, dl, du, dp = t(x0s), t(rule[0]), t(rule[1]), t(opt.bounds[:, 0]), t(opt.bounds[:, 1]), t(p)
  dz = torch.full((B, eng.n), float("nan"), dtype=torch.float64, device=dev)
  dlam = torch.empty(B, eng.m, dtype=torch.float64, device=dev); dcost = torch.empty(B, dtype=torch.float64, device=dev)
  dst = torch.empty(B, dtype=torch.int32, device=dev); dit = torch.empty(B, dtype=torch.int32, device=dev)
  dk = torch.empty(B, 3, dtype=torch.float64, device=dev)
  torch.cuda.synchronize()
  o = eng.default_opts()
  a = lambda x: C.c_void_p(x.data_ptr())
  _lib._chk(eng.lib.myr_solve_x0(eng._h, B, a(dx0), a(dg0), a(dg1), a(dl), a(du), a(dp), 0, C.byref(o), a(dz), a(dlam), a(dcost),
                                 a(dst), a(dit), a(dk), _lib.MEM_DEVICE), "myr_solve_x0")
  assert np.array_equal(dz.cpu().numpy(), ref["z"]) and np.array_equal(dst.cpu().numpy(), ref["status"]) and np.array_equal(dlam.cpu().numpy(), ref["lam"])
  # through the reference-shaped API
  r1 = opt.solve_batch(x0s=x0s)
  monkeypatch.setenv("MYRIAD_SOLVE_X0", "0")
  r0 = opt.solve_batch(x0s=x0s)
  for k in ("xs_and_us", "cost", "lambda", "status", "iters"):
    assert np.array_equal(r0[k], r1[k]), k
  assert np.array_equal(r1["xs_and_us"], ref["z"])


def test_solve_x0_rejects_bad_arguments():
  from myriad_amd import _lib
  eng = _engine(10)
  g = np.zeros(eng.n)
  with pytest.raises(ValueError):
    eng.solve_x0(np.zeros((2, eng.ns + 1)), g, g, g, g)
  with pytest.raises(ValueError):
    eng.solve_x0(np.zeros((2, eng.ns)), g[:-1], g, g, g)
  rc = eng.lib.myr_solve_x0(eng._h, 2, None, None, None, None, None, None, 0, None, None, None, None, None, None, None, _lib.MEM_HOST)
  assert rc != 0 and b"myr_solve_x0" in eng.lib.myr_last_error()
