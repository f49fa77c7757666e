"""The oracle against numbers computed by THE REFERENCE'S OWN LINES (tests/golden/make_reference_fixtures.py; round 5).

tests/golden/reference_callbacks.npz holds, for 20 systems x {Hermite-Simpson, trapezoidal, shooting x Euler / Heun / midpoint / RK4 + single
shooting}, what /root/reference/myriad's get_optimizer(...) returned in the build container -- guess, bounds, objective(z), constraints(z) at a
seeded z, and utils.get_state_trajectory_and_cost on z's controls -- with jax.numpy forwarded to numpy (tests/golden/refshim; jax itself is not
in the image, so by the parity rules this still is not "the reference run here": DESIGN.md section 6).  The oracle's restatement must reproduce
every one of them to rounding: that is what ties the goldens of the solve tests (oracle output) to the reference's code rather than to a reading
of it.  CPU only; the fixtures travel, /root/reference does not."""
import os

import numpy as np
import pytest

from oracle import myriad_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "reference_callbacks.npz"))
KEYS = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.endswith("/objective")})
REFUSED = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.endswith("/error")})
# the reference cannot form this problem (quirk Q2, trapezoidal.py:71: an x_T list with None entries); the stand-in turns its TypeError into NaN
SHIM_ONLY = {"PREDATORPREY/TRAPEZOIDAL/-/7x1"}


def _close(a, b, what, rtol=1e-13):
  a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
  assert a.shape == b.shape, (what, a.shape, b.shape)
  fin = np.isfinite(b)
  assert np.array_equal(np.isfinite(a), fin), what
  scale = max(1.0, float(np.abs(b[fin]).max())) if fin.any() else 1.0
  assert np.abs(a[fin] - b[fin]).max(initial=0.0) <= rtol * scale * 64, (what, float(np.abs(a[fin] - b[fin]).max(initial=0.0)), scale)
  assert np.array_equal(np.isnan(a), np.isnan(b)), what


def _transcription(key):
  name, tr, method, shape = key.split("/")
  N, cpi = (int(v) for v in shape.split("x"))
  system = O.SYSTEMS[name]()
  if tr == "SHOOTING":
    return system, O.make_transcription(system, "SHOOTING", N, cpi, integration_method=method), N, cpi, method
  return system, O.make_transcription(system, "COLLOCATION", N, 1, quadrature_rule=tr), N, 1, None


def test_fixture_covers_every_system_and_transcription():
  systems = {k.split("/")[0] for k in KEYS}
  assert systems == set(O.SYSTEMS), systems ^ set(O.SYSTEMS)
  assert len(KEYS) >= 20 * 7 - 2
  # what the reference itself refuses: collocation of PREDATORPREY's partially pinned terminal state under Hermite-Simpson (TypeError)
  assert REFUSED == ["PREDATORPREY/HERMITE_SIMPSON/-/6x1"], REFUSED


@pytest.mark.parametrize("key", [k for k in KEYS if k not in SHIM_ONLY])
def test_oracle_reproduces_the_reference_lines(key):
  system, t, N, cpi, method = _transcription(key)
  g = lambda f: FIX[key + "/" + f]
  assert t.x_rows == int(g("x_rows")) and t.u_rows == int(g("u_rows")), key
  _close(t.guess, g("guess"), key + " guess")
  b_ref = g("bounds")
  assert np.array_equal(np.isinf(t.bounds), np.isinf(b_ref)) and np.array_equal(np.sign(t.bounds[np.isinf(b_ref)]), np.sign(b_ref[np.isinf(b_ref)])), key
  _close(np.where(np.isinf(t.bounds), 0.0, t.bounds), np.where(np.isinf(b_ref), 0.0, b_ref), key + " bounds")
  z = g("z")
  cb = O.Callbacks(t)
  f_ref = float(g("objective"))
  if np.isfinite(f_ref):
    assert cb.fun(z) == pytest.approx(f_ref, rel=1e-12, abs=1e-13), key
  else:
    assert not np.isfinite(cb.fun(z)), key
  _close(cb.cons(z), g("constraints"), key + " constraints", rtol=1e-13)
  if key + "/rollout_rk4_cost" in FIX.files:       # utils.py:258-298 on the controls of z
    xs, us = t.unravel(z)
    steps = N * cpi
    xr, cr = O.get_state_trajectory_and_cost(system, steps, "RK4", system.x_0, us)
    _close(xr, g("rollout_rk4_xs"), key + " rollout states", rtol=1e-12)
    c_ref = float(g("rollout_rk4_cost"))
    if np.isfinite(c_ref):
      assert cr == pytest.approx(c_ref, rel=1e-11, abs=1e-12), key


SOLVE_KEYS = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.startswith("solve/") and k.endswith("/cost")})


def test_full_size_headline_solve_of_the_reference_is_the_oracles():
  """tests/golden/reference_solve_full.npz (round 6): BASELINE config 2's problem -- CARTPOLE, Hermite-Simpson, N = 100, from the reference's own start state -- through the
  reference's solve() (nlp_solvers/__init__.py:18-98, SLSQP branch; complex-step derivatives of the reference's callbacks; 44 minutes).  The oracle's restatement of that call was
  run once for the CPU baseline (profiles/r05/cpu_baseline_full.json: 230 s, 110 iterations): the two end at the same cost to eleven digits, the reference's end point is feasible
  for the oracle's constraints to 1e-8 and has the reference's cost under the oracle's objective.  (The oracle's 230-s solve is not repeated here.)"""
  import json
  full = np.load(os.path.join(HERE, "golden", "reference_solve_full.npz"))
  key = "solve/CARTPOLE/COLLOCATION/HERMITE_SIMPSON/100x1"
  z_ref, c_ref = full[key + "/xs_and_us"], float(full[key + "/cost"])
  system = O.SYSTEMS["CARTPOLE"]()
  t = O.make_transcription(system, "COLLOCATION", 100, 1, quadrature_rule="HERMITE_SIMPSON")
  cb = O.Callbacks(t)
  assert z_ref.shape == (1005,)
  assert float(cb.fun(z_ref)) == pytest.approx(c_ref, rel=1e-12)
  assert np.abs(cb.cons(z_ref)).max() < 1e-8
  rec = json.load(open(os.path.join(os.path.dirname(HERE), "profiles", "r05", "cpu_baseline_full.json")))["slsqp_full_solve"]
  assert rec["cost"] == pytest.approx(c_ref, rel=1e-10) and rec["converged"]


def _solve_case(key):
  _, name, optimizer, rule, shape = key.split("/")
  N, cpi = (int(v) for v in shape.split("x"))
  system = O.SYSTEMS[name]()
  if optimizer == "SHOOTING":
    return system, O.make_transcription(system, "SHOOTING", N, cpi, integration_method=rule)
  return system, O.make_transcription(system, "COLLOCATION", N, 1, quadrature_rule=rule)


@pytest.mark.parametrize("key", SOLVE_KEYS)
def test_oracle_solve_path_reproduces_the_reference_solve(key):
  """nlp_solvers/__init__.py:18-98 executed by the generator (SLSQP branch, SciPy's defaults apart from maxiter -- jax.grad / jax.jacrev replaced by
  complex-step derivatives of the reference's own callbacks): the oracle's restatement of that call ends at the same point."""
  system, t = _solve_case(key)
  res = O.solve(t, "SLSQP", max_iter=1000)
  z_ref = FIX[key + "/xs_and_us"]
  # SLSQP stops on its default tolerance (ftol 1e-6; the reference passes only maxiter): collocation runs end within 1e-7 of each other, single
  # shooting over a long horizon amplifies the 1e-16 differences of the derivative values into 2e-5 of the cost
  shooting = "/SHOOTING/" in key
  assert float(res["cost"]) == pytest.approx(float(FIX[key + "/cost"]), rel=1e-4 if shooting else 1e-7, abs=1e-9), key
  if not shooting:
    assert np.abs(np.asarray(res["xs_and_us"]) - z_ref).max() <= 1e-4 * max(1.0, np.abs(z_ref).max()), key
  assert np.abs(O.Callbacks(t).cons(z_ref)).max() < 1e-5, key
  # ... and started at the reference's end point the same SciPy call does not get worse.  (It stays put for four of the five problems; VANDERPOL's
  # single-shooting run -- reference and oracle alike -- stops at cost 23.74 on the default tolerance, a restart from there walks on to 2.92:
  # the ill-conditioning of long-horizon single shooting that BASELINE config 3 is about.)
  again = O.solve(t, "SLSQP", max_iter=1000, guess=z_ref)
  assert float(again["cost"]) <= float(FIX[key + "/cost"]) + 2e-6 * abs(float(FIX[key + "/cost"])) + 1e-9, key
  if "VANDERPOL" not in key:
    assert float(again["cost"]) == pytest.approx(float(FIX[key + "/cost"]), rel=2e-6, abs=1e-9), key


FIX171 = np.load(os.path.join(HERE, "golden", "reference_solve_scipy171.npz"))


@pytest.mark.parametrize("key", SOLVE_KEYS)
def test_reference_solve_under_the_scipy_version_it_pins(key):
  """the reference pins scipy==1.7.0 (requirements.txt); the image's default interpreter has SciPy 1.15, /opt/conda's python3.9
  has 1.7.1.  The generator ran the reference's solve() under both (`--solve-only reference_solve_scipy171` with the conda interpreter): the SLSQP end points
  agree -- to the last printed digit for four of the five problems, to SLSQP's own tolerance for the long-horizon single shooting of VANDERPOL -- so the
  goldens do not hang on the SciPy version, and the oracle's path (SciPy 1.15) ends where the reference's pinned one does."""
  assert str(FIX171["scipy_version"]) == "1.7.1" and key + "/cost" in FIX171.files
  c171, c115 = float(FIX171[key + "/cost"]), float(FIX[key + "/cost"])
  shooting_vdp = "VANDERPOL" in key
  assert c171 == pytest.approx(c115, rel=1e-5 if shooting_vdp else 1e-10, abs=1e-12), key
  if not shooting_vdp:
    assert np.abs(FIX171[key + "/xs_and_us"] - FIX[key + "/xs_and_us"]).max() <= 1e-6 * max(1.0, np.abs(FIX[key + "/xs_and_us"]).max()), key
  system, t = _solve_case(key)
  res = O.solve(t, "SLSQP", max_iter=1000)
  assert float(res["cost"]) == pytest.approx(c171, rel=1e-4 if "/SHOOTING/" in key else 1e-7, abs=1e-9), key


FBSM_KEYS = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.startswith("fbsm/") and k.endswith("/sweeps")})


def test_fbsm_fixture_covers_the_indirect_systems():
  assert len(FBSM_KEYS) == 13 and "fbsm/INVASIVEPLANT" in FBSM_KEYS and "fbsm/BEARPOPULATIONS" in FBSM_KEYS, FBSM_KEYS


@pytest.mark.parametrize("key", FBSM_KEYS)
def test_oracle_fbsm_reproduces_the_reference_sweeps(key):
  """trajectory_optimizers/forward_backward_sweep.py:88-116 executed by the generator (fbsm_intervals = 200, at most 40 sweeps through the reference's own
  stopping rule): the oracle's restatement stops after the same number of sweeps with the same state, control and adjoint trajectories -- the discrete
  INVASIVEPLANT and the two-control BEARPOPULATIONS included.  (The batched device kernel `myr_fbsm` is tested against this restatement, tests/test_gpu_fbsm.py.)"""
  name = key.split("/")[1]
  system = O.InvasivePlant() if name == "INVASIVEPLANT" else O.SYSTEMS[name]()
  r = O.fbsm(system, N=200, max_sweeps=40)
  assert r["sweeps"] == int(FIX[key + "/sweeps"]), (key, r["sweeps"], int(FIX[key + "/sweeps"]))
  for f in ("x", "u", "adj"):
    _close(r[f], FIX[key + "/" + f], key + " " + f, rtol=1e-12)


EXGD_KEYS = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.startswith("exgd/") and k.endswith("/fun")})


@pytest.mark.parametrize("key", EXGD_KEYS)
def test_oracle_extragradient_step_reproduces_the_reference_iteration(key):
  """nlp_solvers/extra_gradient.py:10-84 executed by the generator for 25 steps (eta_x 1e-2, eta_v 1e-3, the step decay at iteration 0 included; jax.grad of
  the reference's Lagrangian by complex-step derivatives): the oracle's `Lagrangian.step` (what the device kernels `myr_vjp` / `myr_exgd` are tested against)
  walks the same 25 steps."""
  _, name, optimizer, rule, shape = key.split("/")
  N, cpi = (int(v) for v in shape.split("x"))
  system = O.SYSTEMS[name]()
  t = O.make_transcription(system, "SHOOTING", N, cpi, integration_method=rule) if optimizer == "SHOOTING" else O.make_transcription(system, "COLLOCATION", N, 1, quadrature_rule=rule)
  L = O.Lagrangian(t)
  x = np.array(t.guess, dtype=np.float64); lam = np.ones(O.Callbacks(t).cons(x).size)
  eta_x, eta_v = 1e-2, 1e-3
  for i in range(25):
    if i % 1000 == 0:                      # extra_gradient.py:56-59 (the convergence test at i = 0 cannot fire: x_old = x + 20)
      eta_x *= 0.999; eta_v *= 0.999
    x, lam = L.step(x, lam, eta_x, eta_v)
  _close(x, FIX[key + "/x"], key + " x", rtol=1e-12)
  _close(lam, FIX[key + "/v"], key + " lambda", rtol=1e-12)
  assert O.Callbacks(t).fun(x) == pytest.approx(float(FIX[key + "/fun"]), rel=1e-11, abs=1e-13)


_DRAWS = os.path.join(HERE, "golden", "reference_solve_draws.npz")


@pytest.mark.skipif(not os.path.exists(_DRAWS), reason="tests/golden/reference_solve_draws.npz not generated")
def test_reference_solves_of_the_headline_workloads_instances_are_feasible_for_the_oracle():
  """tests/golden/reference_solve_draws.npz (round 6, tests/golden/make_reference_full_draws.py): rows of the batch bench.py draws (x0 = clip(x_0 + 0.1 xi), seed 2019; rows 0..2, later 0..10)
  and README.md:83's literal, each through the reference's solve() at N = 100.  The start states are the bench's, the reference's end points are feasible for the oracle's
  constraints of the same instance and have the reference's cost under the oracle's objective."""
  from bench import build_workload
  d = np.load(_DRAWS)
  x0b = build_workload(16, 100, 2019)[0]
  for key in sorted({k.rsplit("/", 1)[0] for k in d.files}):
    _, rule, row = key.split("/")
    x0, z_ref, c_ref = d[key + "/x0"], d[key + "/xs_and_us"], float(d[key + "/cost"])
    if row != "x_0": np.testing.assert_array_equal(x0, x0b[int(row)])
    system = O.SYSTEMS["CARTPOLE"]()
    system.x_0 = np.asarray(x0, dtype=np.float64)
    t = O.make_transcription(system, "COLLOCATION", 100, 1, quadrature_rule=rule)
    cb = O.Callbacks(t)
    assert float(cb.fun(z_ref)) == pytest.approx(c_ref, rel=1e-12), key
    assert np.abs(cb.cons(z_ref)).max() < 1e-7, (key, np.abs(cb.cons(z_ref)).max())
    np.testing.assert_allclose(z_ref[:4], x0, rtol=0, atol=1e-9)      # (the pinned first knot)


_SWEEP = os.path.join(HERE, "golden", "reference_solve_sweep.npz")


def _sweep_rows():
  """tools/bench_configs.py: measure(), config 4 -- the generator calls in their order"""
  rng = np.random.default_rng(2019)
  rng.standard_normal((8192, 2))
  B = 2048
  params = np.stack([rng.uniform(0.1, 0.5, B), rng.uniform(1, 5, B), rng.uniform(0.2, 0.8, B)], axis=1)
  return params, rng.uniform(0.5, 0.99, (B, 1))


@pytest.mark.skipif(not os.path.exists(_SWEEP), reason="tests/golden/reference_solve_sweep.npz not generated")
def test_reference_solves_of_the_parameter_sweeps_instances_match_the_oracle():
  """tests/golden/reference_solve_sweep.npz (round 6, tests/golden/make_reference_sweep.py): rows 0..3 of BASELINE config 4's batch -- CANCERTREATMENT single shooting 1 x 100
  with per-instance (r, a, delta) and start state -- through the reference's solve().  The instances are the bench's; the oracle's objective at the reference's end point is the
  reference's cost, and the oracle's restatement of the SciPy call ends at the same cost (single shooting: 1e-4, as for the default-parameter fixtures)."""
  d = np.load(_SWEEP)
  params, x0 = _sweep_rows()
  for key in sorted({k.rsplit("/", 1)[0] for k in d.files}):
    i = int(key.split("/")[1])
    np.testing.assert_array_equal(d[key + "/params"], params[i]); np.testing.assert_array_equal(d[key + "/x0"], x0[i])
    system = O.SYSTEMS["CANCERTREATMENT"](r=float(params[i, 0]), a=float(params[i, 1]), delta=float(params[i, 2]), x_0=float(x0[i, 0]))
    t = O.make_transcription(system, "SHOOTING", 1, 100, integration_method="HEUN")
    z_ref, c_ref = d[key + "/xs_and_us"], float(d[key + "/cost"])
    assert float(O.Callbacks(t).fun(z_ref)) == pytest.approx(c_ref, rel=1e-12), key
    res = O.solve(t, "SLSQP", max_iter=500)
    assert float(res["cost"]) == pytest.approx(c_ref, rel=1e-4), (key, float(res["cost"]), c_ref)
