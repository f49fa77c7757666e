"""The HIP path against numbers computed by THE REFERENCE'S OWN LINES (round 5): `tests/golden/reference_callbacks.npz` holds what /root/reference's
get_optimizer(...) / utils / nlp_solvers.solve returned in the build container (tests/golden/make_reference_fixtures.py; jax.numpy forwarded to numpy -- by the parity
rules still not "the reference run here").  Here the product path -- the Python mirror of the reference's API over the C-ABI over the HIP kernels,
nothing of oracle/ in between -- must reproduce them: guess and bounds of every transcription, objective(z) and constraints(z) at the reference's z (myr_eval), the
RK4 rollout of z's controls (myr_rollout), the extragradient iteration (myr_vjp / myr_exgd), the Forward-Backward Sweep (myr_fbsm), and the optima of the
reference's SLSQP solves (myr_solve).  The fixtures travel; /root/reference does not."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = np.load(os.path.join(HERE, "golden", "reference_callbacks.npz"))
KEYS = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.endswith("/objective")})
CFG = Config(verbose=False, plot=False)
SHIM_ONLY = {"PREDATORPREY/TRAPEZOIDAL/-/7x1"}      # (the reference cannot form it, quirk Q2; the stand-in's NaN is not a reference number)


def _hp(name, tr, method, N, cpi, **kw):
  if tr == "SHOOTING" or tr == "SHOOTING_":
    return HParams(system=SystemType[name], optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod[method], intervals=N, controls_per_interval=cpi, **kw)
  return HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[tr], intervals=N, **kw)


def _close(a, b, what, rtol):
  a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
  assert a.shape == b.shape, (what, a.shape, b.shape)
  fin = np.isfinite(b)
  assert np.array_equal(np.isfinite(a), fin), what
  scale = max(1.0, float(np.abs(b[fin]).max())) if fin.any() else 1.0
  assert np.abs(a[fin] - b[fin]).max(initial=0.0) <= rtol * scale, (what, float(np.abs(a[fin] - b[fin]).max(initial=0.0)), scale)


@pytest.mark.parametrize("key", [k for k in KEYS if k not in SHIM_ONLY])
def test_hip_callbacks_reproduce_the_reference_lines(key):
  name, tr, method, shape = key.split("/")
  N, cpi = (int(v) for v in shape.split("x"))
  hp = _hp(name, tr, method, N, cpi)
  opt = get_optimizer(hp, CFG, hp.system())
  g = lambda f: FIX[key + "/" + f]
  _close(opt.guess, g("guess"), key + " guess", 1e-12)
  b_ref = g("bounds")
  assert np.array_equal(np.isinf(opt.bounds), np.isinf(b_ref)), key
  _close(np.where(np.isinf(opt.bounds), 0.0, opt.bounds), np.where(np.isinf(b_ref), 0.0, b_ref), key + " bounds", 1e-13)
  z = g("z")
  f_ref = float(g("objective"))
  f = float(opt.objective(z))
  if np.isfinite(f_ref):
    assert f == pytest.approx(f_ref, rel=1e-11, abs=1e-12), key
  else:
    assert not np.isfinite(f), key
  _close(opt.constraints(z), g("constraints"), key + " constraints", 1e-11)
  if key + "/rollout_rk4_cost" in FIX.files and np.isfinite(float(g("rollout_rk4_cost"))):
    from myriad_amd.utils import get_state_trajectory_and_cost
    xs, us = opt.unravel(z)
    hp_r = _hp(name, tr, method, N, cpi) if tr != "SHOOTING" else _hp(name, tr, "RK4", N, cpi)
    hp_r.integration_method = IntegrationMethod.RK4
    xr, cr = get_state_trajectory_and_cost(hp_r, hp_r.system(), hp_r.system().x_0, us)
    _close(xr, g("rollout_rk4_xs"), key + " rollout states", 1e-10)
    assert cr == pytest.approx(float(g("rollout_rk4_cost")), rel=1e-10, abs=1e-11), key


# (round 6: + tests/golden/reference_solve_full.npz -- the HEADLINE problem at its full horizon, CARTPOLE Hermite-Simpson N = 100 from the reference's own start state,
#  through the reference's solve(): 44 minutes of SLSQP with complex-step Jacobians, MYRIAD_REF_FULL_SOLVE=1 in tests/golden/make_reference_fixtures.py)
_FULL = np.load(os.path.join(HERE, "golden", "reference_solve_full.npz"))
SOLVEFIX = {k: FIX[k] for k in FIX.files if k.startswith("solve/")}
SOLVEFIX.update({k: _FULL[k] for k in _FULL.files})
SOLVE_KEYS = sorted({k.rsplit("/", 1)[0] for k in SOLVEFIX if k.endswith("/cost")})


@pytest.mark.parametrize("key", SOLVE_KEYS)
def test_hip_sqp_reaches_the_optimum_of_the_reference_solve(key):
  """nlp_solvers/__init__.py:18-98 (SLSQP branch) executed by the generator; SLSQP stops at ftol 1e-6, the device SQP at its KKT tolerances: from the reference's
  end point the device solve stays in its basin and ends within SLSQP's tolerance of it -- never above it --, feasible to 1e-8 by the device's own evaluation."""
  _, name, optimizer, rule, shape = key.split("/")
  N, cpi = (int(v) for v in shape.split("x"))
  hp = _hp(name, "SHOOTING" if optimizer == "SHOOTING" else rule, rule, N, cpi, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  z_ref = SOLVEFIX[key + "/xs_and_us"]; c_ref = float(SOLVEFIX[key + "/cost"])
  assert float(opt.objective(z_ref)) == pytest.approx(c_ref, rel=1e-10, abs=1e-12), key
  r = opt.solve_batch(x0s=np.asarray(opt.system.x_0, dtype=np.float64)[None], guess=z_ref[None])
  assert r["status"][0] == 0, (key, r["status"], r["iters"])
  assert np.abs(opt.constraints(r["xs_and_us"][0])).max() <= 1e-8, key
  tol = 2e-5 if "VANDERPOL" in key else 1e-5
  assert float(r["cost"][0]) <= c_ref + tol * max(1.0, abs(c_ref)), (key, float(r["cost"][0]), c_ref)
  if "VANDERPOL" not in key:       # (the reference's own VANDERPOL run stops far from a minimiser: tests/test_reference_fixtures.py)
    # (BASELINE config 1 at full size, 1012 variables: SLSQP's ftol 1e-6 stops it 2.5e-5 above the KKT point the device SQP reaches, -1.3543637)
    assert float(r["cost"][0]) == pytest.approx(c_ref, rel=5e-5 if shape == "10x100" else 1e-5, abs=1e-7), (key, float(r["cost"][0]), c_ref)


@pytest.mark.parametrize("key", SOLVE_KEYS)
def test_hip_sqp_from_the_reference_guess_reaches_the_reference_basin(key):
  """The same comparison FROM THE GUESS (round 6, VERDICT r5 weak #1a): the device solve starts where the reference's solve() started -- `opt.guess`, pinned equal to
  the reference's guess by test_guess_and_bounds -- with the restoration phase and second starts off, and has to end in the basin the reference's SLSQP run ended
  in: feasible, cost within SLSQP's stopping tolerance of the reference's and never above it.  Exception: VANDERPOL single shooting, where the reference's own
  run stops far from a minimiser (cost 23.74 at 1 x 20 where a KKT point has 2.92, tests/test_reference_fixtures.py) -- there the device has to do better."""
  _, name, optimizer, rule, shape = key.split("/")
  N, cpi = (int(v) for v in shape.split("x"))
  hp = _hp(name, "SHOOTING" if optimizer == "SHOOTING" else rule, rule, N, cpi, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  c_ref = float(SOLVEFIX[key + "/cost"])
  eng = opt.engine
  o = eng.default_opts(); o.max_iter = hp.max_iter; o.restoration = 0
  z0, lb, ub = opt.batch_inputs(np.asarray(opt.system.x_0, dtype=np.float64)[None], opt.system.device_params())
  np.testing.assert_allclose(z0[0], np.asarray(opt.guess, dtype=np.float64), rtol=1e-14, atol=1e-15)      # (the per-instance rule and the reference-shaped guess: one formula, last-bit differences)
  r = opt.device_solve(z0, lb, ub, opt.system.device_params(), o, second_starts=False)
  assert r["status"][0] == 0, (key, r["status"], r["iters"])
  z = r["z"][0]
  assert np.abs(opt.constraints(z)).max() <= 1e-8, key
  cost = float(r["cost"][0])
  tol = 1e-5 * max(1.0, abs(c_ref))
  if "VANDERPOL" in key:
    assert cost <= c_ref + 2e-5 * max(1.0, abs(c_ref)), (key, cost, c_ref)       # the reference's run is not at a minimiser: never above it
  else:
    assert cost <= c_ref + tol, (key, cost, c_ref)
    assert cost == pytest.approx(c_ref, rel=5e-5 if shape == "10x100" else 1e-5, abs=1e-7), (key, cost, c_ref)
    # the same basin in the variables, at the accuracy SLSQP's stopping rule leaves (the objective is flat near the optimum: SURVEY.md App. C measured 1.4e-2
    # between SLSQP and trust-constr in the controls at N = 100)
    z_ref = SOLVEFIX[key + "/xs_and_us"]
    if shape != "10x100":      # (BASELINE config 1: the end controls carry the quadrature weight h / 2 = 5e-4 and are left undetermined at SLSQP's tolerance -- SURVEY.md App. C)
      assert np.abs(z - z_ref).max() <= 5e-2 * max(1.0, np.abs(z_ref).max()), (key, np.abs(z - z_ref).max())
    else:
      nx = (N + 1) * len(opt.system.x_0)
      assert np.abs(z[:nx] - z_ref[:nx]).max() <= 5e-3 * max(1.0, np.abs(z_ref[:nx]).max()), (key, np.abs(z[:nx] - z_ref[:nx]).max())


FBSM_KEYS = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.startswith("fbsm/") and k.endswith("/sweeps")})


@pytest.mark.parametrize("key", FBSM_KEYS)
def test_hip_fbsm_reproduces_the_reference_sweeps(key):
  """forward_backward_sweep.py:88-116 executed by the generator (fbsm_intervals = 200, at most 40 sweeps): the batched device kernel behind the FBSM mirror stops after
  the same number of sweeps with the same state, control and adjoint trajectories."""
  from myriad_amd.trajectory_optimizers.forward_backward_sweep import FBSM
  name = key.split("/")[1]
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.FBSM, fbsm_intervals=200)
  opt = FBSM(hp, CFG, hp.system())
  r = opt.solve_batch(max_sweeps=40)
  assert int(r["sweeps"][0]) == int(FIX[key + "/sweeps"]), (key, r["sweeps"], int(FIX[key + "/sweeps"]))
  for f in ("x", "u", "adj"):
    _close(np.asarray(r[f])[0], FIX[key + "/" + f], key + " " + f, 1e-10)


EXGD_KEYS = sorted({k.rsplit("/", 1)[0] for k in FIX.files if k.startswith("exgd/") and k.endswith("/fun")})


@pytest.mark.parametrize("key", EXGD_KEYS)
def test_hip_extragradient_reproduces_the_reference_iteration(key):
  """extra_gradient.py:10-84 executed by the generator for 25 steps: the device's extragradient step (myr_exgd) walks the same steps."""
  _, name, optimizer, rule, shape = key.split("/")
  N, cpi = (int(v) for v in shape.split("x"))
  hp = _hp(name, "SHOOTING" if optimizer == "SHOOTING" else rule, rule, N, cpi)
  opt = get_optimizer(hp, CFG, hp.system())
  x = np.array(opt.guess, dtype=np.float64); lam = np.ones(np.asarray(opt.constraints(x)).size)
  x, lam = opt.extragradient_step(x, lam, 1e-2 * 0.999, 1e-3 * 0.999, nsteps=25)
  _close(x, FIX[key + "/x"], key + " x", 1e-10)
  _close(lam, FIX[key + "/v"], key + " lambda", 1e-10)


_DRAWS = os.path.join(HERE, "golden", "reference_solve_draws.npz")


@pytest.mark.skipif(not os.path.exists(_DRAWS), reason="tests/golden/reference_solve_draws.npz not generated")
def test_hip_sqp_solves_instances_of_the_headline_batch_to_the_reference_optimum():
  """Round 6, late: instances of the HEADLINE WORKLOAD itself -- rows 0..2 of the batch bench.py draws, CARTPOLE Hermite-Simpson N = 100 -- and README.md:83's literal
  (trapezoidal N = 100) solved by the reference's solve() (tests/golden/make_reference_full_draws.py, ~45 minutes of SLSQP each).  The device solves the same instances in ONE
  batched call from the reference-shaped guess, restoration and second starts off: KKT points, feasible to 1e-8, cost within SLSQP's stopping tolerance of the reference's
  and never above it, states within 5e-3 and all variables within 5 % of the reference's end point."""
  d = np.load(_DRAWS)
  for rule in ("HERMITE_SIMPSON", "TRAPEZOIDAL"):
    keys = sorted(k.rsplit("/", 1)[0] for k in d.files if k.startswith(f"draw/{rule}/") and k.endswith("/cost"))
    if not keys: continue
    hp = _hp("CARTPOLE", rule, None, 100, 1, nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, CFG, hp.system())
    x0 = np.stack([d[k + "/x0"] for k in keys])
    eng = opt.engine
    o = eng.default_opts(); o.max_iter = hp.max_iter; o.restoration = 0
    z0, lb, ub = opt.batch_inputs(x0, opt.system.device_params())
    r = opt.device_solve(z0, lb, ub, opt.system.device_params(), o, second_starts=False)
    assert (r["status"] == 0).all(), (rule, r["status"], r["iters"])
    for b, k in enumerate(keys):
      z, z_ref, c_ref = r["z"][b], d[k + "/xs_and_us"], float(d[k + "/cost"])
      cost = float(r["cost"][b])
      assert r["kkt"][b, 0] <= 1e-8 and np.abs(opt.constraints(z)).max() <= 1e-8, (k, r["kkt"][b])      # (defects: the start state enters through the bounds)
      assert cost <= c_ref + 1e-5 * max(1.0, abs(c_ref)), (k, cost, c_ref)
      assert cost == pytest.approx(c_ref, rel=1e-5), (k, cost, c_ref)
      nx = (opt.engine.n // 5) * 4
      assert np.abs(z[:nx] - z_ref[:nx]).max() <= 5e-3 * max(1.0, np.abs(z_ref[:nx]).max()), (k, np.abs(z[:nx] - z_ref[:nx]).max())
      assert np.abs(z - z_ref).max() <= 5e-2 * max(1.0, np.abs(z_ref).max()), (k, np.abs(z - z_ref).max())


_SWEEP = os.path.join(HERE, "golden", "reference_solve_sweep.npz")


@pytest.mark.skipif(not os.path.exists(_SWEEP), reason="tests/golden/reference_solve_sweep.npz not generated")
def test_hip_sqp_solves_instances_of_the_parameter_sweep_to_the_reference_optimum():
  """Round 6, late: rows 0..3 of BASELINE config 4's batch (CANCERTREATMENT single shooting 1 x 100, per-instance r, a, delta and start state: the draws of
  tools/bench_configs.py) solved by the reference's solve() (tests/golden/make_reference_sweep.py).  The device solves them in one batched call with a parameter row per instance:
  KKT points, cost within SLSQP's stopping tolerance of the reference's and never above it, controls within 5 % of the reference's."""
  d = np.load(_SWEEP)
  keys = sorted({k.rsplit("/", 1)[0] for k in d.files})
  hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.SHOOTING, max_iter=500, nlpsolver=NLPSolverType.SQP)
  assert hp.intervals == 1 and hp.controls_per_interval == 100 and hp.integration_method == IntegrationMethod.HEUN      # (the reference's defaults: config 4's shape)
  opt = get_optimizer(hp, CFG, hp.system())
  params = np.stack([d[k + "/params"] for k in keys]); x0 = np.stack([d[k + "/x0"] for k in keys])
  r = opt.solve_batch(x0s=x0, params=params)
  assert (r["status"] == 0).all(), (r["status"], r["iters"])
  for b, k in enumerate(keys):
    c_ref, z_ref = float(d[k + "/cost"]), d[k + "/xs_and_us"]
    cost = float(r["cost"][b])
    assert cost <= c_ref + 1e-5 * max(1.0, abs(c_ref)), (k, cost, c_ref)
    assert cost == pytest.approx(c_ref, rel=1e-5), (k, cost, c_ref)
    z = r["xs_and_us"][b]
    assert z.shape == z_ref.shape and np.abs(z - z_ref).max() <= 5e-2 * max(1.0, np.abs(z_ref).max()), (k, np.abs(z - z_ref).max())
