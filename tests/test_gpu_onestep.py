"""GPU parity tests of the trapezoidal and shooting solve paths (myr_solve with MYR_TR_TRAPEZOIDAL / MYR_TR_SHOOTING,
lane-per-trajectory kernel over TrapCore / ShootCore) against the oracle's SLSQP path and its callbacks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer

CFG = Config(verbose=False, plot=False)


def _oracle(sysname, opt, hp):
  from oracle import myriad_oracle as O
  s = O.SYSTEMS[sysname]()
  tr = O.make_transcription(s, opt, hp.intervals, hp.controls_per_interval, hp.quadrature_rule.name, hp.integration_method.name)
  return O, s, tr, O.Callbacks(tr)


@pytest.mark.parametrize("sysname,N", [("CARTPOLE", 25), ("VANDERPOL", 30), ("CANCERTREATMENT", 30), ("SIMPLECASE", 20)])
def test_trapezoidal_solve_matches_oracle_slsqp(sysname, N):
  """README.md:83's literal transcription (trapezoidal collocation)."""
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL,
               intervals=N, nlpsolver=NLPSolverType.SQP)
  O, s, tr, cb = _oracle(sysname, "COLLOCATION", hp)
  opt = get_optimizer(hp, CFG, hp.system())
  sol = opt.solve()
  z = sol['xs_and_us']
  assert np.abs(cb.cons(z)).max() <= 1e-8
  assert cb.fun(z) == pytest.approx(sol['cost'], rel=1e-12)
  r = O.solve(tr, "SLSQP", extra_options={"ftol": 1e-13}, cb=cb)
  assert sol['cost'] == pytest.approx(r['cost'], rel=1e-7)
  assert sol['cost'] <= r['cost'] + 1e-7 * max(1.0, abs(r['cost']))
  assert np.abs(z - r['xs_and_us']).max() < 1e-4
  assert sol['lambda'].shape == (N * hp.state_size,)


@pytest.mark.parametrize("sysname,kw,method", [
  ("SIMPLECASE", dict(intervals=10, controls_per_interval=100), "HEUN"),       # BASELINE config 1
  ("VANDERPOL", dict(intervals=1, controls_per_interval=50), "HEUN"),          # config 3 shape
  ("CANCERTREATMENT", dict(intervals=1, controls_per_interval=100), "HEUN"),   # config 4 shape
  ("CARTPOLE", dict(intervals=10, controls_per_interval=5), "HEUN"),
  ("SIMPLECASE", dict(intervals=4, controls_per_interval=10), "EULER"),
  ("VANDERPOL", dict(intervals=1, controls_per_interval=20), "RK4"),
  ("CANCERTREATMENT", dict(intervals=1, controls_per_interval=30), "RK4"),
  ("CARTPOLE", dict(intervals=5, controls_per_interval=4), "RK4"),
  ("VANDERPOL", dict(intervals=2, controls_per_interval=20), "MIDPOINT"),        # the reference's "midpoint" rule (quirk Q11)
  ("CANCERTREATMENT", dict(intervals=1, controls_per_interval=40), "MIDPOINT"),
])
def test_shooting_solve_matches_oracle_slsqp(sysname, kw, method):
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod[method],
               nlpsolver=NLPSolverType.SQP, max_iter=500, **kw)
  O, s, tr, cb = _oracle(sysname, "SHOOTING", hp)
  opt = get_optimizer(hp, CFG, hp.system())
  sol = opt.solve()
  z = sol['xs_and_us']
  assert np.abs(cb.cons(z)).max() <= 1e-8
  assert cb.fun(z) == pytest.approx(sol['cost'], rel=1e-11)
  r = O.solve(tr, "SLSQP", max_iter=500, extra_options={"ftol": 1e-13}, cb=cb)
  assert sol['cost'] == pytest.approx(r['cost'], rel=1e-6)
  # stationarity of the returned multipliers on variables away from their bounds
  lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
  rr = cb.grad(z) + cb.jac(z).T @ sol['lambda']
  inact = (lb < ub) & (z - lb > 1e-3) & (ub - z > 1e-3)
  assert np.abs(rr[inact]).max() < 1e-5


def test_shooting_rk4_reference_test_config():
  """tests/tests.py:166-180 (test_RK4): SHOOTING with the RK4 rule -- three control rows per step (u[2i], u[2i+1], u[2i+2],
  utils.py:91-96), solved by the same lifted Riccati with q = (du_mid, du_next)."""
  hp = HParams(system=SystemType.SIMPLECASE, optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.RK4,
               intervals=2, controls_per_interval=5, nlpsolver=NLPSolverType.SQP)
  O, s, tr, cb = _oracle("SIMPLECASE", "SHOOTING", hp)
  opt = get_optimizer(hp, CFG, hp.system())
  assert opt.guess.size == tr.guess.size == 3 * 1 + (2 * 10 + 1) * 1
  sol = opt.solve()
  z = sol['xs_and_us']
  assert np.abs(cb.cons(z)).max() <= 1e-8 and cb.fun(z) == pytest.approx(sol['cost'], rel=1e-11)
  r = O.solve(tr, "SLSQP", max_iter=500, extra_options={"ftol": 1e-13}, cb=cb)
  assert sol['cost'] == pytest.approx(r['cost'], rel=1e-6)


def test_vanderpol_shooting_batch_config3_shape():
  """BASELINE config 3 (VANDERPOL, SHOOTING, 1 x 50 Heun), one GPU's shard: random x0, all converge; the terminal
  state is re-checked by the independent rollout kernel."""
  from oracle import myriad_oracle as O
  hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  B = 8192
  x0 = O.random_x0(O.VanDerPol(), B, seed=2019)
  res = opt.solve_batch(x0s=x0)
  # single shooting over T=10 with tight control bounds: a few instances per thousand used to crawl to max_iter (the l1
  # penalty blown up by the first blocked steps, then every full step rejected); the penalty relaxation of the merit
  # function (hs_solver.h) brings all of them home
  assert (res['status'] == 0).all(), np.bincount(res['status'])
  assert res['iters'].max() < 1000
  ok = res['status'] == 0
  xs, cost = opt.engine.rollout(x0, res['u'], 50)
  assert np.abs(xs[ok, -1, :]).max() <= 1e-7            # x_T = 0
  np.testing.assert_allclose(cost[ok], res['cost'][ok], rtol=1e-9)
  assert (res['u'] >= -0.75 - 1e-12).all() and (res['u'] <= 1.0 + 1e-12).all()
  assert res['cost'][ok].mean() == pytest.approx(2.9, rel=0.2)


def test_cancertreatment_parameter_sweep_config4_shape():
  """BASELINE config 4 (CANCERTREATMENT, SHOOTING, max_iter=500): sweep over r, a, delta and x_0."""
  hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.SHOOTING, max_iter=500, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(2019)
  B = 2048
  params = np.stack([rng.uniform(0.1, 0.5, B), rng.uniform(1, 5, B), rng.uniform(0.2, 0.8, B)], axis=1)   # r, a, delta
  x0 = rng.uniform(0.5, 0.99, (B, 1))
  res = opt.solve_batch(x0s=x0, params=params)
  assert (res['status'] == 0).mean() >= 0.999, np.bincount(res['status'])
  xs, cost = opt.engine.rollout(x0, res['u'], 100, params=params)
  np.testing.assert_allclose(cost, res['cost'], rtol=1e-9)
  np.testing.assert_allclose(xs[:, -1, :], res['x'][:, -1, :], atol=1e-8)
  assert (res['u'] >= -1e-12).all() and (res['u'] <= 2 + 1e-12).all() and (res['x'] > 0).all()
  # default-parameter instance reproduces the survey's number (App. C: 20.5735535185)
  d = opt.solve()
  assert d['cost'] == pytest.approx(20.57355337, rel=1e-8)


@pytest.mark.parametrize("sysname,opt,quad,method,kw", [
  ("CARTPOLE", "COLLOCATION", "TRAPEZOIDAL", "HEUN", dict(intervals=12)),
  ("VANDERPOL", "COLLOCATION", "TRAPEZOIDAL", "HEUN", dict(intervals=7)),
  ("CANCERTREATMENT", "COLLOCATION", "TRAPEZOIDAL", "HEUN", dict(intervals=5)),
  ("CARTPOLE", "SHOOTING", "TRAPEZOIDAL", "HEUN", dict(intervals=4, controls_per_interval=6)),
  ("VANDERPOL", "SHOOTING", "TRAPEZOIDAL", "HEUN", dict(intervals=1, controls_per_interval=50)),
  ("CANCERTREATMENT", "SHOOTING", "TRAPEZOIDAL", "HEUN", dict(intervals=1, controls_per_interval=100)),
  ("SIMPLECASE", "SHOOTING", "TRAPEZOIDAL", "HEUN", dict(intervals=10, controls_per_interval=100)),
  ("VANDERPOL", "SHOOTING", "TRAPEZOIDAL", "EULER", dict(intervals=3, controls_per_interval=4)),
  ("VANDERPOL", "SHOOTING", "TRAPEZOIDAL", "MIDPOINT", dict(intervals=3, controls_per_interval=4)),
  ("VANDERPOL", "SHOOTING", "TRAPEZOIDAL", "RK4", dict(intervals=3, controls_per_interval=4)),
  ("CARTPOLE", "SHOOTING", "TRAPEZOIDAL", "RK4", dict(intervals=2, controls_per_interval=3)),
  ("CANCERTREATMENT", "SHOOTING", "TRAPEZOIDAL", "RK4", dict(intervals=1, controls_per_interval=8)),
  ("SIMPLECASE", "SHOOTING", "TRAPEZOIDAL", "RK4", dict(intervals=4, controls_per_interval=1)),
  ("CARTPOLE", "SHOOTING", "TRAPEZOIDAL", "MIDPOINT", dict(intervals=2, controls_per_interval=5)),
  ("CANCERTREATMENT", "SHOOTING", "TRAPEZOIDAL", "MIDPOINT", dict(intervals=1, controls_per_interval=12)),
])
def test_eval_callbacks_match_oracle(sysname, opt, quad, method, kw):
  """objective / grad / constraints / jacobian of the trapezoidal and shooting transcriptions (the four callbacks the
  reference jits, nlp_solvers/__init__.py:32-40) from the eval kernels vs the oracle's autodiff."""
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType[opt], quadrature_rule=QuadratureRule[quad],
               integration_method=IntegrationMethod[method], **kw)
  O, s, tr, cb = _oracle(sysname, opt, hp)
  o = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(5)
  z = tr.guess + 0.1 * rng.standard_normal(tr.guess.size)
  if sysname == "CANCERTREATMENT":
    z[:tr.x_rows] = np.abs(z[:tr.x_rows]) + 0.05
    z[tr.x_rows:] = np.abs(z[tr.x_rows:])
  np.testing.assert_allclose(o.constraints(z), cb.cons(z), rtol=1e-11, atol=1e-12)
  assert o.objective(z) == pytest.approx(cb.fun(z), rel=1e-12)
  np.testing.assert_allclose(o.objective_grad(z), cb.grad(z), rtol=1e-10, atol=1e-11)
  np.testing.assert_allclose(o.constraints_jac(z), cb.jac(z), rtol=1e-10, atol=1e-11)


def test_scipy_branch_runs_on_shooting_gpu_callbacks():
  """The reference's default test path (tests/tests.py:46-60: SIMPLECASE, SHOOTING, SLSQP, HEUN, 1 x 50) on GPU callbacks."""
  hp = HParams(system=SystemType.SIMPLECASE, optimizer=OptimizerType.SHOOTING, nlpsolver=NLPSolverType.SLSQP,
               integration_method=IntegrationMethod.HEUN, intervals=1, controls_per_interval=50)
  sol = get_optimizer(hp, CFG, hp.system()).solve()
  assert sol['cost'] == pytest.approx(-1.3544209454574183, rel=1e-5)      # SLSQP at its default ftol=1e-6


@pytest.mark.parametrize("tag,name", [("simplecase_10x100", "SIMPLECASE"), ("vanderpol_1x50", "VANDERPOL"),
                                      ("cancertreatment_1x100", "CANCERTREATMENT")])
def test_shooting_solve_matches_golden_fixtures(tag, name, golden_dir):
  """BASELINE configs 1, 3, 4 against the committed SLSQP(ftol=1e-15) optima (tests/golden/make_shoot_golden.py)."""
  import os
  from myriad_amd import _lib
  d = np.load(os.path.join(golden_dir, f"solve_shoot_{tag}.npz"))
  B = d["z"].shape[0]
  eng = _lib.Engine(name, "SHOOTING", int(d["intervals"]), float(d["T"]), controls_per_interval=int(d["cpi"]),
                    integration_method="HEUN", max_batch=B)
  o = eng.default_opts(); o.max_iter = 500
  r = eng.solve(d["z0"], d["lb"], d["ub"], params=d["params"], opts=o)
  assert (r["status"] == 0).all(), (r["status"], r["iters"], r["kkt"])
  np.testing.assert_allclose(r["cost"], d["cost"], rtol=1e-7)
  assert (r["cost"] <= d["cost"] + 1e-7 * np.maximum(1.0, np.abs(d["cost"]))).all()   # interior-point: compl. tolerance 1e-7
  # at the default tolerances (complementarity 1e-7) weakly active control bounds (the last control of a Heun rollout barely
  # enters the objective) sit at ~sqrt(mu) from the bound: 1e-3 on the controls, 1e-4 on the states
  nx = (int(d["intervals"]) + 1) * eng.ns
  assert np.abs(r["z"] - d["z"])[:, :nx].max() < 1e-4
  assert np.abs(r["z"] - d["z"]).max() < 1e-3
  # SURVEY.md 8(c) tolerance (1e-6 on z) against the polished fixtures (|KKT| <= 1e-12, multipliers stored) needs the barrier driven
  # further down than the default: warm start from the default-tolerance solution with tight tolerances
  assert float(d["kkt"].max()) <= 1e-12
  o.tol_feas, o.tol_stat, o.tol_compl, o.mu_init = 1e-11, 1e-10, 1e-12, 1e-9
  r2 = eng.solve(r["z"], d["lb"], d["ub"], params=d["params"], opts=o)
  assert (r2["status"] == 0).all(), (r2["status"], r2["iters"], r2["kkt"])
  # a WEAKLY active bound (fixture multiplier < 1e-4: the last control of a Heun rollout hardly enters the objective) keeps an
  # interior-point iterate at slack = mu / multiplier from it (1.5e-6 at the smallest barrier parameter): 1e-5 there, 1e-6 elsewhere
  on_bound = (np.abs(d["z"] - d["lb"]) <= 1e-9) | (np.abs(d["z"] - d["ub"]) <= 1e-9)      # (degenerate: on the bound with a zero multiplier)
  weak = on_bound & (d["lb"] < d["ub"]) & ((d["zL"] + d["zU"]) < 1e-4)
  err = np.abs(r2["z"] - d["z"])
  assert err[~weak].max() < 1e-6 and err.max() < 1e-5, (err[~weak].max(), err.max())
  np.testing.assert_allclose(r2["cost"], d["cost"], rtol=1e-10)
  lam_err = np.abs(r2["lam"] - d["lam"]).max(axis=1) / np.maximum(1.0, np.abs(d["lam"]).max(axis=1))
  assert lam_err.max() < 1e-6, lam_err            # sign convention of the reference's mult_g (nlp_solvers/__init__.py:82-86)
  eng.close()


def test_trapezoid_cartpole_batch_all_converge():
  """README.md:83's literal config (CARTPOLE, trapezoidal, N=100) as a batch of random start states: every lane of every
  wavefront converges (a build in which the penalty floor reached this core lost fixed lane positions of a batch)."""
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, intervals=100, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(5)
  B = 1024
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  res = opt.solve_batch(x0s=x0)
  assert (res['status'] == 0).all(), np.nonzero(res['status'])[0][:20]
  assert res['iters'].max() < 100
  xs, cost = opt.engine.rollout(x0, res['u'], 100)
  assert np.isfinite(cost).all()
