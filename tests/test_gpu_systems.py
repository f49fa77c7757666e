"""GPU parity tests of the SURVEY.md 8(f4) systems (BIOREACTOR, GLUCOSE, MOULDFUNGICIDE, SIMPLECASEWITHBOUNDS, HIVTREATMENT,
EPIDEMICSEIRN, SEIR, BEARPOPULATIONS, PENDULUM, MOUNTAINCAR, ROCKETLANDING) through the three transcriptions: the four callbacks the reference jits
(nlp_solvers/__init__.py:32-40) against the oracle's autodiff, and the SQP solution against the oracle's KKT conditions
and its SLSQP path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer

CFG = Config(verbose=False, plot=False)
NEW = ["BIOREACTOR", "GLUCOSE", "MOULDFUNGICIDE", "SIMPLECASEWITHBOUNDS", "HIVTREATMENT", "EPIDEMICSEIRN", "SEIR", "BEARPOPULATIONS",
       "PENDULUM", "MOUNTAINCAR", "ROCKETLANDING"]


def _oracle(sysname, opt, hp):
  from oracle import myriad_oracle as O
  s = O.SYSTEMS[sysname]()
  tr = O.make_transcription(s, opt, hp.intervals, hp.controls_per_interval, hp.quadrature_rule.name, hp.integration_method.name)
  return O, s, tr, O.Callbacks(tr)


def _test_point(tr, rng):
  """a strictly interior point near the guess, relative perturbation (the systems' scales differ by 1e5)"""
  z = tr.guess * (1.0 + 0.05 * rng.standard_normal(tr.guess.size)) + 1e-3 * rng.standard_normal(tr.guess.size)
  lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
  fr = lb < ub
  lo = np.where(np.isfinite(lb), lb + 1e-3 * np.where(np.isfinite(ub), ub - lb, 1.0), -np.inf)
  hi = np.where(np.isfinite(ub), ub - 1e-3 * np.where(np.isfinite(lb), ub - lb, 1.0), np.inf)
  return np.where(fr, np.clip(z, lo, hi), lb)


@pytest.mark.parametrize("opt,quad,kw", [
  ("COLLOCATION", "HERMITE_SIMPSON", dict(intervals=6)),
  ("COLLOCATION", "TRAPEZOIDAL", dict(intervals=7)),
  ("SHOOTING", "TRAPEZOIDAL", dict(intervals=2, controls_per_interval=8)),
])
@pytest.mark.parametrize("sysname", NEW)
def test_eval_callbacks_match_oracle(sysname, opt, quad, kw):
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType[opt], quadrature_rule=QuadratureRule[quad],
               integration_method=IntegrationMethod.HEUN, **kw)
  O, s, tr, cb = _oracle(sysname, opt, hp)
  o = get_optimizer(hp, CFG, hp.system())
  z = _test_point(tr, np.random.default_rng(7))
  c_ref, J_ref, g_ref = cb.cons(z), cb.jac(z), cb.grad(z)
  np.testing.assert_allclose(o.constraints(z), c_ref, rtol=1e-11, atol=1e-12 * max(1.0, np.abs(c_ref).max()))
  assert o.objective(z) == pytest.approx(cb.fun(z), rel=1e-12)
  np.testing.assert_allclose(o.objective_grad(z), g_ref, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(g_ref).max()))
  np.testing.assert_allclose(o.constraints_jac(z), J_ref, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(J_ref).max()))
  if opt == "COLLOCATION":
    rng = np.random.default_rng(1)
    lam = rng.standard_normal(c_ref.size); v = rng.standard_normal(z.size)
    ref = g_ref + J_ref.T @ lam
    np.testing.assert_allclose(o.lagrangian_grad(z, lam), ref, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(o.constraints_jvp(z, v), J_ref @ v, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(J_ref @ v).max()))


@pytest.mark.parametrize("sysname", NEW)
def test_hs_solve_is_a_kkt_point_of_the_oracle_problem(sysname):
  N = 20
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               intervals=N, nlpsolver=NLPSolverType.SQP)
  O, s, tr, cb = _oracle(sysname, "COLLOCATION", hp)
  opt = get_optimizer(hp, CFG, hp.system())
  r = opt.solve_batch()
  if sysname == "ROCKETLANDING":
    # ROCKETLANDING does not become feasible (the oracle's SLSQP fails as well), neither from the reference's guess nor from the
    # second starts; its elastic twin converges with a slack that does not shrink as the penalty grows, and the instance is
    # reported INFEASIBLE (status 4; tests/test_gpu_elastic.py).  Checked here: the outcome is reported per instance, never
    # raised, and the returned iterate is finite and inside its bounds.
    assert r['status'][0] in (0, 4) and np.isfinite(r['cost'][0]) and np.isfinite(r['xs_and_us']).all()
    lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
    assert (r['xs_and_us'][0] >= lb - 1e-9).all() and (r['xs_and_us'][0] <= ub + 1e-9).all()
    return
  # (PENDULUM: the torque-limited swing-up jams at an infeasible point from the reference's straight-line guess and
  # converges from the second start -- test_pendulum_needs_and_gets_a_second_start below)
  assert r['status'][0] == 0, (sysname, r['status'], r['iters'], r['kkt'])
  z, lam = r['xs_and_us'][0], r['lambda'][0]
  c = cb.cons(z)
  scale = max(1.0, np.abs(z).max())          # the solver's 1e-8 is in scaled states (myr_set_var_scale)
  assert np.abs(c).max() <= 1e-8 * scale
  assert cb.fun(z) == pytest.approx(r['cost'][0], rel=1e-11)
  lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
  assert (z >= lb - 1e-12).all() and (z <= ub + 1e-12).all()
  rr = cb.grad(z) + cb.jac(z).T @ lam
  width = np.where(np.isfinite(ub - lb), ub - lb, 1.0)
  inact = (lb < ub) & (z - lb > 1e-3 * width) & (ub - z > 1e-3 * width)
  sd = max(1.0, np.abs(lam).mean() / 100.0)                      # the solver's multiplier scaling of the KKT error
  assert np.abs(rr[inact]).max() < 1e-4 * sd * max(1.0, np.abs(cb.grad(z)).max())


@pytest.mark.parametrize("sysname", ["BIOREACTOR", "MOULDFUNGICIDE", "SIMPLECASEWITHBOUNDS", "GLUCOSE", "BEARPOPULATIONS", "MOUNTAINCAR"])
def test_hs_solve_cost_matches_oracle_slsqp(sysname):
  N = 8
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               intervals=N, nlpsolver=NLPSolverType.SQP)
  O, s, tr, cb = _oracle(sysname, "COLLOCATION", hp)
  sol = get_optimizer(hp, CFG, hp.system()).solve()
  r = O.solve(tr, "SLSQP", max_iter=500, extra_options={"ftol": 1e-14}, cb=cb)
  assert sol['cost'] <= r['cost'] + 1e-6 * max(1.0, abs(r['cost']))
  assert sol['cost'] == pytest.approx(r['cost'], rel=1e-5)


def test_variable_scaling_is_what_makes_the_large_state_systems_converge(monkeypatch):
  """EPIDEMICSEIRN has states O(1e3) next to a control O(1): unscaled, the absolute regularisation constants of the
  solver stall it (status MAXITER, reported not raised); with the power-of-two state scales of system.var_scale() it
  converges, to the same optimum the unscaled run was approaching."""
  hp = HParams(system=SystemType.EPIDEMICSEIRN, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               intervals=20, nlpsolver=NLPSolverType.SQP)
  scaled = get_optimizer(hp, CFG, hp.system()).solve_batch()
  monkeypatch.setenv("MYRIAD_VAR_SCALE", "0")
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0")            # (one attempt: this is about the scaling, not the starting point)
  unscaled = get_optimizer(hp, CFG, hp.system()).solve_batch(max_iter=300)
  assert scaled['status'][0] == 0 and scaled['iters'][0] < 100
  assert unscaled['status'][0] == 1 and unscaled['iters'][0] == 300
  assert scaled['cost'][0] == pytest.approx(unscaled['cost'][0], rel=2e-2)      # 13.3967 vs ~13.49 after 300 stalled iterations


@pytest.mark.parametrize("sysname", ["PENDULUM", "MOUNTAINCAR"])
def test_clipped_fields_match_oracle_outside_the_box(sysname):
  """jnp.clip / angle_normalize of the gym-style systems (pendulum.py:94-120, mountain_car.py:83-89) are kept on the
  device: evaluate well outside the bounds, where they are not identities."""
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=5)
  O, s, tr, cb = _oracle(sysname, "COLLOCATION", hp)
  o = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(11)
  z = 3.0 * tr.guess + 4.0 * rng.standard_normal(tr.guess.size)
  np.testing.assert_allclose(o.constraints(z), cb.cons(z), rtol=1e-11, atol=1e-11)
  assert o.objective(z) == pytest.approx(cb.fun(z), rel=1e-12)
  np.testing.assert_allclose(o.constraints_jac(z), cb.jac(z), rtol=1e-10, atol=1e-11)
  np.testing.assert_allclose(o.objective_grad(z), cb.grad(z), rtol=1e-10, atol=1e-11)


def test_discrete_system_is_refused_like_the_reference():
  hp = HParams(system=SystemType.INVASIVEPLANT, optimizer=OptimizerType.SHOOTING)
  with pytest.raises(NotImplementedError):                       # trajectory_optimizers/base.py:66-67
    get_optimizer(hp, CFG, hp.system())

def _shooting_checks(sysname, sys_host, sys_oracle):
  from oracle import myriad_oracle as O
  for method in ("HEUN", "EULER", "MIDPOINT", "RK4"):
    hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.SHOOTING, intervals=2, controls_per_interval=6,
                 integration_method=IntegrationMethod[method], nlpsolver=NLPSolverType.SQP, max_iter=500)
    tr = O.shooting(sys_oracle, 2, 6, method)
    cb = O.Callbacks(tr)
    o = get_optimizer(hp, CFG, sys_host)
    rng = np.random.default_rng(8)
    z = np.abs(tr.guess * (1.0 + 0.05 * rng.standard_normal(tr.guess.size))) + 0.05
    np.testing.assert_allclose(o.constraints(z), cb.cons(z), rtol=1e-11, atol=1e-11 * max(1.0, np.abs(z).max()))
    assert o.objective(z) == pytest.approx(cb.fun(z), rel=1e-12)
    g_ref = cb.grad(z)
    np.testing.assert_allclose(o.objective_grad(z), g_ref, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(g_ref).max()))
    np.testing.assert_allclose(o.constraints_jac(z), cb.jac(z), rtol=1e-10, atol=1e-11)
  # solve (Heun, 1 x 20 as the reference's default shape)
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=20, nlpsolver=NLPSolverType.SQP, max_iter=500)
  tr = O.shooting(sys_oracle, 1, 20, "HEUN")
  cb = O.Callbacks(tr)
  r = get_optimizer(hp, CFG, sys_host).solve_batch()
  assert r['status'][0] == 0, (sysname, r['status'], r['iters'], r['kkt'])
  z, lam = r['xs_and_us'][0], r['lambda'][0]
  assert np.abs(cb.cons(z)).max() <= 1e-8 * max(1.0, np.abs(z).max())
  assert cb.fun(z) == pytest.approx(r['cost'][0], rel=1e-10)
  lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
  rr = cb.grad(z) + cb.jac(z).T @ lam
  width = np.where(np.isfinite(ub - lb), ub - lb, 1.0)
  inact = (lb < ub) & (z - lb > 1e-3 * width) & (ub - z > 1e-3 * width)
  if inact.any():
    assert np.abs(rr[inact]).max() < 1e-4 * max(1.0, np.abs(cb.grad(z)).max())


@pytest.mark.parametrize("sysname", ["BACTERIA", "TUMOUR"])
def test_terminal_cost_systems_follow_the_reference_rule(sysname):
  """Linear terminal costs (bacteria.py:84-86, tumour.py:106-108): applied by the TRAPEZOIDAL objective
  (trapezoidal.py:126-127) and the rollout (utils.py:295-296), NOT by the Hermite-Simpson objective
  (hermite_simpson.py:243-257 has no terminal term); shooting applies it to the integrated end state."""
  from myriad_amd.utils import get_state_trajectory_and_cost
  for quad, N in (("TRAPEZOIDAL", 7), ("HERMITE_SIMPSON", 5)):
    hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[quad], intervals=N,
                 nlpsolver=NLPSolverType.SQP)
    O, s, tr, cb = _oracle(sysname, "COLLOCATION", hp)
    o = get_optimizer(hp, CFG, hp.system())
    rng = np.random.default_rng(3)
    z = np.abs(tr.guess * (1.0 + 0.05 * rng.standard_normal(tr.guess.size))) + 0.05
    np.testing.assert_allclose(o.constraints(z), cb.cons(z), rtol=1e-11, atol=1e-10)
    assert o.objective(z) == pytest.approx(cb.fun(z), rel=1e-12, abs=1e-12)
    np.testing.assert_allclose(o.objective_grad(z), cb.grad(z), rtol=1e-10, atol=1e-11)
    lam = rng.standard_normal(cb.cons(z).size)
    ref = cb.grad(z) + cb.jac(z).T @ lam
    np.testing.assert_allclose(o.lagrangian_grad(z, lam), ref, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(ref).max()))
  # solve (trapezoidal): KKT of the oracle problem, terminal gradient included
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=20,
               nlpsolver=NLPSolverType.SQP)
  O, s, tr, cb = _oracle(sysname, "COLLOCATION", hp)
  opt = get_optimizer(hp, CFG, hp.system())
  r = opt.solve_batch()
  assert r['status'][0] == 0, (r['status'], r['iters'], r['kkt'])
  z, lam = r['xs_and_us'][0], r['lambda'][0]
  assert np.abs(cb.cons(z)).max() <= 1e-8 * max(1.0, np.abs(z).max())
  assert cb.fun(z) == pytest.approx(r['cost'][0], rel=1e-11)
  lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
  rr = cb.grad(z) + cb.jac(z).T @ lam
  width = np.where(np.isfinite(ub - lb), ub - lb, 1.0)
  inact = (lb < ub) & (z - lb > 1e-3 * width) & (ub - z > 1e-3 * width)
  assert np.abs(rr[inact]).max() < 1e-4 * max(1.0, np.abs(cb.grad(z)).max())
  # rollout cost includes the terminal term
  hp2 = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, intervals=20)
  us = r['u'][0]
  _, c = get_state_trajectory_and_cost(hp2, hp2.system(), hp2.system().x_0, us)
  _, c_ref = O.get_state_trajectory_and_cost(s, hp2.num_steps, hp2.integration_method.name, s.x_0, us)
  assert c == pytest.approx(c_ref, rel=1e-11)
  # shooting (the reference's default optimiser): the terminal cost sits on the INTEGRATED end state of the last interval
  # (shooting.py:206-208); callbacks against the oracle's autodiff, then a solve checked against the oracle's KKT conditions
  _shooting_checks(sysname, hp.system(), O.SYSTEMS[sysname]())


@pytest.mark.parametrize("sysname,kw", [("HARVEST", {}), ("TIMBERHARVEST", {"r": 0.3})])
def test_time_dependent_cost_systems(sysname, kw):
  """g(x, u, t) with explicit time (harvest.py:61-62, timber_harvest.py:84-85): the point's time -- linspace(0, T, K) of
  hermite_simpson.py:252 / trapezoidal.py:124, the stage times of utils.py:31-54 in the rollout -- reaches the generated
  cost through a slot of the parameter vector."""
  from oracle import myriad_oracle as O
  from myriad_amd import systems as HS_
  from myriad_amd.utils import get_state_trajectory_and_cost
  so = O.SYSTEMS[sysname](**kw)
  sh = getattr(HS_, type(so).__name__)(**kw)
  for quad, N in (("HERMITE_SIMPSON", 6), ("TRAPEZOIDAL", 7)):
    hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[quad], intervals=N,
                 nlpsolver=NLPSolverType.SQP)
    tr = O.make_transcription(so, "COLLOCATION", N, 1, quad, "HEUN")
    cb = O.Callbacks(tr)
    o = get_optimizer(hp, CFG, sh)
    rng = np.random.default_rng(4)
    z = np.abs(tr.guess * (1.0 + 0.05 * rng.standard_normal(tr.guess.size))) + 0.05
    np.testing.assert_allclose(o.constraints(z), cb.cons(z), rtol=1e-11, atol=1e-11)
    assert o.objective(z) == pytest.approx(cb.fun(z), rel=1e-12)
    np.testing.assert_allclose(o.objective_grad(z), cb.grad(z), rtol=1e-10, atol=1e-11)
    lam = rng.standard_normal(cb.cons(z).size)
    ref = cb.grad(z) + cb.jac(z).T @ lam
    np.testing.assert_allclose(o.lagrangian_grad(z, lam), ref, rtol=1e-10, atol=1e-11 * max(1.0, np.abs(ref).max()))
    r = o.solve_batch()
    assert r['status'][0] == 0, (quad, r['status'], r['iters'], r['kkt'])
    zs, lm = r['xs_and_us'][0], r['lambda'][0]
    assert np.abs(cb.cons(zs)).max() <= 1e-8 * max(1.0, np.abs(zs).max())
    assert cb.fun(zs) == pytest.approx(r['cost'][0], rel=1e-11)
    lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
    rr = cb.grad(zs) + cb.jac(zs).T @ lm
    width = np.where(np.isfinite(ub - lb), ub - lb, 1.0)
    inact = (lb < ub) & (zs - lb > 1e-3 * width) & (ub - zs > 1e-3 * width)
    if inact.any():
      assert np.abs(rr[inact]).max() < 1e-4 * max(1.0, np.abs(cb.grad(zs)).max())
  hp2 = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, intervals=20)
  us = 0.3 * np.ones((21, 1))
  for method in ("EULER", "HEUN", "MIDPOINT"):
    hp2.integration_method = IntegrationMethod[method]
    _, c = get_state_trajectory_and_cost(hp2, sh, sh.x_0, us)
    _, c_ref = O.get_state_trajectory_and_cost(so, hp2.num_steps, method, so.x_0, us)
    assert c == pytest.approx(c_ref, rel=1e-11), method
  _shooting_checks(sysname, sh, so)


def test_predator_prey_shooting_and_collocation_refusal():
  """PREDATORPREY (terminal cost + ONE pinned terminal state, predator_prey.py:47): shooting works (shooting.py:255-258 loops
  over the non-None entries), the collocation optimisers fail on the None entries as the reference's do."""
  from oracle import myriad_oracle as O
  so = O.PredatorPrey()
  hp = HParams(system=SystemType.PREDATORPREY, optimizer=OptimizerType.SHOOTING, intervals=40, controls_per_interval=1,
               nlpsolver=NLPSolverType.SQP, max_iter=500)
  tr = O.shooting(so, 40, 1, "HEUN")
  cb = O.Callbacks(tr)
  o = get_optimizer(hp, CFG, hp.system())
  np.testing.assert_array_equal(o.bounds, tr.bounds)
  np.testing.assert_allclose(o.guess, tr.guess, rtol=1e-12, atol=1e-12)
  rng = np.random.default_rng(2)
  z = np.abs(tr.guess * (1.0 + 0.02 * rng.standard_normal(tr.guess.size))) + 0.02
  np.testing.assert_allclose(o.constraints(z), cb.cons(z), rtol=1e-11, atol=1e-11)
  assert o.objective(z) == pytest.approx(cb.fun(z), rel=1e-12)
  np.testing.assert_allclose(o.objective_grad(z), cb.grad(z), rtol=1e-10, atol=1e-11)
  np.testing.assert_allclose(o.constraints_jac(z), cb.jac(z), rtol=1e-10, atol=1e-11)
  r = o.solve_batch()
  assert r['status'][0] == 0, (r['status'], r['iters'], r['kkt'])
  zs = r['xs_and_us'][0]
  assert np.abs(cb.cons(zs)).max() <= 1e-8 and cb.fun(zs) == pytest.approx(r['cost'][0], rel=1e-10)
  assert zs[40 * 3 + 2] == pytest.approx(5.0, abs=1e-9)        # the pinned terminal state x_2(T) = B
  for quad in (QuadratureRule.TRAPEZOIDAL, QuadratureRule.HERMITE_SIMPSON):
    hpc = HParams(system=SystemType.PREDATORPREY, optimizer=OptimizerType.COLLOCATION, quadrature_rule=quad, intervals=10)
    with pytest.raises(TypeError):
      get_optimizer(hpc, CFG, hpc.system())


def test_pendulum_needs_and_gets_a_second_start(monkeypatch):
  """From the reference's straight-line guess the torque-limited swing-up ends at an infeasible stationary point (MAXITER,
  reported).  With the elastic phase switched off (tests/test_gpu_elastic.py covers it) the second start -- oscillating controls with
  the states of their rollout -- reaches a feasible KKT point; `iters` counts both attempts."""
  hp = HParams(system=SystemType.PENDULUM, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               intervals=20, nlpsolver=NLPSolverType.SQP)
  monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0")
  jam = get_optimizer(hp, CFG, hp.system()).solve_batch()
  assert jam['status'][0] == 1 and jam['iters'][0] == hp.max_iter and jam['kkt'][0, 0] > 1e-4      # infeasible
  monkeypatch.delenv("MYRIAD_SECOND_STARTS")
  opt = get_optimizer(hp, CFG, hp.system())
  r = opt.solve_batch()
  assert r['status'][0] == 0 and r['iters'][0] > hp.max_iter and r['start'][0] == 2 and r['restored'][0] == 0
  assert np.abs(opt.constraints(r['xs_and_us'][0])).max() <= 1e-8
  assert r['cost'][0] < 0.5 * jam['cost'][0]                    # 25.54 against 61.4 at the jam
  x = r['x'][0]
  assert abs(x[-1, 0] - np.pi) < 1e-9 and np.abs(x[:, 0]).max() > 0.5 * np.pi and (np.diff(np.sign(x[:, 1])) != 0).sum() >= 2   # swings
  sol = opt.solve()                                             # the reference-shaped call takes the same path
  assert sol['cost'] == pytest.approx(r['cost'][0], rel=1e-9)
  monkeypatch.delenv("MYRIAD_ELASTIC")                          # default: the elastic phase comes first and finds the same optimum
  r3 = get_optimizer(hp, CFG, hp.system()).solve_batch()
  assert r3['status'][0] == 0 and r3['restored'][0] == 1 and r3['cost'][0] == pytest.approx(r['cost'][0], rel=1e-6)


@pytest.mark.parametrize("sysname", ["BEARPOPULATIONS", "GLUCOSE", "HIVTREATMENT"])
def test_trapezoidal_solve_is_a_kkt_point_of_the_oracle_problem(sysname):
  """The trapezoidal solver on a system with two controls (BEARPOPULATIONS) and on two with one: feasibility, cost and
  stationarity against the oracle's callbacks.  (Regression: the wavefront kernel's general sweep produced NaN for
  BEARPOPULATIONS -- it addressed the Hessian records at the Hermite-Simpson stride; fixed, see the agreement test below.)"""
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL,
               intervals=30, nlpsolver=NLPSolverType.SQP)
  O, s, tr, cb = _oracle(sysname, "COLLOCATION", hp)
  opt = get_optimizer(hp, CFG, hp.system())
  r = opt.solve_batch()
  assert r['status'][0] == 0 and r['attempts'][0] == 1, (sysname, r['status'], r['iters'], r['kkt'])
  z, lam = r['xs_and_us'][0], r['lambda'][0]
  assert np.abs(cb.cons(z)).max() <= 1e-8 * max(1.0, np.abs(z).max())
  assert cb.fun(z) == pytest.approx(r['cost'][0], rel=1e-11)
  lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
  rr = cb.grad(z) + cb.jac(z).T @ lam
  width = np.where(np.isfinite(ub - lb), ub - lb, 1.0)
  inact = (lb < ub) & (z - lb > 1e-3 * width) & (ub - z > 1e-3 * width)
  sd = max(1.0, np.abs(lam).mean() / 100.0)
  assert np.abs(rr[inact]).max() < 1e-4 * sd * max(1.0, np.abs(cb.grad(z)).max())


@pytest.mark.parametrize("sysname", ["BEARPOPULATIONS"])      # (the other closed-form system outside the matrix-core form, ROCKETLANDING, has no optimum to agree on; PENDULUM's elastic twin, three controls, is covered by the swing-up test below)
def test_trapezoidal_general_sweep_agrees_with_the_lane_kernel(sysname, monkeypatch):
  """A system outside the matrix-core sweep (BEARPOPULATIONS: two controls) under the trapezoidal scheme on
  the wavefront kernel's GENERAL sweep -- which read the end point's Hessian record at the Hermite-Simpson stride (point 2k + 2
  instead of k + 1) until round 3 and ended in NaN -- against the lane kernel: same status, same iteration count, same optimum."""
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL,
               intervals=30, nlpsolver=NLPSolverType.SQP)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  s0 = hp.system()
  rng = np.random.default_rng(1)
  x0s = s0.x_0[None] * (1.0 + 0.02 * rng.standard_normal((5, s0.x_0.shape[0])))
  res = {}
  for mode in ("wave", "lane"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    res[mode] = get_optimizer(hp, CFG, hp.system()).solve_batch(x0s=x0s)
  w, l = res["wave"], res["lane"]
  assert (w["status"] == 0).all() and (l["status"] == 0).all(), (w["status"], w["iters"], l["status"], l["iters"])
  assert np.array_equal(w["iters"], l["iters"])
  np.testing.assert_allclose(w["cost"], l["cost"], rtol=1e-9)
  np.testing.assert_allclose(w["xs_and_us"], l["xs_and_us"], rtol=1e-6, atol=1e-8)


def test_small_batches_take_the_one_wavefront_kernel(monkeypatch):
  """Cases on which the two-wavefront fused kernel (W = 2, round 3's small-batch mode) went wrong -- tools/dev/w2_probe.py:
  MOULDFUNGICIDE N = 100 stalled at cost 560, N = 6 ended in NaN; the first of three CANCERTREATMENT instances at N = 100 stalled at
  17.1 -- solve by default (W = 1 for every batch size since then; MYRIAD_FUSED_WAVES=2 is a development switch)."""
  monkeypatch.delenv("MYRIAD_FUSED_WAVES", raising=False)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  for name, N, B, cost in (("MOULDFUNGICIDE", 100, 1, 85.35321226), ("MOULDFUNGICIDE", 6, 1, 85.47233574), ("CANCERTREATMENT", 100, 3, 20.57331484)):
    hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
                 intervals=N, nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, CFG, hp.system())
    x0 = np.tile(opt.system.x_0, (B, 1)) * (1.0 + 0.01 * np.arange(B)[:, None])
    r = opt.solve_batch(x0s=x0, max_iter=300)
    assert (r["status"] == 0).all(), (name, N, r["status"], r["iters"], r["cost"])
    assert r["cost"][0] == pytest.approx(cost, rel=1e-8)
    assert np.abs(opt.constraints(r["xs_and_us"][0])).max() <= 1e-8


def test_rocketlanding_wave_and_lane_kernels_take_the_same_iterations(monkeypatch):
  """ROCKETLANDING (six states, two controls: the only closed-form system with more than four states) has no optimum to compare --
  no solver reaches feasibility -- but the wavefront kernel's general sweep and the lane kernel must walk the same path: after a
  fixed number of iterations (MAXITER on both) the iterates agree.  (Its INFEASIBLE verdict must not be an artefact of one kernel.)
  Both schemes; under Hermite-Simpson both kernels are also pinned to the numbers of the host build of the solver source.  (Round 3's
  lane kernel of this instantiation was miscompiled -- objective 1.802207 for 2.637376 at the first iterate -- and refused; the cause is
  the machine scheduler on the one giant block of the backward loop, cured by splitting the region: tools/dev/repro/lane_rocket_misched.)"""
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  hp = HParams(system=SystemType.ROCKETLANDING, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL,
               intervals=20, nlpsolver=NLPSolverType.SQP)
  res = {}
  for mode in ("wave", "lane"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    res[mode] = get_optimizer(hp, CFG, hp.system()).solve_batch(max_iter=12)
  w, l = res["wave"], res["lane"]
  assert w["status"][0] == l["status"][0] == 1 and w["iters"][0] == l["iters"][0] == 12
  scale = np.maximum(1.0, np.abs(l["xs_and_us"]))
  assert (np.abs(w["xs_and_us"] - l["xs_and_us"]) / scale).max() < 1e-9
  assert w["kkt"][0, 0] == pytest.approx(l["kkt"][0, 0], rel=1e-9)
  # Hermite-Simpson: the host build of the solver source (tests/hostsim extended by this system, ASan / UBSan clean, scaled variables
  # as on the device) gives cost 2.637376 / 2.6331646 / 2.6311737 and max|c| 6.47651254 / 6.45960817 / 6.44727143 after 0 / 1 / 2 iterations
  hp = HParams(system=SystemType.ROCKETLANDING, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               intervals=20, nlpsolver=NLPSolverType.SQP)
  for mode in ("wave", "lane"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    for it, cost, feas in ((0, 2.637376, 6.47651254), (1, 2.6331646, 6.45960817), (2, 2.6311737, 6.44727143)):
      r = get_optimizer(hp, CFG, hp.system()).solve_batch(max_iter=it)
      assert r["cost"][0] == pytest.approx(cost, rel=2e-7) and r["kkt"][0, 0] == pytest.approx(feas, rel=1e-8), (mode, it)
    res[mode] = get_optimizer(hp, CFG, hp.system()).solve_batch(max_iter=12)
  w, l = res["wave"], res["lane"]
  scale = np.maximum(1.0, np.abs(l["xs_and_us"]))
  assert (np.abs(w["xs_and_us"] - l["xs_and_us"]) / scale).max() < 1e-7


@pytest.mark.parametrize("kw", [
    dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=60),
    dict(optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=60),
    dict(optimizer=OptimizerType.SHOOTING, intervals=6, controls_per_interval=10),
], ids=["trapezoidal", "single-shooting", "multiple-shooting"])
def test_pendulum_swing_up_converges_under_every_transcription(kw):
  """The collocation transcriptions need the elastic phase (or a second start), shooting does not; the discretisations agree
  on the optimum to their order (25.4 .. 25.7)."""
  hp = HParams(system=SystemType.PENDULUM, nlpsolver=NLPSolverType.SQP, **kw)
  opt = get_optimizer(hp, CFG, hp.system())
  r = opt.solve_batch()
  assert r['status'][0] == 0
  assert np.abs(opt.constraints(r['xs_and_us'][0])).max() <= 1e-8
  assert 25.3 < r['cost'][0] < 25.8


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("name,N,lim", [("BEARPOPULATIONS", 30, 300), ("BEARPOPULATIONS", 7, 300), ("ROCKETLANDING", 20, 12), ("ROCKETLANDING", 9, 6),
                                       ("PENDULUM_ELASTIC", 20, 8), ("VANDERPOL_ELASTIC", 20, 300), ("MOUNTAINCAR_ELASTIC", 9, 8),
                                       ("CARTPOLE_ELASTIC", 20, 300), ("CARTPOLE_ELASTIC", 7, 6)])
def test_wider_systems_run_on_the_fused_matrix_core_kernel(monkeypatch, name, N, lim, rule):
  """Round 5: two controls (BEARPOPULATIONS), six states (ROCKETLANDING: its right-hand sides need a second column tile, and its pinned
  terminal states exercise the bookkeeping rows there) and the three-control elastic twins of the two-state systems run on the fused-phase
  kernel with the block form of the matrix-core sweep (hs_solver_fused.h: riccati_mfma_gen) instead of round 2's wavefront kernel and its
  column-per-lane vector sweep.  The library says so (myr_solve_plan), and both kernels walk the same path: same status and iteration
  count, iterates that agree -- to convergence where there is an optimum, over a fixed number of iterations where there is none
  (ROCKETLANDING, and the twins that stop at a cap)."""
  from myriad_amd import _lib
  from oracle import myriad_oracle as O
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  twin = name.endswith("_ELASTIC")
  s = O.Elastic(O.SYSTEMS[name[:-8]](), 1.0) if twin else O.SYSTEMS[name]()
  tr = O.hermite_simpson(s, N) if rule == "HERMITE_SIMPSON" else O.trapezoidal(s, N)
  if twin and rule == "TRAPEZOIDAL": lim = min(lim, 8)      # (the trapezoidal twins stop at their cap on this problem in every kernel: compare the path)
  rng = np.random.default_rng(11)
  B = 6
  z0 = np.tile(tr.guess, (B, 1))
  lb, ub = np.tile(tr.bounds[:, 0], (B, 1)), np.tile(tr.bounds[:, 1], (B, 1))
  x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))     # perturbed start states (pinned through their bounds)
  z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
  res = {}
  for mode in ("wave", "wave1"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    eng = _lib.Engine(name, rule, N, s.T)
    o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
    res[mode] = eng.solve(z0, lb, ub, params=s.params() if twin else None, opts=o)
    res[mode]["plan"] = eng.solve_plan()
    eng.close()
  f, w = res["wave"], res["wave1"]
  if not twin:      # (a twin's solves are the restoration's, not "the" solve of a handle: no plan is recorded for them)
    assert f["plan"]["form"] == "fused" and f["plan"]["waves_per_trajectory"] == 2 and w["plan"]["form"] == "wave"      # (a small batch: two wavefronts per trajectory)
  assert np.array_equal(f["status"], w["status"]) and np.array_equal(f["iters"], w["iters"]), (f["status"], w["status"], f["iters"], w["iters"])
  if lim == 300:
    assert (f["status"] == 0).all(), (f["status"], f["iters"])
  fin = np.isfinite(w["z"])
  assert np.array_equal(np.isfinite(f["z"]), fin)
  d = np.abs(f["z"] - w["z"])[fin] / np.maximum(1.0, np.abs(w["z"])[fin])
  assert d.max(initial=0.0) <= 1e-6, d.max()
  np.testing.assert_allclose(f["cost"], w["cost"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("name", ["BEARPOPULATIONS", "ROCKETLANDING", "PENDULUM_ELASTIC", "CARTPOLE_ELASTIC", "ROCKETLANDING_ELASTIC"])
def test_block_sweep_on_short_horizons(monkeypatch, name, rule):
  """The block sweep prefetches two to four stages ahead and runs its midpoint product one stage early: horizons shorter than the prefetch depth, of
  odd length, of one and two intervals -- the fused kernel against the lane kernel (which shares none of that machinery), three iterations from
  perturbed start states, both schemes.  (Trapezoidal: the reference pins row -nu of the state block, trapezoidal.py:71, so N + 1 > nu.  One interval
  is run for the systems of moderate scale only: on ROCKETLANDING at N = 1 -- states of 1e3, inertia corrections from the first iteration -- the three
  kernels differ by 3e-5 after ONE iteration, the fused one and round 2's together against the lane kernel, tools/dev/exp/exp70.py.)"""
  from myriad_amd import _lib
  from oracle import myriad_oracle as O
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  twin = name.endswith("_ELASTIC")
  s = O.Elastic(O.SYSTEMS[name[:-8]](), 1.0) if twin else O.SYSTEMS[name]()
  for N in (1, 2, 3, 4, 5, 8, 11):
    if (rule == "TRAPEZOIDAL" and N + 1 <= s.nu) or (N == 1 and "ROCKETLANDING" in name):
      continue
    tr = O.hermite_simpson(s, N) if rule == "HERMITE_SIMPSON" else O.trapezoidal(s, N)
    rng = np.random.default_rng(100 + N)
    B = 3
    z0 = np.tile(tr.guess, (B, 1))
    lb, ub = np.tile(tr.bounds[:, 0], (B, 1)), np.tile(tr.bounds[:, 1], (B, 1))
    x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))
    z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
    res = {}
    for mode in ("wave", "lane"):
      monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
      eng = _lib.Engine(name, rule, N, s.T)
      o = eng.default_opts(); o.restoration = 0; o.max_iter = 3
      res[mode] = eng.solve(z0, lb, ub, params=s.params() if twin else None, opts=o)
      eng.close()
    f, l = res["wave"], res["lane"]
    assert np.array_equal(f["status"], l["status"]) and np.array_equal(f["iters"], l["iters"]), (N, f["status"], l["status"], f["iters"], l["iters"])
    fin = np.isfinite(l["z"])
    assert np.array_equal(np.isfinite(f["z"]), fin), N
    d = np.abs(f["z"] - l["z"])[fin] / np.maximum(1.0, np.abs(l["z"])[fin])
    assert d.max(initial=0.0) <= 2e-6, (N, d.max())
    np.testing.assert_allclose(f["cost"], l["cost"], rtol=2e-8, atol=1e-12)


def _wide_problem(name, rule, N, B, seed):
  from oracle import myriad_oracle as O
  twin = name.endswith("_ELASTIC")
  s = O.Elastic(O.SYSTEMS[name[:-8]](), 1.0) if twin else O.SYSTEMS[name]()
  tr = O.hermite_simpson(s, N) if rule == "HERMITE_SIMPSON" else O.trapezoidal(s, N)
  rng = np.random.default_rng(seed)
  z0 = np.tile(tr.guess, (B, 1))
  lb, ub = np.tile(tr.bounds[:, 0], (B, 1)), np.tile(tr.bounds[:, 1], (B, 1))
  x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))
  z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
  return s, z0, lb, ub, (s.params() if twin else None)


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("name", ["ROCKETLANDING", "CARTPOLE_ELASTIC", "ROCKETLANDING_ELASTIC"])
def test_wide_stages_run_their_forward_recursion_over_stored_maps(monkeypatch, name, rule):
  """Round 6: stages of seven or more knot variables (ROCKETLANDING: 8, CARTPOLE's twin: 9, ROCKETLANDING's twin: 14) run the forward phase's recursion
  s_{k+1} = A_k s_k + b_k sequentially over maps stored in global scratch (hs_solver_fused.h: fwd_seq) instead of a wave scan over NW x NW affine maps in
  registers.  Horizons on either side of the 64-stage block of a wavefront (one round, a round that ends on the block edge, two rounds, three for the
  one-wavefront form), both wavefront forms, against the lane kernel, which shares none of that machinery: same statuses and iteration counts after three
  iterations, iterates within 2e-6."""
  from myriad_amd import _lib
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  for N in (63, 64, 65, 100, 130):
    s, z0, lb, ub, prm = _wide_problem(name, rule, N, 3, 200 + N)
    res = {}
    for form, env in (("w1", {"MYRIAD_SOLVE_MODE": "wave", "MYRIAD_FUSED_WAVES": "1"}), ("w2", {"MYRIAD_SOLVE_MODE": "wave", "MYRIAD_FUSED_WAVES": "2"}),
                      ("lane", {"MYRIAD_SOLVE_MODE": "lane", "MYRIAD_FUSED_WAVES": "0"})):
      for k, v in env.items(): monkeypatch.setenv(k, v)
      eng = _lib.Engine(name, rule, N, s.T)
      o = eng.default_opts(); o.restoration = 0; o.max_iter = 3
      res[form] = eng.solve(z0, lb, ub, params=prm, opts=o)
      eng.close()
    l = res["lane"]
    fin = np.isfinite(l["z"])
    for form in ("w1", "w2"):
      f = res[form]
      assert np.array_equal(f["status"], l["status"]) and np.array_equal(f["iters"], l["iters"]), (form, N, f["status"], l["status"], f["iters"], l["iters"])
      assert np.array_equal(np.isfinite(f["z"]), fin), (form, N)
      d = np.abs(f["z"] - l["z"])[fin] / np.maximum(1.0, np.abs(l["z"])[fin])
      assert d.max(initial=0.0) <= 2e-6, (form, N, d.max())


@pytest.mark.parametrize("name,rule,waves", [("ROCKETLANDING", "HERMITE_SIMPSON", 1), ("ROCKETLANDING_ELASTIC", "HERMITE_SIMPSON", 2),
                                             ("ROCKETLANDING_ELASTIC", "TRAPEZOIDAL", 1), ("CARTPOLE_ELASTIC", "HERMITE_SIMPSON", 1)])
def test_wide_systems_fill_the_cu_at_large_batch_sizes(monkeypatch, name, rule, waves):
  """Round 6: the wide systems keep the bound multipliers in global scratch where the solver's LDS would otherwise let two workgroups onto a CU (four
  one-wavefront workgroups then fit: ROCKETLANDING, CARTPOLE's twin, the trapezoidal form of ROCKETLANDING's twin), and where even then only two fit
  (ROCKETLANDING's twin under Hermite-Simpson) two wavefronts share a trajectory at EVERY batch size.  A batch of more trajectories than slots -- every
  slot takes several, one after the other, inheriting LDS and scratch -- of three distinct problems repeated: every copy returns the bits of the three
  solved alone, with and without poison in what a trajectory inherits."""
  from myriad_amd import _lib
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  monkeypatch.delenv("MYRIAD_FUSED_WAVES", raising=False); monkeypatch.delenv("MYRIAD_SOLVE_MODE", raising=False)
  N, B = 100, 1100
  s, z3, lb3, ub3, prm = _wide_problem(name, rule, N, 3, 77)
  rep = np.arange(B) % 3
  out = {}
  for tag, poison, zz, ll, uu in (("alone", None, z3, lb3, ub3), ("batch", None, z3[rep], lb3[rep], ub3[rep]), ("poison", "nan", z3[rep], lb3[rep], ub3[rep])):
    if poison: monkeypatch.setenv("MYRIAD_POISON", poison)
    else: monkeypatch.delenv("MYRIAD_POISON", raising=False)
    if tag == "alone": monkeypatch.setenv("MYRIAD_FUSED_WAVES", str(waves))      # (three trajectories would take the two-wavefront form: other sums, other bits)
    else: monkeypatch.delenv("MYRIAD_FUSED_WAVES", raising=False)
    eng = _lib.Engine(name, rule, N, s.T, max_batch=B)
    o = eng.default_opts(); o.restoration = 0; o.max_iter = 4
    out[tag] = eng.solve(zz, ll, uu, params=prm, opts=o)
    out[tag]["plan"] = eng.solve_plan()
    eng.close()
  if not name.endswith("_ELASTIC"):
    assert out["batch"]["plan"]["form"] == "fused" and out["batch"]["plan"]["waves_per_trajectory"] == waves, out["batch"]["plan"]
  a = out["alone"]
  for tag in ("batch", "poison"):
    r = out[tag]
    for k in ("z", "lam", "cost", "status", "iters"):
      assert np.array_equal(np.asarray(r[k]), np.asarray(a[k])[rep], equal_nan=True) if np.asarray(r[k]).dtype.kind == "f" else np.array_equal(np.asarray(r[k]), np.asarray(a[k])[rep]), (tag, k)


@pytest.mark.parametrize("N,waves", [(150, 2), (200, 1), (300, 2)])
def test_long_horizons_take_two_wavefronts_per_trajectory_where_as_many_workgroups_fit(monkeypatch, N, waves):
  """Round 6: the rule of the wide systems holds for every system once the horizon is long enough.  CARTPOLE, Hermite-Simpson: at N = 150 the solver's LDS lets
  two one-wavefront workgroups onto a CU and two two-wavefront ones as well -- two wavefronts per trajectory at every batch size (B = 4096: 34.1 -> 27.2 ms);
  at N = 200 two against one -- the one-wavefront form stays; at N = 300 one against one (130.9 -> 88.2 ms).  A batch beyond two trajectories per CU: the plan
  says which form ran, every trajectory converges, and the copies of three problems return the bits of the three solved alone in the same form, with and
  without poison in what a slot's next trajectory inherits (the two-wavefront form with the two-level sweep serves several trajectories per slot here)."""
  from myriad_amd import _lib
  from bench import build_workload
  for k in ("MYRIAD_FUSED_WAVES", "MYRIAD_SOLVE_MODE", "MYRIAD_PARK_ITER"): monkeypatch.delenv(k, raising=False)
  B = 600
  x0, z3, lb3, ub3, T = build_workload(3, N, 5)
  rep = np.arange(B) % 3
  out = {}
  for tag, poison, zz, ll, uu in (("alone", None, z3, lb3, ub3), ("batch", None, z3[rep], lb3[rep], ub3[rep]), ("poison", "random", z3[rep], lb3[rep], ub3[rep])):
    if poison: monkeypatch.setenv("MYRIAD_POISON", poison)
    else: monkeypatch.delenv("MYRIAD_POISON", raising=False)
    if tag == "alone": monkeypatch.setenv("MYRIAD_FUSED_WAVES", str(waves)); monkeypatch.setenv("MYRIAD_PARK_ITER", "0")
    else: monkeypatch.delenv("MYRIAD_FUSED_WAVES", raising=False); monkeypatch.delenv("MYRIAD_PARK_ITER", raising=False)
    eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
    out[tag] = eng.solve(zz, ll, uu)
    out[tag]["plan"] = eng.solve_plan()
    eng.close()
  assert out["batch"]["plan"]["form"] == "fused" and out["batch"]["plan"]["waves_per_trajectory"] == waves, out["batch"]["plan"]
  a = out["alone"]
  assert (a["status"] == 0).all()
  for tag in ("batch", "poison"):
    r = out[tag]
    for k in ("z", "lam", "cost"):
      assert np.array_equal(np.asarray(r[k]), np.asarray(a[k])[rep]), (tag, k)
    assert np.array_equal(r["status"], a["status"][rep]) and np.array_equal(r["iters"], a["iters"][rep]), tag


@pytest.mark.parametrize("name,rule", [("ROCKETLANDING", "HERMITE_SIMPSON"), ("CARTPOLE_ELASTIC", "HERMITE_SIMPSON"), ("ROCKETLANDING_ELASTIC", "TRAPEZOIDAL")])
def test_forms_with_the_bound_multipliers_in_scratch_park_and_resume(monkeypatch, name, rule):
  """The two-phase launch of the one-wavefront forms that keep zL, zU in the slot's global scratch (round 6): the parked record carries them beside the
  solver's LDS; parked after 1, 3 and 7 iterations, resumed in another slot, under poison, a trajectory returns the bits of a whole solve."""
  from myriad_amd import _lib
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  monkeypatch.setenv("MYRIAD_FUSED_WAVES", "1"); monkeypatch.delenv("MYRIAD_SOLVE_MODE", raising=False)
  N, B = 20, 7
  s, z0, lb, ub, prm = _wide_problem(name, rule, N, B, 31)
  def run(park, poison):
    monkeypatch.setenv("MYRIAD_PARK_ITER", str(park))
    if poison: monkeypatch.setenv("MYRIAD_POISON", poison)
    else: monkeypatch.delenv("MYRIAD_POISON", raising=False)
    eng = _lib.Engine(name, rule, N, s.T)
    o = eng.default_opts(); o.restoration = 0; o.max_iter = 12
    r = eng.solve(z0, lb, ub, params=prm, opts=o)
    r["plan"] = eng.solve_plan()
    eng.close()
    return r
  ref = run(0, None)
  for park, poison in ((1, None), (3, "nan"), (7, "random")):
    r = run(park, poison)
    if not name.endswith("_ELASTIC"):
      assert r["plan"]["park_iter"] == park and r["plan"]["launches_per_solve"] == 2, r["plan"]
    for k in ("z", "lam", "cost", "status", "iters"):
      assert np.array_equal(np.asarray(r[k]), np.asarray(ref[k]), equal_nan=(np.asarray(r[k]).dtype.kind == "f")), (park, poison, k)
