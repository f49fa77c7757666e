"""GPU tests of the batched Forward-Backward Sweep: parity with the oracle's restatement of
forward_backward_sweep.py / integrate_fbsm, and the independent cross-check it provides for the direct solver
(continuous-time Pontryagin solution vs the SQP solution of the transcribed problem)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from myriad_amd.config import Config, HParams, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer

CFG = Config(verbose=False, plot=False)


@pytest.mark.parametrize("st", [SystemType.SIMPLECASE, SystemType.CANCERTREATMENT])
def test_fbsm_matches_oracle_restatement(st):
  from oracle import myriad_oracle as O
  hp = HParams(system=st, optimizer=OptimizerType.FBSM, fbsm_intervals=1000)
  opt = get_optimizer(hp, CFG, hp.system())
  sol = opt.solve()
  ref = O.fbsm({SystemType.SIMPLECASE: O.SimpleCase, SystemType.CANCERTREATMENT: O.CancerTreatment}[st](), 1000)
  assert set(sol) == {'x', 'u', 'adj'}
  assert sol['x'].shape == (1001, 1) and sol['u'].shape == (1001, 1) and sol['adj'].shape == (1001, 1)
  assert int(opt.solve_batch()['sweeps'][0]) == ref['sweeps']
  for k in ('x', 'u', 'adj'):
    np.testing.assert_allclose(sol[k], ref[k], rtol=1e-11, atol=1e-12, err_msg=k)


ALL12 = ["SIMPLECASE", "CANCERTREATMENT", "BACTERIA", "BEARPOPULATIONS", "BIOREACTOR", "EPIDEMICSEIRN", "GLUCOSE", "HARVEST", "HIVTREATMENT",
         "MOULDFUNGICIDE", "SIMPLECASEWITHBOUNDS", "TIMBERHARVEST"]


@pytest.mark.parametrize("name", ALL12)
def test_fbsm_all_indirect_systems_match_oracle(name):
  """adj_ODE / optim_characterization of every continuous IndirectFHCS system without terminal state conditions (device:
  csrc/fbsm.h; oracle: numpy restatement of the same reference lines), incl. adj_T (BACTERIA), two controls
  (BEARPOPULATIONS), explicit time (HARVEST, TIMBERHARVEST), bang-bang characterisations (BIOREACTOR, TIMBERHARVEST) and
  the unclipped one (GLUCOSE)."""
  from oracle import myriad_oracle as O
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.FBSM, fbsm_intervals=200)
  opt = get_optimizer(hp, CFG, hp.system())
  r = opt.solve_batch(max_sweeps=300)
  ref = O.fbsm(O.SYSTEMS[name](), 200, max_sweeps=300)
  assert int(r['sweeps'][0]) == ref['sweeps']
  for k in ('x', 'u', 'adj'):
    sc = max(1.0, np.abs(ref[k]).max())
    np.testing.assert_allclose(r[k][0], ref[k], rtol=1e-9, atol=1e-11 * sc, err_msg=k)


def test_fbsm_parameter_sweep_matches_per_instance_oracle():
  from oracle import myriad_oracle as O
  hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.FBSM, fbsm_intervals=200)
  opt = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(2019)
  B = 70                                                          # more than one wavefront
  P = np.stack([rng.uniform(0.1, 0.5, B), rng.uniform(1, 5, B), rng.uniform(0.2, 0.8, B)], 1)   # r, a, delta
  x0 = rng.uniform(0.5, 0.99, (B, 1))
  r = opt.solve_batch(x0s=x0, params=P)
  for b in (0, 13, 63, 64, 69):
    ref = O.fbsm(O.CancerTreatment(r=P[b, 0], a=P[b, 1], delta=P[b, 2], x_0=x0[b, 0]), 200)
    assert int(r['sweeps'][b]) == ref['sweeps']
    np.testing.assert_allclose(r['u'][b], ref['u'], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(r['adj'][b], ref['adj'], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("st", [SystemType.SIMPLECASE, SystemType.CANCERTREATMENT])
def test_direct_sqp_solution_agrees_with_pontryagin_solution(st):
  """Independent check of the direct path: the HS-collocation SQP optimum and the FBSM fixed point solve the same
  continuous problem; they agree to discretisation error (and the stopping tolerance 1e-3 of the sweep)."""
  N = 50
  hp = HParams(system=st, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=N)
  direct = get_optimizer(hp, CFG, hp.system()).solve()
  hp2 = HParams(system=st, optimizer=OptimizerType.FBSM, fbsm_intervals=1000)
  ind = get_optimizer(hp2, CFG, hp2.system()).solve()
  T = hp.system().T
  tk = np.linspace(0, T, 2 * N + 1); tf = np.linspace(0, T, 1001)
  xf = np.interp(tk, tf, ind['x'][:, 0]); uf = np.interp(tk, tf, ind['u'][:, 0])
  assert np.abs(direct['x'][:, 0] - xf).max() < 5e-3 * max(1.0, np.abs(xf).max())
  assert np.abs(direct['u'][:, 0] - uf)[1:-1].max() < 2e-2 * max(1.0, np.abs(uf).max())
  # costs: Simpson quadrature of the running cost along the FBSM trajectory vs the NLP objective
  s = hp.system()
  g = np.array([s.cost(ind['x'][i], ind['u'][i]) for i in range(1001)])
  h = T / 1000
  c_ind = h / 3 * (g[0] + g[-1] + 4 * g[1:-1:2].sum() + 2 * g[2:-1:2].sum())
  assert abs(direct['cost'] - c_ind) < 2e-3 * max(1.0, abs(c_ind))


def test_fbsm_rejects_systems_without_adjoint():
  for st in (SystemType.CARTPOLE, SystemType.SEIR):               # no adjoint dynamics
    hp = HParams(system=st, optimizer=OptimizerType.FBSM)
    with pytest.raises(NotImplementedError):
      get_optimizer(hp, CFG, hp.system())


def test_fbsm_secant_solver_for_a_terminal_state_condition():
  """FBSM.sequencesolver (forward_backward_sweep.py:118-158) on PREDATORPREY: secant iterations on adj(T) of the pinned
  state, each one a device sweep sequence; same iterates as the oracle's restatement."""
  from oracle import myriad_oracle as O
  hp = HParams(system=SystemType.PREDATORPREY, optimizer=OptimizerType.FBSM, fbsm_intervals=200)
  opt = get_optimizer(hp, CFG, hp.system())
  sol = opt.solve()
  ref = O.fbsm_secant(O.PredatorPrey(), 200)
  assert opt.secant_iterations == ref["secant_iterations"]
  assert abs(sol['x'][-1, 2] - 5.0) <= 1e-10
  for k in ('x', 'u', 'adj'):
    np.testing.assert_allclose(sol[k], ref[k], rtol=1e-8, atol=1e-10, err_msg=k)


def test_discrete_fbsm_invasive_plant_matches_oracle():
  """The discrete variant (forward_backward_sweep.py:33-41, utils.py:184-188) on INVASIVEPLANT: N = int(T) unit steps,
  u with one row per step, the reference's index pairing in the backward recurrence; default parameters (B = 1: the
  fixed point removes everything in the last step) and weights with interior controls."""
  from oracle import myriad_oracle as O
  hp = HParams(system=SystemType.INVASIVEPLANT, optimizer=OptimizerType.FBSM, fbsm_intervals=1000)   # fbsm_intervals is ignored (:33-35)
  opt = get_optimizer(hp, CFG, hp.system())
  assert opt.N == 10 and opt.h == 1 and opt.u_guess.shape == (10, 5) and opt.x_guess.shape == (11, 5)
  sol = opt.solve()
  ref = O.fbsm(O.InvasivePlant())
  assert sol['x'].shape == (11, 5) and sol['u'].shape == (10, 5) and sol['adj'].shape == (11, 5)
  for k in ('x', 'u', 'adj'):
    np.testing.assert_allclose(sol[k], ref[k], rtol=1e-12, atol=1e-13, err_msg=k)
  np.testing.assert_allclose(sol['u'][-1], 1.0, rtol=0, atol=1e-6)   # eradication in the last step (u <- (1 + u) / 2)
  # parameter and start-state sweep, more than one wavefront
  rng = np.random.default_rng(5)
  B = 130
  P = np.stack([rng.uniform(2.0, 100.0, B), rng.uniform(0.5, 1.5, B), rng.uniform(0.005, 0.05, B)], 1)   # B, k, eps
  x0 = rng.uniform(0.2, 10.0, (B, 5))
  r = opt.solve_batch(x0s=x0, params=P)
  assert r['u'].shape == (B, 10, 5)
  assert np.all((r['u'] >= 0.0) & (r['u'] <= 1.0))
  for b in (0, 31, 63, 64, 127, 128, 129):
    ref = O.fbsm(O.InvasivePlant(B=P[b, 0], k=P[b, 1], eps=P[b, 2], x_0=x0[b]))
    assert int(r['sweeps'][b]) == ref['sweeps']
    for k in ('x', 'u', 'adj'):
      np.testing.assert_allclose(r[k][b], ref[k], rtol=1e-11, atol=1e-12, err_msg=f"{k}[{b}]")


def test_discrete_system_only_has_the_fbsm_entry_point():
  from myriad_amd import _lib
  e = _lib.Engine("INVASIVEPLANT", "HERMITE_SIMPSON", 4, 10.0, max_batch=1)
  assert (e.ns, e.nu, e.np) == (5, 5, 3)
  with pytest.raises(NotImplementedError, match="discrete-time"):   # MYR_E_UNSUPPORTED
    e.eval(np.zeros((1, e.n)))
  with pytest.raises(NotImplementedError, match="discrete-time"):
    e.rollout(np.zeros((1, 5)), np.zeros((1, 5, 5)), 4)
  e.close()
