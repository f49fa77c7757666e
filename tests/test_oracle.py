"""CPU tests of the oracle itself (no GPU): the reference's one known-answer test, the survey's
App. C numbers, and the committed golden vectors."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import myriad_oracle as O


def test_reference_kat_rk4_integrate():
  """/root/reference/tests/tests.py:19-43: RK4 `integrate` of y'=y, y(0)=1, 99 steps, equals odeint (=e) to
  6 decimals.  (The test passes a length-100 control array and relies on index clamping, quirk Q6.)"""
  from scipy.integrate import odeint
  res = odeint(lambda t, y: y, np.array([1.]), [0., 1.], tfirst=True)
  N = 100
  t = torch.linspace(0., 1., N, dtype=torch.float64)
  h = t[1]
  _, states = O.integrate(lambda x, u, tt: x, torch.tensor([1.], dtype=torch.float64), t, h, N - 1, t, "RK4")
  np.testing.assert_almost_equal(res[-1], states[-1].numpy(), decimal=6)


@pytest.mark.parametrize("method,order", [("EULER", 1), ("HEUN", 2), ("MIDPOINT", 1), ("RK4", 4)])
def test_integrator_orders(method, order):
  """Convergence order of each step rule in utils.py:31-54 on y'=-y.  The reference's "midpoint" rule takes a
  FULL Euler step to its "mid" state (utils.py:47-50: x_mid = x + h f(x,u1)), so it is first order (quirk Q11)."""
  errs = []
  for n in (20, 40):
    mc = 2 if method == "RK4" else 1
    us = torch.zeros(mc * n + 1, 1, dtype=torch.float64)
    xT, _ = O.integrate_time_independent(lambda x, u: -x, torch.tensor([1.], dtype=torch.float64), us, 1.0 / n, n, method)
    errs.append(abs(float(xT[0]) - np.exp(-1.0)))
  assert np.log2(errs[0] / errs[1]) == pytest.approx(order, abs=0.25)


def test_shapes_match_survey_appendix_a():
  """SURVEY.md App. A.5 sizes (n, m, pinned variables) for the BASELINE configs."""
  cases = [("SIMPLECASE", "SHOOTING", dict(intervals=10, controls_per_interval=100), 1012, 10, 1),
           ("CARTPOLE", "HS", dict(intervals=100), 1005, 800, 8),
           ("VANDERPOL", "SHOOTING", dict(intervals=1, controls_per_interval=50), 55, 2, 4),
           ("CANCERTREATMENT", "SHOOTING", dict(intervals=1, controls_per_interval=100), 103, 1, 1),
           ("CARTPOLE", "TRAP", dict(intervals=100), 505, 400, 8)]
  for name, kind, kw, n, m, pinned in cases:
    s = O.SYSTEMS[name]()
    tr = {"HS": O.hermite_simpson, "TRAP": O.trapezoidal, "SHOOTING": O.shooting}[kind](s, **kw)
    assert tr.guess.shape == (n,)
    assert tr.bounds.shape == (n, 2)
    assert tr.constraints(torch.as_tensor(tr.guess)).shape == (m,)
    assert int((tr.bounds[:, 0] == tr.bounds[:, 1]).sum()) == pinned


def test_guesses_match_survey_appendix_c():
  s = O.CancerTreatment(); tr = O.shooting(s, 1, 100)
  np.testing.assert_allclose(tr.x_guess.ravel(), [0.975, 0.6579], atol=5e-5)       # one Heun step of size 20
  s = O.SimpleCase(); tr = O.shooting(s, 10, 100)
  np.testing.assert_allclose(tr.x_guess.ravel()[:3], [1., 0.9524, 0.9092], atol=5e-5)
  s = O.VanDerPol(); tr = O.shooting(s, 1, 50)
  np.testing.assert_allclose(tr.x_guess, [[0., 1.], [0., 0.]])


def test_slsqp_path_reproduces_survey_costs():
  """The SciPy SLSQP branch (nlp_solvers/__init__.py:50-52) on the restated callbacks reproduces the
  independently-measured survey numbers."""
  s = O.CancerTreatment(); tr = O.shooting(s, 1, 100)
  r = O.solve(tr, "SLSQP", max_iter=500)
  assert r["scipy"].success and r["cost"] == pytest.approx(20.5735535185, rel=1e-9)
  s = O.VanDerPol(); tr = O.shooting(s, 1, 50)
  r = O.solve(tr, "SLSQP")
  assert r["scipy"].success and r["cost"] == pytest.approx(2.8731963348, rel=1e-8)


def test_golden_eval_vectors_match_oracle(golden_dir):
  files = sorted(glob.glob(os.path.join(golden_dir, "eval_hs_*.npz")))
  assert files
  for path in files:
    name = os.path.basename(path).split("_")[2].upper()
    if "N100" in path:
      continue   # big one is exercised on the GPU side
    d = np.load(path)
    s = O.SYSTEMS[name]()
    N = int(d["N"])
    tr = O.hermite_simpson(s, N)
    cb = O.Callbacks(tr)
    for b in range(d["z"].shape[0]):
      np.testing.assert_allclose(cb.cons(d["z"][b]), d["c"][b], rtol=0, atol=1e-14)
      np.testing.assert_allclose(cb.fun(d["z"][b]), d["f"][b], rtol=1e-14)
      J = O.hs_dense_from_blocks(d["jblk"][b], N, s.ns, s.nu)
      np.testing.assert_allclose(cb.jac(d["z"][b]), J, rtol=0, atol=1e-14)


def test_rollout_and_defect_helpers():
  """utils.py:258-324 restatement: cost of zero control on SIMPLECASE equals the integral of -x."""
  s = O.SimpleCase()
  xs, c = O.get_state_trajectory_and_cost(s, 200, "RK4", s.x_0, np.zeros((401, 1)))
  # x' = -x^2/2, x(0)=1 -> x(t) = 2/(t+2); integral of -x over [0,1] = -2 ln(3/2)
  assert xs[-1, 0] == pytest.approx(2.0 / 3.0, rel=1e-9)
  assert c == pytest.approx(-2.0 * np.log(1.5), rel=1e-9)
  assert O.get_defect(s, xs) is None
  cp = O.CartPole()
  np.testing.assert_allclose(O.get_defect(cp, np.array([[1., np.pi, 0.5, 0.]])), [0., 0., 0.5, 0.])


def test_lagrangian_restatement_is_consistent():
  """extra_gradient.py:21-33: grad_x L = grad f + J^T lam, grad_lam L = c, J v by forward mode, and one `step`."""
  s = O.VanDerPol()
  for tr in (O.hermite_simpson(s, 5), O.trapezoidal(s, 5)):
    L, cb = O.Lagrangian(tr), O.Callbacks(tr)
    rng = np.random.default_rng(1)
    z = tr.guess + 0.1 * rng.standard_normal(tr.guess.size)
    lam = rng.standard_normal(cb.cons(z).size); v = rng.standard_normal(z.size)
    np.testing.assert_allclose(L.grad_x(z, lam), cb.grad(z) + cb.jac(z).T @ lam, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(L.grad_lmbda(z, lam), cb.cons(z), rtol=0, atol=0)
    np.testing.assert_allclose(L.jvp(z, v), cb.jac(z) @ v, rtol=1e-12, atol=1e-13)
    eps = 1e-6
    fd = (L.value(z + eps * v, lam) - L.value(z - eps * v, lam)) / (2 * eps)
    assert abs(fd - L.grad_x(z, lam) @ v) < 1e-6 * max(1.0, abs(fd))
    x1, l1 = L.step(z, lam, 1e-2, 1e-3)
    lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
    assert (x1 >= lb).all() and (x1 <= ub).all()
    np.testing.assert_allclose(l1, lam + 1e-3 * cb.cons(x1), rtol=1e-14)


def test_fbsm_restatement_is_a_pontryagin_fixed_point():
  """forward_backward_sweep.py:88-116: at the returned iterate the control equals its optimality characterisation
  (to the sweep's own 1e-3 stopping tolerance) and the state/adjoint satisfy their ODEs (RK4 residual)."""
  for S in (O.SimpleCase, O.CancerTreatment):
    s = S()
    r = O.fbsm(s, 400)
    assert 2 <= r["sweeps"] < 50
    u_star = s.optim_characterization(r["adj"], r["x"])
    assert np.abs(u_star - r["u"]).max() < 2e-2 * max(1.0, np.abs(r["u"]).max())
    h = s.T / 400
    xdot = (r["x"][2:] - r["x"][:-2]) / (2 * h)
    assert np.abs(xdot - s.np_dynamics(r["x"][1:-1], r["u"][1:-1])).max() < 5e-2 * max(1.0, np.abs(xdot).max())
    adot = (r["adj"][2:] - r["adj"][:-2]) / (2 * h)
    assert np.abs(adot - s.adj_ODE(r["adj"][1:-1], r["x"][1:-1], r["u"][1:-1])).max() < 5e-2 * max(1.0, np.abs(adot).max())
    assert r["adj"][-1, 0] == 0.0                                # transversality: adj(T) = adj_T = 0


def test_discrete_fbsm_restatement_first_sweep_by_hand():
  """Discrete variant (forward_backward_sweep.py:33-41, utils.py:184-188) on INVASIVEPLANT with T = 2: the first sweep
  written out focus by focus -- the forward recurrence with u = 0, the backward one pairing x_i with u_{i-1} (the
  reference's indexing), the shifted characterisation (invasive_plant.py:86-88) and the halving update."""
  B, k, eps = 7.0, 1.3, 0.02
  s = O.InvasivePlant(B=B, k=k, eps=eps, x_0=(.5, 1., 1.5, 2., 10.), T=2.)
  r = O.fbsm(s, max_sweeps=1)
  assert r["sweeps"] == 1 and r["x"].shape == (3, 5) and r["u"].shape == (2, 5) and r["adj"].shape == (3, 5)
  g = lambda x: x + x * k / (eps + x)
  for j, rho in enumerate((.5, 1., 1.5, 2., 10.)):
    x1 = g(rho); x2 = g(x1)
    a2 = 1.0
    a1 = a2 * (1 - 0.0) * (1 + eps * k / (eps + x2) ** 2)
    a0 = a1 * (1 - 0.0) * (1 + eps * k / (eps + x1) ** 2)
    u0 = 0.5 * min(1.0, max(0.0, 0.5 * a1 / B * g(rho)))
    u1 = 0.5 * min(1.0, max(0.0, 0.5 * a2 / B * x2))           # g(x1) = x2
    np.testing.assert_allclose(r["x"][:, j], [rho, x1, x2], rtol=1e-15)
    np.testing.assert_allclose(r["adj"][:, j], [a0, a1, a2], rtol=1e-15)
    np.testing.assert_allclose(r["u"][:, j], [u0, u1], rtol=1e-15)
  # default weights: the fixed point removes every focus in the last step (u = 1, x_T = 0, cost = 5 B)
  r = O.fbsm(O.InvasivePlant())
  assert r["sweeps"] < 100
  np.testing.assert_allclose(r["u"][-1], 1.0, rtol=0, atol=1e-6)  # u <- (1 + u) / 2: 1 - 2^-sweeps
  np.testing.assert_allclose(r["x"][-1], 0.0, rtol=0, atol=1e-4)


def test_golden_shooting_solutions_are_feasible_optima_of_the_oracle_problem(golden_dir):
  """tests/golden/solve_shoot_*.npz: recomputed feasibility / cost, bounds, and first-order optimality in the
  null space of the constraints at the stored optimum (projected gradient on the inactive variables)."""
  cases = {"simplecase_10x100": O.SimpleCase, "vanderpol_1x50": O.VanDerPol, "cancertreatment_1x100": O.CancerTreatment}
  for tag, S in cases.items():
    d = np.load(os.path.join(golden_dir, f"solve_shoot_{tag}.npz"))
    for b in range(d["z"].shape[0]):
      kw = dict(zip(S.param_names, d["params"][b])) if S is O.CancerTreatment else {}
      s = S(**kw) if kw else S()
      s.x_0 = d["x0"][b].copy()
      tr = O.shooting(s, int(d["intervals"]), int(d["cpi"]), "HEUN")
      cb = O.Callbacks(tr)
      z = d["z"][b]
      np.testing.assert_array_equal(tr.bounds[:, 0], d["lb"][b])
      assert np.abs(cb.cons(z)).max() <= 1e-10
      assert cb.fun(z) == pytest.approx(float(d["cost"][b]), rel=1e-13)
      assert (z >= d["lb"][b] - 1e-12).all() and (z <= d["ub"][b] + 1e-12).all()
      if b == 0 and tag != "simplecase_10x100":
        J, g = cb.jac(z), cb.grad(z)
        free = (d["lb"][b] < d["ub"][b]) & (z - d["lb"][b] > 1e-6) & (d["ub"][b] - z > 1e-6)
        lam = np.linalg.lstsq(J[:, free].T, -g[free], rcond=None)[0]
        assert np.abs(g[free] + J[:, free].T @ lam).max() < 1e-5 * max(1.0, np.abs(g).max())
