#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: fixtures computed by THE REFERENCE'S OWN LINES (round 5).

Runs in the build container only (reads /root/reference in place, copies none of it, writes no bytecode): installs the name-forwarding stand-in
for the small JAX surface of the hot-path files (tests/golden/refshim), imports the reference's `myriad.trajectory_optimizers.get_optimizer`
and `myriad.utils`, and records, for every system x {Hermite-Simpson, trapezoidal, shooting x Euler / Heun / midpoint / RK4}:

  guess, bounds                                the reference's initial point and box (hermite_simpson.py:37-81, trapezoidal.py:33-77, shooting.py:47-77,247-275)
  z = guess + seeded noise (clipped away from nothing: objective / constraints are evaluated wherever z is)
  objective(z), constraints(z)                 the callbacks nlp_solvers/__init__.py:32-40 hands to the NLP code
  get_state_trajectory_and_cost(...)           utils.py:258-298 on the controls of z (RK4, and the transcription's own rule)
  integrate(...) on tests/tests.py:19-43's setting is already carried in tests/test_oracle.py

-> tests/golden/reference_callbacks.npz.  tests/test_reference_fixtures.py (CPU suite) holds the ORACLE to these numbers at 1e-13.
The fixtures are data (inputs and outputs); nothing of the reference's text is stored.

What this is not: "the reference run here" in the sense of the parity rules (jax itself is absent; a stand-in for a library the image lacks does
not count), so DESIGN.md keeps "parity unpinned at the JAX / IPOPT boundary".  It replaces "read by eye" with "computed by the reference's lines".
  usage: PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_fixtures.py"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refshim  # noqa: E402

refshim.install()
sys.path.insert(0, "/root/reference")
import numpy as np  # noqa: E402

from myriad.config import Config, HParams, IntegrationMethod, OptimizerType, QuadratureRule  # noqa: E402
from myriad.systems import SystemType  # noqa: E402
from myriad.trajectory_optimizers import get_optimizer  # noqa: E402
from myriad.utils import get_state_trajectory_and_cost  # noqa: E402

CFG = Config(verbose=False, plot=False, jit=False)
SKIP = {"INVASIVEPLANT"}      # discrete time: the reference's direct optimisers refuse it (base.py:66-67)
CASES = [("HERMITE_SIMPSON", None, 6, 1), ("TRAPEZOIDAL", None, 7, 1),
         ("SHOOTING", "EULER", 3, 4), ("SHOOTING", "HEUN", 3, 4), ("SHOOTING", "MIDPOINT", 2, 5), ("SHOOTING", "RK4", 2, 3),
         ("SHOOTING", "HEUN", 1, 6)]


def main():
  out = {}
  log = []
  rng = np.random.default_rng(5)
  # `--solve-only NAME`: only the reference's solve() runs, into tests/golden/NAME.npz -- used with /opt/conda/bin/python3.9 (SciPy 1.7.1: the
  # reference pins scipy==1.7.0, requirements.txt) beside the default interpreter's SciPy 1.15
  solve_only = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--solve-only" else None
  # the BASELINE configurations at their full sizes (BASELINE.json configs 1-4 and README.md:83's literal; config 5's network needs haiku)
  FULL = {"CARTPOLE": [("HERMITE_SIMPSON", None, 100, 1), ("TRAPEZOIDAL", None, 100, 1)], "VANDERPOL": [("SHOOTING", "HEUN", 1, 50)],
          "CANCERTREATMENT": [("SHOOTING", "HEUN", 1, 100)], "SIMPLECASE": [("SHOOTING", "HEUN", 10, 100)]}
  for st in ([] if solve_only else SystemType):
    if st.name in SKIP:
      continue
    for tr, method, N, cpi in CASES + FULL.get(st.name, []):
      key = f"{st.name}/{tr}/{method or '-'}/{N}x{cpi}"
      try:
        kw = dict(system=st, intervals=N, controls_per_interval=cpi)
        if tr == "SHOOTING":
          kw.update(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod[method])
        else:
          kw.update(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[tr])
        hp = HParams(**kw)
        system = st.value() if callable(st.value) else hp.system()
        opt = get_optimizer(hp, CFG, hp.system())
      except Exception as e:          # the reference itself refuses some combinations (PREDATORPREY under collocation: x_T with None)
        log.append(f"{key}: {type(e).__name__} at construction: {str(e)[:80]}")
        out[key + "/error"] = np.array(type(e).__name__)
        continue
      guess = np.asarray(opt.guess, dtype=np.float64)
      bounds = np.asarray(opt.bounds, dtype=np.float64)
      z = guess + 0.01 * (1.0 + np.abs(guess)) * rng.standard_normal(guess.shape)
      try:
        f = float(np.asarray(opt.objective(z)))
        c = np.asarray(opt.constraints(z), dtype=np.float64).ravel()
      except Exception as e:
        log.append(f"{key}: {type(e).__name__} in the callbacks: {str(e)[:80]}")
        out[key + "/error"] = np.array(type(e).__name__)
        continue
      out[key + "/guess"] = guess; out[key + "/bounds"] = bounds; out[key + "/z"] = z
      out[key + "/objective"] = np.array(f); out[key + "/constraints"] = c
      xs, us = opt.unravel(z)
      out[key + "/x_rows"] = np.array(np.asarray(xs).shape[0]); out[key + "/u_rows"] = np.array(np.asarray(us).shape[0])
      # post-solve rollout of the controls of z (utils.py:258-298): the rule the metric config uses (RK4) where the control rows allow it
      sysobj = hp.system()
      try:
        hp_r = HParams(**{**kw, "integration_method": IntegrationMethod.RK4})
        us_arr = np.asarray(us, dtype=np.float64)
        if us_arr.shape[0] >= 2 * hp_r.num_steps + 1:
          xs_r, cost_r = get_state_trajectory_and_cost(hp_r, sysobj, sysobj.x_0, us_arr)
          out[key + "/rollout_rk4_xs"] = np.asarray(xs_r, dtype=np.float64); out[key + "/rollout_rk4_cost"] = np.array(float(np.asarray(cost_r)))
      except Exception as e:
        log.append(f"{key}: {type(e).__name__} in the RK4 rollout: {str(e)[:80]}")
      log.append(f"{key}: n={guess.size} m={c.size} f={f:.12g} |c|max={np.abs(c).max():.6g}")
  # ---- the reference's solve() itself (nlp_solvers/__init__.py:18-98, SLSQP branch: the SciPy call with jax.grad / jax.jacrev callbacks, here
  # complex-step derivatives of the reference's own objective / constraints) on problems small enough for Python loops
  from myriad.config import NLPSolverType
  SOLVES = [("CARTPOLE", dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=5)),
            ("CARTPOLE", dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=8)),
            ("VANDERPOL", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=1, controls_per_interval=20)),
            ("CANCERTREATMENT", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=1, controls_per_interval=20)),
            ("SIMPLECASE", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=2, controls_per_interval=10)),
            # BASELINE configs 3, 4 and 1 at their full shapes (SURVEY.md App. C measured the same three with its own probe: 2.8731963348, 20.5735535185, -1.3543305221)
            ("VANDERPOL", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=1, controls_per_interval=50)),
            ("CANCERTREATMENT", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=1, controls_per_interval=100, max_iter=500)),
            ("SIMPLECASE", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=10, controls_per_interval=100)),
            # round 6 (VERDICT r5 next #6b): collocation solves larger than N = 8 computed by the reference's solve() -- the headline system at a quarter of its horizon
            ("CARTPOLE", dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=25)),
            ("CARTPOLE", dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=25))]
  if os.environ.get("MYRIAD_REF_FULL_SOLVE"):
    # round 6, late: the HEADLINE problem at its full horizon (BASELINE config 2's shape, one trajectory from the reference's own start state) through the
    # reference's solve() -- tens of minutes of SLSQP with complex-step Jacobians of 1005 variables; run as
    #   MYRIAD_REF_FULL_SOLVE=1 python tests/golden/make_reference_fixtures.py --solve-only reference_solve_full
    SOLVES = [("CARTPOLE", dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=100))]
  for name, kw in SOLVES:
    hp = HParams(system=SystemType[name], nlpsolver=NLPSolverType.SLSQP, **kw)
    opt = get_optimizer(hp, CFG, hp.system())
    res = opt.solve()                                        # base.py:69-79 -> nlp_solvers.solve
    key = "solve/%s/%s/%s/%dx%d" % (name, kw["optimizer"].name, (kw.get("quadrature_rule") or kw.get("integration_method")).name, hp.intervals, hp.controls_per_interval)
    out[key + "/xs_and_us"] = np.real(np.asarray(res["xs_and_us"], dtype=np.float64)); out[key + "/cost"] = np.array(float(np.real(res["cost"])))
    out[key + "/x"] = np.asarray(res["x"], dtype=np.float64); out[key + "/u"] = np.asarray(res["u"], dtype=np.float64)
    c = np.asarray(opt.constraints(np.asarray(res["xs_and_us"])), dtype=np.float64)
    log.append(f"{key}: cost={float(np.real(res['cost'])):.12g} |c|max={np.abs(c).max():.3g}")
  # ---- the reference's Forward-Backward Sweep (trajectory_optimizers/forward_backward_sweep.py:88-116 over utils.py:138-197), every indirect system
  # without a terminal state condition (and the discrete INVASIVEPLANT), fbsm_intervals = 200, capped at 40 sweeps through the stopping rule
  if not solve_only:
    from myriad.trajectory_optimizers.forward_backward_sweep import FBSM
    for st in SystemType:
      try:
        hp = HParams(system=st, optimizer=OptimizerType.FBSM, fbsm_intervals=200)
        system = hp.system()
        if not hasattr(system, "adj_ODE"):
          continue
        opt = FBSM(hp, CFG, system)
      except Exception as e:
        log.append(f"fbsm/{st.name}: {type(e).__name__} at construction: {str(e)[:80]}")
        continue
      if opt.terminal_cdtion:
        continue            # (PREDATORPREY: the secant sequence solver -- a loop over the sweeps recorded here)
      count = {"n": 0}
      stop0 = opt.stopping_criterion
      def capped(xi, ui, ai, delta=0.001, _stop0=stop0, _count=count):
        _count["n"] += 1
        return bool(_stop0(xi, ui, ai, delta)) and _count["n"] < 40
      opt.stopping_criterion = capped
      try:
        sol = opt.solve()
      except Exception as e:
        log.append(f"fbsm/{st.name}: {type(e).__name__} in solve: {str(e)[:80]}")
        continue
      key = f"fbsm/{st.name}"
      out[key + "/x"] = np.asarray(sol["x"], dtype=np.float64); out[key + "/u"] = np.asarray(sol["u"], dtype=np.float64)
      out[key + "/adj"] = np.asarray(sol["adj"], dtype=np.float64); out[key + "/sweeps"] = np.array(count["n"])
      log.append(f"{key}: {count['n']} sweeps, u[0]={float(np.asarray(sol['u']).ravel()[0]):.10g} x[-1]={np.asarray(sol['x'])[-1].tolist()}")
  # ---- the reference's extragradient iteration (nlp_solvers/extra_gradient.py:10-84; jax.grad of its Lagrangian by complex-step derivatives), 25 steps
  if not solve_only:
    from myriad.nlp_solvers.extra_gradient import extra_gradient
    EXGD = [("CARTPOLE", dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=6)),
            ("VANDERPOL", dict(optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=5)),
            ("SIMPLECASE", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=2, controls_per_interval=5)),
            ("MOULDFUNGICIDE", dict(optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.EULER, intervals=1, controls_per_interval=8))]
    for name, kw in EXGD:
      hp = HParams(system=SystemType[name], **kw)
      opt = get_optimizer(hp, CFG, hp.system())
      res = extra_gradient(fun=opt.objective, x0=np.asarray(opt.guess, dtype=np.float64), method="exgd", constraints={"fun": opt.constraints},
                           bounds=np.asarray(opt.bounds, dtype=np.float64), jac=None, options={"maxiter": 25, "eta_x": 1e-2, "eta_v": 1e-3})
      key = "exgd/%s/%s/%s/%dx%d" % (name, kw["optimizer"].name, (kw.get("quadrature_rule") or kw.get("integration_method")).name, hp.intervals, hp.controls_per_interval)
      out[key + "/x"] = np.real(np.asarray(res["x"])).astype(np.float64); out[key + "/v"] = np.real(np.asarray(res["v"])).astype(np.float64)
      out[key + "/fun"] = np.array(float(np.real(res["fun"])))
      log.append(f"{key}: 25 steps, fun={float(np.real(res['fun'])):.12g} |v|max={np.abs(np.real(np.asarray(res['v']))).max():.6g}")
  import scipy
  out["scipy_version"] = np.array(scipy.__version__); out["numpy_version"] = np.array(np.__version__)
  path = os.path.join(HERE, (solve_only or "reference_callbacks") + ".npz")
  np.savez_compressed(path, **out)
  open(os.path.join(HERE, (solve_only or "reference_callbacks") + ".log"), "w").write("\n".join(log) + "\n")
  print("\n".join(log[-12:]))
  print(f"{len([k for k in out if k.endswith('/objective')])} cases -> {path} ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
  main()
