"""INTEGRATION.md must not rot: the reference-side binding it shows (the `hip_sqp` / `hip_fbsm` ctypes stubs a maintainer
would add at nlp_solvers/__init__.py:14,44 and forward_backward_sweep.py:88-116) is extracted from the document AS
WRITTEN and run against the library.  CPU part: the blocks exist, compile, and every C entry point they call is
declared in include/myriad_hip.h with the same number of arguments.  GPU part: the stubs solve."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "myriad_amd", "libmyriad_hip.so")


def _blocks():
  txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
  blocks = re.findall(r"```python\n(.*?)```", txt, re.S)
  stub = [b for b in blocks if "def hip_sqp" in b]
  fbsm = [b for b in blocks if "def hip_fbsm" in b]
  assert len(stub) == 1 and len(fbsm) == 1
  return stub[0], fbsm[0]


def _starts_block():
  txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
  blk = [b for b in re.findall(r"```python\n(.*?)```", txt, re.S) if "def hip_starts" in b]
  assert len(blk) == 1
  return blk[0]


def _header_arity():
  hdr = open(os.path.join(ROOT, "include", "myriad_hip.h")).read()
  hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
  out = {}
  for name, args in re.findall(r"\b(myr_\w+)\s*\(([^;{]*?)\)\s*;", hdr):
    args = args.strip()
    out[name] = 0 if args in ("", "void") else args.count(",") + 1
  return out


def _call_arity(src):
  """(name, number of arguments) of every lib.myr_*( ... ) call in `src` (balanced-parenthesis scan)."""
  calls = []
  for m in re.finditer(r"lib\.(myr_\w+)\(", src):
    i, depth, nargs, seen = m.end(), 1, 0, False
    while depth:
      ch = src[i]
      if ch in "([{":
        depth += 1
      elif ch in ")]}":
        depth -= 1
      elif ch == "," and depth == 1:
        nargs += 1
      if depth and not ch.isspace():
        seen = True
      i += 1
    calls.append((m.group(1), nargs + 1 if seen else 0))
  return calls


def test_integration_stubs_compile_and_match_the_header():
  stub, fbsm = _blocks()
  compile(stub, "INTEGRATION.md:hip_sqp", "exec")
  compile(fbsm, "INTEGRATION.md:hip_fbsm", "exec")
  starts = _starts_block()
  compile(starts, "INTEGRATION.md:hip_starts", "exec")
  arity = _header_arity()
  calls = _call_arity(stub) + _call_arity(fbsm) + _call_arity(starts)
  assert "myr_solve_x0" in {c for c, _ in calls}
  assert {"myr_create", "myr_solve", "myr_fbsm", "myr_destroy", "myr_get_dims", "myr_last_error"} <= {c for c, _ in calls}
  for name, n in calls:
    assert name in arity, f"{name} is not declared in include/myriad_hip.h"
    assert n == arity[name], f"INTEGRATION.md calls {name} with {n} arguments, the header declares {arity[name]}"
  assert "myr_last_error.restype" in stub


def _namespace(monkeypatch):
  monkeypatch.setenv("MYRIAD_HIP_LIB", LIB)
  stub, fbsm = _blocks()
  ns = {}
  exec(compile(stub, "INTEGRATION.md:hip_sqp", "exec"), ns)
  exec(compile(fbsm, "INTEGRATION.md:hip_fbsm", "exec"), ns)
  exec(compile(_starts_block(), "INTEGRATION.md:hip_starts", "exec"), ns)
  return ns


@pytest.mark.gpu
def test_hip_sqp_stub_solves_like_the_package(monkeypatch):
  from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  ns = _namespace(monkeypatch)
  for kw in (dict(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
                  integration_method=IntegrationMethod.RK4, intervals=10),
             dict(system=SystemType.SIMPLECASE, optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN,
                  intervals=10, controls_per_interval=100)):
    hp = HParams(nlpsolver=NLPSolverType.SQP, **kw)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    ref = opt.solve()
    sol = ns["hip_sqp"](hp, opt._opt_inputs())
    assert sol["success"]
    assert sol["fun"] == pytest.approx(ref["cost"], rel=1e-12)
    np.testing.assert_allclose(sol["x"], ref["xs_and_us"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(sol["v"], ref["lambda"], rtol=0, atol=1e-10)
  # error path: decode() of myr_last_error works (restype set) and reports the C-side message
  lib, C = ns["lib"], ns["C"]
  bad = ns["Desc"](999, 0, 1, 10, 1, 0, 1, 0, 1.0)
  h = C.c_void_p()
  assert lib.myr_create(C.byref(bad), C.byref(h)) != 0
  assert isinstance(ns["_err"](), str) and ns["_err"]()


@pytest.mark.gpu
def test_hip_sqp_stub_solves_pendulum_from_the_reference_guess_with_no_host_logic(monkeypatch):
  """the stand-in for IPOPT's restoration phase travels with the C-ABI.  The torque-limited PENDULUM swing-up jams
  at an infeasible stationary point from the reference's straight-line guess (what the reference leaves to IPOPT's restoration phase,
  nlp_solvers/__init__.py:57-58); the ctypes stub of INTEGRATION.md -- myr_create, myr_solve, myr_destroy, nothing else -- returns
  the optimum, because myr_solve itself runs the elastic phase / the second starts (myr_solve_opts.restoration, default on)."""
  from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  ns = _namespace(monkeypatch)
  for k in ("MYRIAD_ELASTIC", "MYRIAD_SECOND_STARTS"):
    monkeypatch.delenv(k, raising=False)
  for rule, N, cost in ((QuadratureRule.HERMITE_SIMPSON, 20, 25.539), (QuadratureRule.TRAPEZOIDAL, 40, 25.43)):
    hp = HParams(system=SystemType.PENDULUM, optimizer=OptimizerType.COLLOCATION, quadrature_rule=rule, integration_method=IntegrationMethod.RK4,
                 intervals=N, nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    sol = ns["hip_sqp"](hp, opt._opt_inputs())
    assert sol["success"], (rule, sol["nit"])
    assert sol["fun"] == pytest.approx(cost, rel=2e-3)


@pytest.mark.gpu
def test_hip_fbsm_stub_matches_the_mirror(monkeypatch):
  from myriad_amd.config import Config, HParams, OptimizerType
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  ns = _namespace(monkeypatch)
  hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.FBSM, fbsm_intervals=200)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  ref = opt.solve_batch()
  lo, hi, bang = opt._clip_bounds()
  s = opt.system
  xs, us, adjs, sweeps = ns["hip_fbsm"]("CANCERTREATMENT", s.T, 200, s.x_0, lo, hi, params=s.device_params(), adj_T=s.adj_T, bang=bang)
  assert int(sweeps[0]) == int(ref["sweeps"][0])
  np.testing.assert_array_equal(xs, ref["x"]); np.testing.assert_array_equal(us, ref["u"]); np.testing.assert_array_equal(adjs, ref["adj"])


@pytest.mark.gpu
def test_hip_starts_stub_matches_solve_batch(monkeypatch):
  """The myr_solve_x0 binding of INTEGRATION.md as written: B start states of CARTPOLE (x_T given: linspace rule) and VANDERPOL (no
  x_T: constant guess) against the package's solve_batch -- identical to the bit."""
  from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  ns = _namespace(monkeypatch)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  rng = np.random.default_rng(3)
  for st in (SystemType.CARTPOLE, SystemType.VANDERPOL):
    hp = HParams(system=st, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
                 integration_method=IntegrationMethod.RK4, intervals=12, nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    x0s = opt.system.x_0[None] + 0.05 * rng.standard_normal((7, opt.system.x_0.shape[0]))
    ref = opt.solve_batch(x0s=x0s)
    sol = ns["hip_starts"](hp, opt._opt_inputs(), x0s)
    assert sol["success"].all()
    np.testing.assert_array_equal(sol["x"], ref["xs_and_us"]); np.testing.assert_array_equal(sol["v"], ref["lambda"])
    np.testing.assert_array_equal(sol["fun"], ref["cost"]); np.testing.assert_array_equal(sol["nit"], ref["iters"])
