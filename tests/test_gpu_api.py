"""GPU tests of the reference-shaped Python API (get_optimizer(...).solve(), solve_with_params, run_trajectory_opt,
rollout) running on the HIP kernels through the C-ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType

CFG = Config(verbose=False, plot=False)


def _hp(N=25, **kw):
  return HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
                 integration_method=IntegrationMethod.RK4, intervals=N, **kw)


def test_run_trajectory_opt_returns_cost_and_defect(golden_dir):
  """useful_scripts.py:26-76 contract: (integrated cost of the true-dynamics RK4 rollout under the solved controls,
  terminal defect); pinned by the oracle's restatement of utils.py:258-324 on the golden solution."""
  from myriad_amd.useful_scripts import run_trajectory_opt
  from oracle import myriad_oracle as O
  c, defect = run_trajectory_opt(_hp(25), CFG)
  d = np.load(os.path.join(golden_dir, "solve_hs_cartpole_N25.npz"))
  s = O.CartPole()
  u_gold = d["z"][0][51 * 4:].reshape(51, 1)
  xs, c_or = O.get_state_trajectory_and_cost(s, 25, "RK4", s.x_0, u_gold)
  assert c == pytest.approx(c_or, rel=1e-6)
  np.testing.assert_allclose(defect, O.get_defect(s, xs), atol=1e-5)
  assert defect.shape == (4,) and np.abs(defect).max() < 1.0       # open-loop RK4 re-integration of an unstable swing-up


def test_optimizer_solve_result_keys_and_shapes():
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = _hp(10)
  opt = get_optimizer(hp, CFG, hp.system())
  sol = opt.solve()
  assert set(sol) == {'x', 'u', 'xs_and_us', 'cost', 'lambda'}           # nlp_solvers/__init__.py:90-96
  assert sol['x'].shape == (21, 4) and sol['u'].shape == (21, 1) and sol['xs_and_us'].shape == (105,) and sol['lambda'].shape == (80,)
  assert sol['cost'] == pytest.approx(85.80432338009395, rel=1e-9)        # golden N=10
  assert np.abs(opt.constraints(sol['xs_and_us'])).max() <= 1e-8
  assert opt.objective(sol['xs_and_us']) == pytest.approx(sol['cost'], rel=1e-12)
  # warm start from the solution (base.py:81-93 solve_with_params(params, guess))
  sol2 = opt.solve_with_params({'g': 9.81, 'm1': 1.0, 'm2': 0.3, 'length': 0.5}, guess=sol['xs_and_us'])
  assert sol2['cost'] == pytest.approx(sol['cost'], rel=1e-7)
  # a different model: heavier pole needs more effort
  sol3 = opt.solve_with_params({'g': 9.81, 'm1': 1.0, 'm2': 0.6, 'length': 0.5})
  assert sol3['cost'] > sol['cost'] * 1.05


def test_scipy_branch_on_gpu_callbacks_agrees_with_sqp():
  """The reference's NLPSolverType.SLSQP branch (nlp_solvers/__init__.py:50-52) fed by the HIP eval kernel."""
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = _hp(5, nlpsolver=NLPSolverType.SLSQP)
  opt = get_optimizer(hp, CFG, hp.system())
  sol = opt.solve()
  hp2 = _hp(5, nlpsolver=NLPSolverType.SQP)
  sol2 = get_optimizer(hp2, CFG, hp2.system()).solve()
  assert sol['cost'] == pytest.approx(sol2['cost'], rel=1e-4)
  assert 'lambda' not in sol                                              # absent for SLSQP, as in the reference


def test_solve_with_params_reaches_the_scipy_branch():
  """base.py:81-93: solve_with_params wraps parametrized_objective / parametrized_constraints, so the SLSQP branch must
  solve the NON-default model too (fun, constraints, Jacobian and gradient all at m2 = 0.6)."""
  from myriad_amd.trajectory_optimizers import get_optimizer
  heavy = {'g': 9.81, 'm1': 1.0, 'm2': 0.6, 'length': 0.5}
  hp = _hp(5, nlpsolver=NLPSolverType.SLSQP)
  opt = get_optimizer(hp, CFG, hp.system())
  hp2 = _hp(5, nlpsolver=NLPSolverType.SQP)
  opt2 = get_optimizer(hp2, CFG, hp2.system())
  sol_sqp = opt2.solve_with_params(heavy)
  sol_default = opt2.solve()
  # swing-up is non-convex (SLSQP from the straight-line guess may pick another basin), so start SLSQP at the optimum of
  # the HEAVY model: it must stay there -- it would walk away if any of its four callbacks still saw the default model
  sol_slsqp = opt.solve_with_params(heavy, guess=sol_sqp['xs_and_us'])
  assert sol_slsqp['cost'] == pytest.approx(sol_sqp['cost'], rel=1e-5)
  np.testing.assert_allclose(sol_slsqp['xs_and_us'], sol_sqp['xs_and_us'], atol=1e-3)
  assert abs(sol_slsqp['cost'] - sol_default['cost']) > 0.02 * sol_default['cost']
  assert np.abs(opt.parametrized_constraints(heavy, sol_slsqp['xs_and_us'])).max() <= 1e-6
  assert np.abs(opt.constraints(sol_slsqp['xs_and_us'])).max() > 1e-4      # NOT feasible for the default model
  # and from the reference guess it returns a point that is feasible for the heavy model, not for the default one
  sol_cold = opt.solve_with_params(heavy)
  assert np.abs(opt.parametrized_constraints(heavy, sol_cold['xs_and_us'])).max() <= 1e-6
  assert np.abs(opt.constraints(sol_cold['xs_and_us'])).max() > 1e-4


def test_solve_batch_extension_parameter_sweep():
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = _hp(10)
  opt = get_optimizer(hp, CFG, hp.system())
  rng = np.random.default_rng(1)
  B = 33
  params = np.array([9.81, 1.0, 0.3, 0.5]) * (1 + 0.1 * rng.uniform(-1, 1, (B, 4)))
  x0s = np.clip(0.1 * rng.standard_normal((B, 4)), -1, 1)
  res = opt.solve_batch(x0s=x0s, params=params)
  assert (res['status'] == 0).all() and res['x'].shape == (B, 21, 4) and res['u'].shape == (B, 21, 1)
  assert np.array_equal(res['x'][:, 0, :], x0s)
  ev = opt.engine.eval(res['xs_and_us'], params=params, want=("c",))
  assert np.abs(ev["c"]).max() <= 1e-8


@pytest.mark.parametrize("method", ["EULER", "HEUN", "MIDPOINT", "RK4"])
def test_rollout_kernel_matches_oracle(method):
  from myriad_amd import _lib
  from oracle import myriad_oracle as O
  rng = np.random.default_rng(2)
  for name in ("CARTPOLE", "VANDERPOL", "CANCERTREATMENT", "SIMPLECASE"):
    s = O.SYSTEMS[name]()
    S, B = 40, 5
    rows = (2 if method == "RK4" else 1) * S + 1
    us = 0.2 * rng.standard_normal((B, rows, 1))
    if name == "CANCERTREATMENT":
      us = np.abs(us)
    x0 = np.tile(s.x_0, (B, 1)) * (1 + 0.01 * rng.standard_normal((B, s.ns)))
    eng = _lib.Engine(name, "SHOOTING", 1, s.T / 8, controls_per_interval=S, integration_method=method)
    xs, cost = eng.rollout(x0, us, S)
    s.T = s.T / 8
    for b in range(B):
      oxs, oc = O.get_state_trajectory_and_cost(s, S, method, x0[b], us[b])
      np.testing.assert_allclose(xs[b], oxs, rtol=1e-12, atol=1e-13)
      assert cost[b] == pytest.approx(oc, rel=1e-12, abs=1e-14)
    eng.close()


def test_gather_solutions_over_rccl_single_rank():
  """The N>1 bench path's only collective, on the RCCL backend (a 1-rank group is all a 1-GPU box allows; the
  world-2 logic is covered on gloo in tests/test_host_api.py)."""
  import os
  import torch
  import torch.distributed as dist
  from myriad_amd.batched import gather_solutions
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
  dist.init_process_group("nccl", rank=0, world_size=1)
  try:
    z = torch.arange(12, dtype=torch.float64, device="cuda").reshape(4, 3)
    st = torch.tensor([0, 1, 0, 3], dtype=torch.int32, device="cuda")
    out = gather_solutions({"z": z, "status": st}, [4])
    assert torch.equal(out["z"], z) and torch.equal(out["status"], st)
  finally:
    dist.destroy_process_group()


def test_solve_batch_fan_out_over_two_handles_of_one_device_matches_one_handle():
  """SURVEY.md 8(b) Threading: the batch axis fans out beneath the unchanged API -- one handle + one host thread per listed
  device.  `devices=[0, 0]` runs the two shards on one GPU (a 1-GPU box): same result, instance by instance, as one handle."""
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = _hp(25)
  rng = np.random.default_rng(3)
  opt1 = get_optimizer(hp, CFG, hp.system()); opt1.devices = [0]
  x0s = np.clip(np.array(opt1.system.x_0)[None] + 0.1 * rng.standard_normal((48, 4)), -2, 2)
  a = opt1.solve_batch(x0s=x0s)
  opt2 = get_optimizer(hp, CFG, hp.system()); opt2.devices = [0, 0]; opt2.min_shard = 8
  assert len(opt2.engines_for(48)) == 2
  b = opt2.solve_batch(x0s=x0s)
  assert (a["status"] == 0).all() and (b["status"] == 0).all()
  for k in ("xs_and_us", "cost", "lambda", "iters", "status", "start", "attempts"):
    assert np.array_equal(a[k], b[k]), k
  # default device list: every visible device (1 here), never more shards than min_shard allows
  opt3 = get_optimizer(hp, CFG, hp.system())
  assert opt3._device_list() == list(range(_lib_device_count())) and len(opt3.engines_for(48)) == 1


def _lib_device_count():
  from myriad_amd import _lib
  return max(1, _lib.device_count())


def test_solve_plan_reports_the_launch_form_and_an_explicit_park_iter_selects_the_one_wavefront_form():
  """myr_solve_plan (round 5): the library says how it launched the last solve -- what bench.py quotes instead of restating the library's rules.  A small
  batch of a closed-form collocation problem runs two wavefronts per trajectory as whole solves; an explicit myr_solve_opts.park_iter > 0 selects the
  one-wavefront form with its two-phase launch (include/myriad_hip.h: park_iter) and returns the same bits; single shooting reports its own kernel."""
  from myriad_amd import _lib
  from myriad_amd.trajectory_optimizers import get_optimizer
  from myriad_amd.config import NLPSolverType
  hp = _hp(20, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, CFG, hp.system())
  x0 = np.clip(0.1 * np.random.default_rng(3).standard_normal((40, 4)), -2, 2)
  z0, lb, ub = opt.batch_inputs(x0)
  eng = opt.engine
  a = eng.solve(z0, lb, ub)
  p = eng.solve_plan()
  assert p["form"] == "fused" and p["waves_per_trajectory"] == 2 and p["park_iter"] == 0 and p["launches_per_solve"] == 1 and p["slots"] == 40
  o = eng.default_opts(); o.park_iter = 5
  b = eng.solve(z0, lb, ub, opts=o)
  p = eng.solve_plan()
  assert p["waves_per_trajectory"] == 1 and p["park_iter"] == 5 and p["launches_per_solve"] == 2
  assert (a["status"] == 0).all() and np.array_equal(a["status"], b["status"]) and np.array_equal(a["iters"], b["iters"])
  np.testing.assert_allclose(a["z"], b["z"], rtol=0, atol=1e-9)       # (W = 1 and W = 2 differ in the order of their merit sums only)
  hs = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=20, nlpsolver=NLPSolverType.SQP)
  os_ = get_optimizer(hs, CFG, hs.system())
  os_.solve_batch(x0s=np.tile(os_.system.x_0, (3, 1)))
  assert os_.engine.solve_plan()["form"] == "shooting_wave"
