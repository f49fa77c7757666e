"""GPU parity tests of the neural-ODE path (BASELINE config 5): CARTPOLE + NodeSystem (2 x 64 sigmoid MLP) through the
Hermite-Simpson transcription -- eval kernel vs the oracle's autodiff, solve through solve_with_params."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer

CFG = Config(verbose=False, plot=False)


def _setup(N):
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               integration_method=IntegrationMethod.RK4, intervals=N, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
  node = NeuralODE.load_fitted_cartpole()
  node_system = NodeSystem(node, hp.system())             # useful_scripts.py:82-84
  return hp, node, get_optimizer(hp, CFG, node_system)


def test_node_eval_matches_oracle_autodiff():
  from oracle import myriad_oracle as O
  N = 6
  hp, node, opt = _setup(N)
  s = O.NodeCartPole(node.params)
  tr = O.hermite_simpson(s, N)
  cb = O.Callbacks(tr)
  rng = np.random.default_rng(0)
  z = tr.guess + 0.3 * rng.standard_normal(tr.guess.size)
  np.testing.assert_allclose(opt.parametrized_constraints(node.params, z), cb.cons(z), rtol=1e-11, atol=1e-12)
  assert opt.parametrized_objective(node.params, z) == pytest.approx(cb.fun(z), rel=1e-12)
  np.testing.assert_allclose(opt.constraints_jac(z, params=node.params), cb.jac(z), rtol=1e-10, atol=1e-11)
  np.testing.assert_allclose(opt.objective_grad(z, params=node.params), cb.grad(z), rtol=1e-12, atol=1e-13)
  # the network reproduces the true field where it was fitted (R^2 = 0.997): sanity of the committed weights
  true = get_optimizer(hp, CFG, hp.system())
  ct = true.constraints(tr.guess)
  diff = np.abs(opt.parametrized_constraints(node.params, tr.guess) - ct).max()
  assert diff < 0.6 and diff < 0.1 * np.abs(ct).max()      # measured 0.49 against constraints of size 6.4 (h = 1/3)


def test_node_solve_with_params_converges_and_is_kkt_point():
  """run_node_trajectory_opt's core (useful_scripts.py:79-89): plan through the network."""
  from oracle import myriad_oracle as O
  N = 20
  hp, node, opt = _setup(N)
  sol = opt.solve_with_params(node.params)
  z = sol['xs_and_us']
  s = O.NodeCartPole(node.params)
  cb = O.Callbacks(O.hermite_simpson(s, N))
  assert np.abs(cb.cons(z)).max() <= 1e-8
  assert cb.fun(z) == pytest.approx(sol['cost'], rel=1e-12)
  lb, ub = opt.bounds[:, 0], opt.bounds[:, 1]
  r = cb.grad(z) + cb.jac(z).T @ sol['lambda']
  inact = (lb < ub) & (z - lb > 1e-3) & (ub - z > 1e-3)
  assert np.abs(r[inact]).max() < 1e-5
  # planning through the fitted network lands near the true-dynamics optimum (golden N=25: 87.95; N=20 similar)
  assert 60.0 < sol['cost'] < 130.0


def test_node_batch_config5_shape():
  """BASELINE config 5 shape on one GPU's shard: N=100, 128 random x0, one shared weight set."""
  from oracle import myriad_oracle as O
  hp, node, opt = _setup(100)
  x0 = O.random_x0(O.CartPole(), 128, seed=2019)
  res = opt.solve_batch(x0s=x0, params=opt.system.device_params())
  assert (res['status'] == 0).all(), np.bincount(res['status'])          # every instance (round 2 tolerated 2 %)
  ev = opt.engine.eval(res['xs_and_us'], params=opt.system.device_params(), want=("c",))
  assert np.abs(ev["c"][res['status'] == 0]).max() <= 1e-8


@pytest.mark.parametrize("B,knob,values", [
    (12, "MYRIAD_NODE_COOP", ("1", "0")),     # fewer trajectories than CUs: four wavefronts share one trajectory's network passes / one wavefront does them
    (300, "MYRIAD_NODE_WPB", ("4", "1")),     # between one and two trajectories per CU: cooperative with four wavefronts / one wavefront alone
    (1100, "MYRIAD_NODE_COOP", ("0", "1")),   # four independent solves per workgroup sharing the weights / cooperative
])
def test_node_workgroup_modes_agree(monkeypatch, B, knob, values):
  """The three ways the network solver occupies a workgroup (hs_solver_wave.h: cooperative, four independent wavefronts, one
  wavefront) run the same arithmetic per trajectory: identical iteration counts, costs equal to rounding."""
  N = 20
  rng = np.random.default_rng(5)
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0")     # kernels are compared: no host-side rescue of a failed device solve
  monkeypatch.setenv("MYRIAD_SOLVE_MODE", "wave1")    # round 2's kernel (the fused kernel, the default since round 3, has one form)
  out = []
  for v in values:
    monkeypatch.setenv(knob, v)
    hp, node, opt = _setup(N)
    out.append(opt.solve_batch(x0s=x0, params=opt.system.device_params()))
  a, b = out
  assert (a["status"] == 0).all() and (b["status"] == 0).all()
  assert (a["iters"] == b["iters"]).all()
  np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-12)
  np.testing.assert_allclose(a["xs_and_us"], b["xs_and_us"], rtol=0, atol=1e-9)


def test_node_fused_kernel_agrees_with_round2_kernel(monkeypatch):
  """Network dynamics on the fused kernel (hs_solver_fused.h, four wavefronts per trajectory: matrix-core passes AND the
  parallel phases shared) against round 2's HsWave: one algorithm, same optima, iteration counts equal up to the order of the sums."""
  N = 40
  rng = np.random.default_rng(8)
  B = 24
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0")
  out = {}
  for mode in ("wave", "wave1"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    hp, node, opt = _setup(N)
    out[mode] = opt.solve_batch(x0s=x0, params=opt.system.device_params())
  a, b = out["wave"], out["wave1"]
  assert (a["status"] == 0).all() and (b["status"] == 0).all()
  np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-9)
  assert (a["iters"] == b["iters"]).mean() >= 0.85, np.bincount(np.abs(a["iters"] - b["iters"]))
  same = a["iters"] == b["iters"]
  assert np.abs(a["xs_and_us"][same] - b["xs_and_us"][same]).max() < 1e-6


def test_node_cooperative_mode_with_per_trajectory_weights(monkeypatch):
  """params [B, np] (a weight set per trajectory): the solving wavefront reloads the weights per trajectory while its helpers
  wait at the mailbox barrier."""
  rng = np.random.default_rng(6)
  B = 6
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  out = []
  monkeypatch.setenv("MYRIAD_SOLVE_MODE", "wave1")            # round 2's kernel: cooperative / one wavefront
  for v in ("1", "0"):
    monkeypatch.setenv("MYRIAD_NODE_COOP", v)
    hp, node, opt = _setup(20)
    out.append(opt.solve_batch(x0s=x0, params=np.tile(opt.system.device_params(), (B, 1))))
  a, b = out
  assert (a["status"] == 0).all() and (a["iters"] == b["iters"]).all()
  np.testing.assert_allclose(a["cost"], b["cost"], rtol=1e-12)
  # the fused kernel (default): a weight set per trajectory is reloaded into LDS per trajectory; same optima as shared weights
  monkeypatch.setenv("MYRIAD_SOLVE_MODE", "wave")
  hp, node, opt = _setup(20)
  f1 = opt.solve_batch(x0s=x0, params=np.tile(opt.system.device_params(), (B, 1)))
  f0 = opt.solve_batch(x0s=x0, params=opt.system.device_params())
  assert (f1["status"] == 0).all() and (f1["iters"] == f0["iters"]).all()
  np.testing.assert_allclose(f1["cost"], f0["cost"], rtol=1e-12)
  np.testing.assert_allclose(f1["cost"], a["cost"], rtol=1e-9)


@pytest.mark.parametrize("N,B", [(20, 8), (100, 64), (100, 128)])
def test_node_helper_workgroups_return_the_bits_of_the_launch_without(monkeypatch, N, B):
  """Round 5 (hs_solver_fused.h: NodeBoard): a batch of at most half the CUs gets helper workgroups that take a share of the tiles of every
  network pass -- other CUs of the owner's XCD, synchronised through global memory.  A tile is the same instruction sequence on the same inputs
  whoever runs it: status, iterations and every bit of z*, lambda*, cost are those of the launch without helpers, also under poison."""
  x0 = np.clip(0.1 * np.random.default_rng(100 * N + B).standard_normal((B, 4)), -2, 2)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  out = {}
  for nh, poison in (("0", None), ("1", None), ("3", None), (None, "random")):
    for k in ("MYRIAD_NODE_HELPERS", "MYRIAD_POISON"):
      monkeypatch.delenv(k, raising=False)
    if nh is not None:
      monkeypatch.setenv("MYRIAD_NODE_HELPERS", nh)
    if poison:
      monkeypatch.setenv("MYRIAD_POISON", poison)
    hp, node, opt = _setup(N)
    r = opt.solve_batch(x0s=x0, params=opt.system.device_params())
    out[(nh, poison)] = {k: np.array(r[k]) for k in ("status", "iters", "cost", "xs_and_us", "lambda")}
    opt.engine.close()
  ref = out[("0", None)]
  assert (ref["status"] == 0).all()
  for key, r in out.items():
    for k in ref:
      assert np.array_equal(r[k], ref[k]), (key, k)


@pytest.mark.parametrize("N,B", [(20, 12), (100, 300)])
def test_node_two_wavefront_throughput_form_walks_the_path_of_the_four_wavefront_form(monkeypatch, N, B):
  """Round 5: beyond one trajectory per CU the network kernel runs two wavefronts per trajectory and two trajectories per CU (bound multipliers in the
  global scratch slot so that two workgroups fit the LDS; one trajectory's sweep overlaps the other's matrix-core passes).  Tiles, scans and sums are the
  four-wavefront form's.  Round 6: the sweep is the two-level one, and the two forms cut the horizon into two resp. four chunks -- the same Newton steps in
  another order of operations: same statuses, same iteration counts, optima within 1e-9 (bit-identical up to round 5).  Within the two-wavefront form the
  bits do not depend on helper workgroups, LDS / scratch poison or the register / stack fill."""
  x0 = np.clip(0.1 * np.random.default_rng(7 * N + B).standard_normal((B, 4)), -2, 2)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  out = {}
  for name, env in (("W4", {"MYRIAD_FUSED_WAVES": "4"}), ("W2", {"MYRIAD_FUSED_WAVES": "2"}), ("W2 no helpers", {"MYRIAD_FUSED_WAVES": "2", "MYRIAD_NODE_HELPERS": "0"}),
                    ("W2 poison", {"MYRIAD_FUSED_WAVES": "2", "MYRIAD_POISON": "random"}),
                    ("W2 fill", {"MYRIAD_FUSED_WAVES": "2", "MYRIAD_REG_FILL": "nan", "MYRIAD_STACK_FILL": "nan"}),
                    ("W4 fill", {"MYRIAD_FUSED_WAVES": "4", "MYRIAD_REG_FILL": "nan", "MYRIAD_STACK_FILL": "nan"}),
                    ("W4 poison", {"MYRIAD_FUSED_WAVES": "4", "MYRIAD_POISON": "random"})):
    for k in ("MYRIAD_FUSED_WAVES", "MYRIAD_NODE_HELPERS", "MYRIAD_POISON", "MYRIAD_REG_FILL", "MYRIAD_STACK_FILL"):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    hp, node, opt = _setup(N)
    r = opt.solve_batch(x0s=x0, params=opt.system.device_params())
    if name.startswith("W2"):
      assert opt.engine.solve_plan()["waves_per_trajectory"] == 2
    out[name] = {k: np.array(r[k]) for k in ("status", "iters", "cost", "xs_and_us", "lambda")}
    opt.engine.close()
  ref = out["W2"]
  assert (ref["status"] == 0).all()
  for name, r in out.items():
    if name.startswith("W4") and name != "W4":       # the four-wavefront form under register / stack fill and poison: its own bits
      for k in ref:
        assert np.array_equal(r[k], out["W4"][k]), (name, k)
      continue
    if name == "W4":
      assert np.array_equal(r["status"], ref["status"]) and np.array_equal(r["iters"], ref["iters"]), (name, r["iters"], ref["iters"])
      np.testing.assert_allclose(r["cost"], ref["cost"], rtol=1e-11, atol=0.0)
      assert np.abs(r["xs_and_us"] - ref["xs_and_us"]).max() <= 1e-9 and np.abs(r["lambda"] - ref["lambda"]).max() <= 1e-7 * max(1.0, np.abs(ref["lambda"]).max())
      continue
    for k in ref:
      assert np.array_equal(r[k], ref[k]), (name, k)


# ---- NodeSystem under SHOOTING: the reference's DEFAULT route for a NodeSystem (config.py:66 optimizer = SHOOTING; solve_with_params goes through
# ---- parametrized_objective / parametrized_constraints of the shooting transcription, shooting.py:144-167, 212-228; useful_scripts.py:79-89) --------------

def _setup_shooting(intervals, cpi, method):
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.SHOOTING, integration_method=method, intervals=intervals, controls_per_interval=cpi,
               hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
  node = NeuralODE.load_fitted_cartpole()
  return hp, node, get_optimizer(hp, CFG, NodeSystem(node, hp.system()))


@pytest.mark.parametrize("method", [IntegrationMethod.HEUN, IntegrationMethod.RK4, IntegrationMethod.EULER, IntegrationMethod.MIDPOINT])
@pytest.mark.parametrize("intervals,cpi", [(4, 3), (1, 10)])
def test_node_shooting_eval_matches_oracle_autodiff(method, intervals, cpi):
  """myr_eval of the shooting transcription with the network as the dynamics: objective, constraints and both derivatives against the oracle's autodiff
  of its restatement of shooting.py:144-167 (parametrized_objective) and :212-228 (parametrized_constraints)."""
  from oracle import myriad_oracle as O
  hp, node, opt = _setup_shooting(intervals, cpi, method)
  tr = O.shooting(O.NodeCartPole(node.params), intervals, cpi, method.name)
  cb = O.Callbacks(tr)
  rng = np.random.default_rng(1)
  z = tr.guess + 0.2 * rng.standard_normal(tr.guess.size)
  np.testing.assert_allclose(opt.parametrized_constraints(node.params, z), cb.cons(z), rtol=1e-11, atol=1e-11)
  assert opt.parametrized_objective(node.params, z) == pytest.approx(cb.fun(z), rel=1e-11)
  np.testing.assert_allclose(opt.constraints_jac(z, params=node.params), cb.jac(z), rtol=1e-9, atol=1e-10)
  np.testing.assert_allclose(opt.objective_grad(z, params=node.params), cb.grad(z), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("method", [IntegrationMethod.HEUN, IntegrationMethod.RK4])
def test_node_shooting_solve_with_params_is_kkt_point(method):
  """run_node_trajectory_opt with the reference's default flags (useful_scripts.py:79-89): multiple shooting through the network."""
  from oracle import myriad_oracle as O
  intervals, cpi = 20, 2
  hp, node, opt = _setup_shooting(intervals, cpi, method)
  sol = opt.solve_with_params(node.params)
  z = sol['xs_and_us']
  cb = O.Callbacks(O.shooting(O.NodeCartPole(node.params), intervals, cpi, method.name))
  assert np.abs(cb.cons(z)).max() <= 1e-8
  assert cb.fun(z) == pytest.approx(sol['cost'], rel=1e-10)
  lb, ub = opt.bounds[:, 0], opt.bounds[:, 1]
  r = cb.grad(z) + cb.jac(z).T @ sol['lambda']
  inact = (lb < ub) & (z - lb > 1e-3) & (ub - z > 1e-3)
  assert np.abs(r[inact]).max() < 1e-5
  assert 40.0 < sol['cost'] < 200.0


def test_node_shooting_batch():
  """A B = 1024 batch of multiple-shooting problems through the network (20 intervals x 5 controls, random x0, one shared weight set): every instance
  converges (73 ms on one MI355X: the rollouts, linearisations and second derivatives run as matrix-core passes over the stage points of all
  intervals / steps, shoot_solver_wave.h).  SINGLE shooting of this swing-up (intervals = 1, 100 controls) is another matter: through the network as
  through the true dynamics it is ill-conditioned -- the true-dynamics case of the reference's smoke matrix needs a second start
  (tests/test_gpu_smoke.py), and through the network 52 of 64 random starts reach a KKT point, after thousands of iterations (tools/dev/node_shoot_probe.py)."""
  from oracle import myriad_oracle as O
  hp, node, opt = _setup_shooting(20, 5, IntegrationMethod.HEUN)
  x0 = O.random_x0(O.CartPole(), 1024, seed=2019)
  res = opt.solve_batch(x0s=x0, params=opt.system.device_params())
  assert (res['status'] == 0).all(), np.bincount(res['status'])
  ev = opt.engine.eval(res['xs_and_us'], params=opt.system.device_params(), want=("c",))
  assert np.abs(ev["c"]).max() <= 1e-8


@pytest.mark.parametrize("method,intervals,cpi", [(IntegrationMethod.HEUN, 4, 3), (IntegrationMethod.RK4, 3, 2), (IntegrationMethod.EULER, 4, 3), (IntegrationMethod.MIDPOINT, 3, 3)])
def test_node_shooting_wavefront_kernel_agrees_with_the_lane_kernel(monkeypatch, method, intervals, cpi):
  """The wavefront kernel of the network system evaluates the network by matrix-core passes over point lists (round 6); the lane kernel
  (MYRIAD_SOLVE_MODE=lane) walks the layers per lane (node_system.h).  Same algorithm, two evaluations of the same network: same status, iteration
  counts within one, the same optimum -- for every integration rule of the reference (utils.py:31-54)."""
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  x0 = np.clip(0.1 * np.random.default_rng(5).standard_normal((3, 4)), -2, 2)      # (small: the lane kernel takes seconds per solve)
  out = {}
  for mode in ("wave", "lane"):
    monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    hp, node, opt = _setup_shooting(intervals, cpi, method)
    out[mode] = opt.solve_batch(x0s=x0, params=opt.system.device_params(), max_iter=150)
    opt.engine.close()
  w, l = out["wave"], out["lane"]
  # (long solves are chaotic in the last bits of their merit sums -- at 4 x 5 midpoint the two kernels share the iteration counts 80, 97, 98 of three
  #  instances and part on the three that take 180 and more: compared are the solves both kernels finish within 100 iterations)
  ok = (w["status"] == 0) & (l["status"] == 0) & (w["iters"] <= 100) & (l["iters"] <= 100)
  assert ok.any(), (w["status"], l["status"], w["iters"], l["iters"])
  assert np.abs(w["iters"][ok].astype(int) - l["iters"][ok].astype(int)).max() <= 1, (w["iters"], l["iters"])
  np.testing.assert_allclose(w["cost"][ok], l["cost"][ok], rtol=1e-8)
  same = ok & (w["iters"] == l["iters"])
  assert np.abs(w["xs_and_us"][same] - l["xs_and_us"][same]).max(initial=0.0) <= 1e-6


@pytest.mark.parametrize("N", [2, 3, 5, 7])
def test_two_level_sweep_with_fewer_stages_than_wavefronts(monkeypatch, N):
  """The two-level sweep cuts the horizon into W chunks of N / W stages: with N = 2 or 3 and four wavefronts some chunks are EMPTY (the join skips
  them, block 0 is filled from the first non-empty one), with N = 5 or 7 they have one or two stages (terminal weights of unreachable directions are
  floored).  Both multi-wavefront forms of the network kernel against round 2's kernel, whose sweep is the plain recursion: same statuses and iteration
  counts, optima within 1e-10."""
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  x0 = np.clip(0.1 * np.random.default_rng(N).standard_normal((4, 4)), -2, 2)
  out = {}
  for tag, env in (("w4", {"MYRIAD_FUSED_WAVES": "4"}), ("w2", {"MYRIAD_FUSED_WAVES": "2"}), ("r2", {"MYRIAD_SOLVE_MODE": "wave1"})):
    for k in ("MYRIAD_FUSED_WAVES", "MYRIAD_SOLVE_MODE"):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    hp, node, opt = _setup(N)
    out[tag] = opt.solve_batch(x0s=x0, params=opt.system.device_params(), max_iter=200)
    opt.engine.close()
  ref = out["r2"]
  assert (ref["status"] == 0).all()
  for tag in ("w4", "w2"):
    r = out[tag]
    assert np.array_equal(r["status"], ref["status"]) and np.array_equal(r["iters"], ref["iters"]), (tag, r["status"], r["iters"], ref["iters"])
    assert np.abs(r["xs_and_us"] - ref["xs_and_us"]).max() <= 1e-10 and np.abs(r["cost"] - ref["cost"]).max() <= 1e-10


def test_run_node_trajectory_opt_with_the_references_default_flags():
  """useful_scripts.py:79-100 as a drop-in: HParams() names SHOOTING (config.py:66) -- here CARTPOLE, 20 intervals x 5 controls, Heun --, the plan comes from
  `solve_with_params(node.params)` through the network, the returned cost and defect from rolling the TRUE dynamics forward under the planned controls."""
  from myriad_amd.useful_scripts import run_node_trajectory_opt
  hp = HParams(system=SystemType.CARTPOLE, intervals=20, controls_per_interval=5, hidden_layers=(64, 64))
  assert hp.optimizer == OptimizerType.SHOOTING
  c, defect = run_node_trajectory_opt(hp, CFG)
  assert np.isfinite(c) and 40.0 < float(c) < 250.0
  assert defect is not None and np.all(np.isfinite(defect))
  # the network is a fit (R^2 = 0.997), not the true field: the true-dynamics rollout under its plan misses the target by a visible but bounded amount
  assert np.abs(np.asarray(defect)).max() < 3.0
