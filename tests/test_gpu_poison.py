"""The kernel-form probes of rounds 3 / 4 as tests: every system x {Hermite-Simpson, trapezoidal} x N x B through
  * the fused wavefront kernel with one and with two wavefronts per trajectory, round 2's wavefront kernel and the lane kernel: one
    algorithm, so equal status and optimum, and equal iterates wherever the forms differ by nothing but the order of their sums;
  * MYRIAD_POISON: the LDS and the scratch slot a trajectory inherits overwritten with a signalling NaN, with 1e20 and with
    "plausible leftovers" (finite values of order one, different in every word): a solve that reads before it writes cannot give
    the same bits under all of them;
  * fresh handles: the same problem on three handles created one after the other (new scratch, whatever the previous kernel left in
    LDS) -- the way the two-wavefront kernel of round 3 showed its defect (three processes, three answers).
What it replaces: tools/dev/w2_probe.py, node_coop_probe.py, rocket_probe.py of round 3 (removed; tools/dev/wprobe.py and fresh_stats.py are the
command-line forms of the same comparisons).
Reference behaviour held: /root/reference/tests/test_smoke.py:29-61 (every system returns, from the reference's guess)."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NS_ = (6, 20, 50, 100)
BS_ = (1, 3)
KNOBS = ("MYRIAD_FUSED_WAVES", "MYRIAD_SOLVE_MODE", "MYRIAD_POISON", "MYRIAD_LANE_UNVERIFIED", "MYRIAD_NODE_COOP", "MYRIAD_NODE_WPB", "MYRIAD_PARK_ITER",
         "MYRIAD_REG_FILL", "MYRIAD_STACK_FILL", "MYRIAD_NODE_HELPERS")


def _systems():
  from myriad_amd.systems import SystemType
  return [st.name for st in SystemType if st.name != "INVASIVEPLANT"]       # (discrete: refused by the direct optimisers, base.py:66-67)


def _solve(monkeypatch, env, system, rule, N, B, max_iter=300):
  from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  for k in KNOBS:
    monkeypatch.delenv(k, raising=False)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  hp = HParams(system=SystemType[system], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=N, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  x0 = np.tile(opt.system.x_0, (B, 1)) * (1.0 + 0.01 * np.arange(B)[:, None])
  o = opt.solve_batch(x0s=x0, max_iter=max_iter)
  out = {k: np.array(o[k]) for k in ("status", "iters", "cost", "xs_and_us", "lambda", "kkt")}
  out["bits"] = hashlib.sha1(b"".join(np.ascontiguousarray(out[k]).tobytes() for k in ("xs_and_us", "lambda", "cost", "kkt", "status", "iters"))).hexdigest()
  opt.engine.close()
  return out


def _collocation_ok(system):
  """PREDATORPREY pins ONE terminal state (x_T = [None, None, B]): collocation raises TypeError as the reference does"""
  return system != "PREDATORPREY"


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("system", _systems())
def test_poisoned_inheritance_and_fresh_handles_give_identical_bits(monkeypatch, system, rule):
  """Every wavefront form of the collocation solver that serves this system: the result does not depend on what the trajectory finds
  in LDS / scratch (three poison patterns), nor on the handle it runs on (three fresh handles)."""
  if not _collocation_ok(system):
    pytest.skip("collocation is refused for a partially pinned terminal state (reference behaviour)")
  forms = [{"MYRIAD_FUSED_WAVES": "1"}, {"MYRIAD_FUSED_WAVES": "2"}, {"MYRIAD_SOLVE_MODE": "wave1"}]
  for N in (6, 100):
    for form in forms:
      ref = _solve(monkeypatch, form, system, rule, N, 3)
      for again in range(2):
        r = _solve(monkeypatch, form, system, rule, N, 3)
        assert r["bits"] == ref["bits"], (form, N, "fresh handle", again, ref["status"], r["status"], ref["iters"], r["iters"])
      for pat in ("nan", "big", "random"):
        r = _solve(monkeypatch, dict(form, MYRIAD_POISON=pat), system, rule, N, 3)
        assert r["bits"] == ref["bits"], (form, N, "poison " + pat, ref["status"], r["status"], ref["iters"], r["iters"], ref["cost"], r["cost"])


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("system", _systems())
def test_inherited_registers_and_stack_do_not_reach_the_result(monkeypatch, system, rule):
  """Round 5: what LDS and scratch-slot poison cannot reach.  Vector / accumulation registers and the queue's private-segment memory are not cleared
  between kernels, and this compiler sometimes places a spill (v_accvgpr_write -- AGPRs are spill space on gfx950) at the top of a join block IN FRONT
  of the `s_or_b64 exec` that re-activates the lanes which skipped the divergent region: those lanes never reach the spill slot, and the reload hands
  them what the last wavefront on the SIMD left in the register (tools/dev/scan_exec_prologue.py; tools/dev/exp/exp57..59 located round 4's
  "handle-to-handle nondeterminism" of the speculative rung + called sweep build in a54:a55 of exactly such a block).  MYRIAD_REG_FILL / MYRIAD_STACK_FILL
  leave a pattern in every VGPR / AGPR of every SIMD and in 4 KB of private memory per lane before every solver launch: a kernel that computes with a
  register or stack slot it never wrote cannot return the same bits under zeros, NaNs and finite leftovers.  Small N on purpose: with 7 or 13 points
  most lanes of a wavefront skip the point loops, which is when the lanes that miss a misplaced spill exist."""
  if not _collocation_ok(system):
    pytest.skip("collocation is refused for a partially pinned terminal state (reference behaviour)")
  forms = [{"MYRIAD_FUSED_WAVES": "1"}, {"MYRIAD_FUSED_WAVES": "2"}, {"MYRIAD_SOLVE_MODE": "wave1"}, {"MYRIAD_SOLVE_MODE": "lane"}]
  for N in (6, 20):
    for form in forms:
      ref = _solve(monkeypatch, dict(form, MYRIAD_REG_FILL="zero", MYRIAD_STACK_FILL="zero"), system, rule, N, 3, max_iter=60)
      for pat in ("nan", "random"):
        r = _solve(monkeypatch, dict(form, MYRIAD_REG_FILL=pat, MYRIAD_STACK_FILL=pat), system, rule, N, 3, max_iter=60)
        assert r["bits"] == ref["bits"], (form, N, "registers / stack " + pat, ref["status"], r["status"], ref["iters"], r["iters"], ref["cost"], r["cost"])


def _same_optimum(a, b, tag):
  assert np.array_equal(a["status"], b["status"]), (tag, a["status"], b["status"], a["iters"], b["iters"])
  ok = a["status"] == 0
  np.testing.assert_allclose(a["cost"][ok], b["cost"][ok], rtol=1e-9, atol=1e-12, err_msg=str(tag))


# solves longer than this many iterations are chaotic in the last bits of their merit sums (BIOREACTOR's singular arc takes 90 - 300
# iterations): the forms may then part by an iteration or two on the way to the same optimum
LONG = 60


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("system", _systems())
def test_wavefront_forms_take_the_same_iterations(monkeypatch, system, rule):
  """fused W = 1 against fused W = 2 and against round 2's kernel: same status, same optimum, same iteration count (short solves).
  Round 6: for one control and at most four states the W = 2 form runs the TWO-LEVEL sweep (two chunks condensed side by side, joined at the interface):
  the same Newton steps in another order of operations -- and the inertia test sees other pivots (a chunk is swept from the terminal form rho I instead
  of the true cost-to-go).  On most systems, every BASELINE configuration among them, the two forms still take identical iteration counts and end
  within 1e-9 of each other; on the systems whose stage pivots sit at the regularisation threshold (flat objectives, singular arcs: TWO_LEVEL_PARTS)
  they take different, equally valid regularisation paths to the same optimum (tools/dev/exp/exp85.sh, exp86.sh: not a question of the weight rho)."""
  if not _collocation_ok(system):
    pytest.skip("collocation is refused for a partially pinned terminal state (reference behaviour)")
  TWO_LEVEL_PARTS = {"TUMOUR", "PENDULUM", "CANCERTREATMENT", "BIOREACTOR", "HIVTREATMENT"}
  for N in NS_:
    for B in BS_:
      w1 = _solve(monkeypatch, {"MYRIAD_FUSED_WAVES": "1"}, system, rule, N, B)
      w2 = _solve(monkeypatch, {"MYRIAD_FUSED_WAVES": "2"}, system, rule, N, B)
      r2 = _solve(monkeypatch, {"MYRIAD_SOLVE_MODE": "wave1"}, system, rule, N, B)
      short = (w1["iters"] <= LONG) & (w1["status"] == 0)
      if system in TWO_LEVEL_PARTS:
        # (which of the long solves end inside the iteration limit may differ: BIOREACTOR Hermite-Simpson N = 100 ends [0, 1, 1] on the plain recursion and
        #  [0, 0, 0] on the two-level sweep within 300 iterations)
        ok = (w1["status"] == 0) & (w2["status"] == 0)
        assert ok.any() or not ((w1["status"] == 0).any() and (w2["status"] == 0).any()), (system, rule, N, B, w1["status"], w2["status"])
        np.testing.assert_allclose(w1["cost"][ok], w2["cost"][ok], rtol=5e-4, atol=1e-9, err_msg=str((system, rule, N, B)))
      else:
        _same_optimum(w1, w2, (system, rule, N, B, "W=2"))
        assert np.array_equal(w1["iters"][short], w2["iters"][short]), (system, rule, N, B, w1["iters"], w2["iters"])
        same = short & (w1["iters"] == w2["iters"])
        assert (np.abs(w1["xs_and_us"][same] - w2["xs_and_us"][same]) / np.maximum(1.0, np.abs(w1["xs_and_us"][same]))).max(initial=0.0) <= 1e-9
      # round 2's kernel sums in another order throughout: same optimum; the same path on short solves, up to one iteration
      conv = (w1["status"] == 0) & (r2["status"] == 0)
      np.testing.assert_allclose(w1["cost"][conv], r2["cost"][conv], rtol=1e-8, atol=1e-10, err_msg=str((system, rule, N, B)))
      assert np.array_equal(w1["status"][short], r2["status"][short]), (system, rule, N, B, w1["status"], r2["status"], w1["iters"], r2["iters"])
      assert np.abs(w1["iters"][short].astype(int) - r2["iters"][short].astype(int)).max(initial=0) <= 1, (system, rule, N, B, w1["iters"], r2["iters"])


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("system", _systems())
def test_lane_kernel_agrees_with_the_wavefront_kernel(monkeypatch, system, rule):
  """tools/dev/rocket_probe.py over every system: the lane-per-trajectory kernel against the default (wavefront) path, iteration limits
  0, 2, 8 and a whole solve at N = 6 and 20 -- the check that found the miscompiled lane instantiation of ROCKETLANDING (round 3: refused; round 4:
  the scheduling region of its backward loop is split, hs_solver.h, and it agrees like the others)."""
  if not _collocation_ok(system):
    pytest.skip("collocation is refused for a partially pinned terminal state (reference behaviour)")
  for N in (6, 20):
    for lim in (0, 2, 8, 300):
      w = _solve(monkeypatch, {}, system, rule, N, 3, max_iter=lim)
      l = _solve(monkeypatch, {"MYRIAD_SOLVE_MODE": "lane"}, system, rule, N, 3, max_iter=lim)
      if lim < 300:       # the first iterates agree to rounding
        fin = np.isfinite(w["xs_and_us"]) & np.isfinite(l["xs_and_us"])
        d = np.abs(w["xs_and_us"] - l["xs_and_us"])[fin] / np.maximum(1.0, np.abs(l["xs_and_us"])[fin])
        assert d.max(initial=0.0) <= 1e-7, (system, rule, N, lim, d.max())
        assert np.array_equal(np.isfinite(w["xs_and_us"]), np.isfinite(l["xs_and_us"]))
      else:
        short = (w["iters"] <= LONG) & (w["status"] == 0)
        assert np.array_equal(w["status"][short], l["status"][short]), (system, rule, N, w["status"], l["status"], w["iters"], l["iters"])
        np.testing.assert_allclose(w["cost"][short], l["cost"][short], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("system", _systems())
def test_two_phase_launch_returns_the_bits_of_whole_solves(monkeypatch, system, rule):
  """The two-phase launch of the one-wavefront kernel (hs_solver_fused.h: ParkArgs) is a scheduling matter: parked after k iterations -- LDS and
  loop scalars to global memory -- and resumed in another slot, in another order, under poison, a trajectory returns the bits of a whole solve.
  k = 1 (parked before anything settled), 4 and 9, and every solve that ends before k is simply finished in phase 1."""
  if not _collocation_ok(system):
    pytest.skip("collocation is refused for a partially pinned terminal state (reference behaviour)")
  N, B = 20, 5
  ref = _solve(monkeypatch, {"MYRIAD_FUSED_WAVES": "1", "MYRIAD_PARK_ITER": "0"}, system, rule, N, B)
  for env in ({"MYRIAD_PARK_ITER": "1"}, {"MYRIAD_PARK_ITER": "4", "MYRIAD_POISON": "nan"}, {"MYRIAD_PARK_ITER": "9", "MYRIAD_POISON": "random"}):
    r = _solve(monkeypatch, dict(env, MYRIAD_FUSED_WAVES="1"), system, rule, N, B)
    assert np.array_equal(r["status"], ref["status"]) and np.array_equal(r["iters"], ref["iters"]), (system, rule, env, r["status"], ref["status"], r["iters"], ref["iters"])
    assert r["bits"] == ref["bits"], (system, rule, env)


def _node_solve(monkeypatch, env, N, B, seed):
  from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
  from myriad_amd.trajectory_optimizers import get_optimizer
  for k in KNOBS:
    monkeypatch.delenv(k, raising=False)
  monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_ELASTIC", "0")
  for k, v in env.items():
    monkeypatch.setenv(k, v)
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4,
               intervals=N, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
  x0 = np.clip(0.1 * np.random.default_rng(seed).standard_normal((B, 4)), -2, 2)
  o = opt.solve_batch(x0s=x0, params=opt.system.device_params(), max_iter=300)
  out = {k: np.array(o[k]) for k in ("status", "iters", "cost", "xs_and_us", "lambda", "kkt")}
  out["bits"] = hashlib.sha1(b"".join(np.ascontiguousarray(out[k]).tobytes() for k in ("xs_and_us", "lambda", "cost", "kkt", "status", "iters"))).hexdigest()
  opt.engine.close()
  return out


@pytest.mark.parametrize("N,B", [(20, 3), (100, 12)])
def test_network_kernel_two_phase_launch_returns_the_bits_of_whole_solves(monkeypatch, N, B):
  """the two-phase launch under the four-wavefront network kernel (config 5 at B = 1024 is four rounds of one trajectory per CU): parked after 2, 7 or 12
  iterations, under poison, a trajectory returns the bits of a whole solve"""
  ref = _node_solve(monkeypatch, {"MYRIAD_PARK_ITER": "0"}, N, B, N * 1000 + B)
  for env in ({"MYRIAD_PARK_ITER": "2"}, {"MYRIAD_PARK_ITER": "7", "MYRIAD_POISON": "nan"}, {"MYRIAD_PARK_ITER": "12", "MYRIAD_POISON": "random"}):
    r = _node_solve(monkeypatch, env, N, B, N * 1000 + B)
    assert np.array_equal(r["status"], ref["status"]) and np.array_equal(r["iters"], ref["iters"]), (N, B, env, r["iters"], ref["iters"])
    assert r["bits"] == ref["bits"], (N, B, env)


@pytest.mark.parametrize("N,B", [(10, 1), (10, 12), (20, 3), (50, 12), (100, 3), (100, 128), (100, 300)])
def test_network_dynamics_four_wavefront_kernel(monkeypatch, N, B):
  """tools/dev/node_coop_probe.py: the default kernel of config 5 (four wavefronts share a trajectory: the same cross-wavefront
  exchange as the two-wavefront form of the closed-form systems) -- poison patterns and fresh handles give identical bits, and it takes
  the iterations of round 2's one-wavefront kernel."""
  ref = _node_solve(monkeypatch, {}, N, B, N * 1000 + B)
  assert _node_solve(monkeypatch, {}, N, B, N * 1000 + B)["bits"] == ref["bits"]
  for pat in ("nan", "big", "random"):
    r = _node_solve(monkeypatch, {"MYRIAD_POISON": pat}, N, B, N * 1000 + B)
    assert r["bits"] == ref["bits"], (pat, ref["status"], r["status"], ref["iters"], r["iters"])
  r2 = _node_solve(monkeypatch, {"MYRIAD_SOLVE_MODE": "wave1"}, N, B, N * 1000 + B)
  assert _node_solve(monkeypatch, {"MYRIAD_SOLVE_MODE": "wave1", "MYRIAD_POISON": "random"}, N, B, N * 1000 + B)["bits"] == r2["bits"]
  _same_optimum(ref, r2, ("NODE", N, B))
  assert (ref["iters"] == r2["iters"]).mean() >= 0.9, (ref["iters"], r2["iters"])


@pytest.mark.parametrize("cfg", ["VANDERPOL:1:50", "CANCERTREATMENT:1:100", "SIMPLECASE:10:1", "CARTPOLE:20:2", "VANDERPOL:1:5", "SIMPLECASE:2:3", "BEARPOPULATIONS:2:4"])
@pytest.mark.parametrize("mode", ["wave", "lane"])
def test_shooting_kernels_do_not_compute_with_inherited_registers(monkeypatch, cfg, mode):
  """the register / stack fill of test_inherited_registers_and_stack_do_not_reach_the_result for the shooting kernels (wavefront and lane form), short and
  long horizons (a handful of steps leaves most lanes of the step loops idle)"""
  from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  name, N, cpi = cfg.split(":")
  res = {}
  for pat in ("zero", "nan", "random"):
    for k in KNOBS:
      monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0"); monkeypatch.setenv("MYRIAD_SOLVE_MODE", mode)
    monkeypatch.setenv("MYRIAD_REG_FILL", pat); monkeypatch.setenv("MYRIAD_STACK_FILL", pat)
    hp = HParams(system=SystemType[name], optimizer=OptimizerType.SHOOTING, intervals=int(N), controls_per_interval=int(cpi), nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    x0 = np.tile(opt.system.x_0, (5, 1)) * (1.0 + 0.001 * np.arange(5)[:, None])
    o = opt.solve_batch(x0s=x0, max_iter=60)
    res[pat] = hashlib.sha1(b"".join(np.ascontiguousarray(o[k]).tobytes() for k in ("xs_and_us", "lambda", "cost", "status", "iters"))).hexdigest()
    opt.engine.close()
  assert len(set(res.values())) == 1, res


@pytest.mark.parametrize("N,B", [(6, 3), (20, 12), (100, 64)])
def test_network_kernel_does_not_compute_with_inherited_registers(monkeypatch, N, B):
  """... and for the network kernel (four wavefronts per trajectory, matrix-core passes as functions of their own, helper workgroups at B = 64)"""
  ref = _node_solve(monkeypatch, {"MYRIAD_REG_FILL": "zero", "MYRIAD_STACK_FILL": "zero"}, N, B, N * 1000 + B)
  for pat in ("nan", "random"):
    r = _node_solve(monkeypatch, {"MYRIAD_REG_FILL": pat, "MYRIAD_STACK_FILL": pat}, N, B, N * 1000 + B)
    assert r["bits"] == ref["bits"], (pat, ref["status"], r["status"], ref["iters"], r["iters"])


@pytest.mark.parametrize("cfg", ["VANDERPOL:1:50", "CANCERTREATMENT:1:100", "SIMPLECASE:10:1", "CARTPOLE:20:2"])
def test_shooting_wavefront_kernel_under_poison(monkeypatch, cfg):
  """The shooting wavefront kernel keeps its whole iterate in LDS: poisoned LDS at every hand-over, identical bits."""
  from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  name, N, cpi = cfg.split(":")
  res = {}
  for pat in (None, "nan", "big", "random"):
    for k in KNOBS:
      monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("MYRIAD_SECOND_STARTS", "0")
    if pat:
      monkeypatch.setenv("MYRIAD_POISON", pat)
    hp = HParams(system=SystemType[name], optimizer=OptimizerType.SHOOTING, intervals=int(N), controls_per_interval=int(cpi), nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    x0 = np.tile(opt.system.x_0, (200, 1)) * (1.0 + 0.001 * np.arange(200)[:, None])
    o = opt.solve_batch(x0s=x0, max_iter=300)
    res[pat] = hashlib.sha1(b"".join(np.ascontiguousarray(o[k]).tobytes() for k in ("xs_and_us", "lambda", "cost", "status", "iters"))).hexdigest()
    opt.engine.close()
  assert len(set(res.values())) == 1, res


def test_headline_batch_is_bit_reproducible(monkeypatch):
  """BASELINE config 2 at full size (CARTPOLE Hermite-Simpson N = 100, B = 4096 random x0, the persistent kernel's four resident rounds: every slot
  hands its LDS and scratch from trajectory to trajectory): three fresh handles and the three poison patterns give bit-identical z*, lambda*,
  cost, status and iteration counts -- whatever order the ticket counter deals the trajectories in."""
  import sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  from bench import build_workload
  from myriad_amd import _lib
  N, B = 100, 4096
  x0, z0, lb, ub, T = build_workload(B, N, 2019)
  ref = None
  # ({}: the library's choice for this batch -- the two-phase launch, ten iterations for everybody, then longest-first; MYRIAD_PARK_ITER=0: whole solves)
  for env in ({}, {}, {"MYRIAD_POISON": "nan"}, {"MYRIAD_POISON": "big"}, {"MYRIAD_POISON": "random"}, {"MYRIAD_PARK_ITER": "0"},
              {"MYRIAD_PARK_ITER": "0", "MYRIAD_POISON": "nan"}, {"MYRIAD_PARK_ITER": "14", "MYRIAD_POISON": "random"}):
    for k in KNOBS:
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
    r = eng.solve(z0, lb, ub)
    eng.close()
    assert (r["status"] == 0).all()
    bits = hashlib.sha1(b"".join(np.ascontiguousarray(r[k]).tobytes() for k in ("z", "lam", "cost", "kkt", "status", "iters"))).hexdigest()
    ref = ref or bits
    assert bits == ref, env


@pytest.mark.parametrize("rule", ["HERMITE_SIMPSON", "TRAPEZOIDAL"])
@pytest.mark.parametrize("base", ["PENDULUM", "VANDERPOL", "MOUNTAINCAR", "CARTPOLE", "ROCKETLANDING"])
def test_twin_kernels_do_not_compute_with_inherited_state(monkeypatch, base, rule):
  """The elastic twins on the fused kernel (round 5: block sweep, both wavefront forms; up to 14 variables per point, the largest unrolled blocks of the
  library): the gates
  of the systems above -- poison in LDS and the scratch slots, patterns in every register and in private memory, fresh handles -- on the phase's own
  first problem (the base system's guess and bounds widened by free slacks), a few iterations and a whole twin solve; the lane kernel under the
  register / stack fill as well."""
  from myriad_amd import _lib
  from oracle import myriad_oracle as O
  b = O.SYSTEMS[base]()
  s = O.Elastic(b, 1.0)
  mk = O.hermite_simpson if rule == "HERMITE_SIMPSON" else O.trapezoidal
  for N, lim in ((6 if rule == "HERMITE_SIMPSON" else 9, 4), (20, 40)):
    trb, tr = mk(b, N), mk(s, N)
    u_rows = (tr.guess.size - trb.guess.size) // s.ns
    nx = trb.guess.size - u_rows * b.nu
    def widen(v, fill):
      return np.concatenate([v[:nx], np.hstack([v[nx:].reshape(u_rows, b.nu), np.full((u_rows, s.ns), fill)]).ravel()])
    B = 3
    z0 = np.tile(widen(trb.guess, 0.0), (B, 1)); lb = np.tile(widen(trb.bounds[:, 0], -np.inf), (B, 1)); ub = np.tile(widen(trb.bounds[:, 1], np.inf), (B, 1))
    x0 = z0[:, :b.ns] * (1.0 + 0.01 * np.arange(B)[:, None])
    z0[:, :b.ns] = x0; lb[:, :b.ns] = x0; ub[:, :b.ns] = x0

    def run(env):
      for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
      for k, v in env.items():
        monkeypatch.setenv(k, v)
      eng = _lib.Engine(base + "_ELASTIC", rule, N, s.T)
      o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
      r = eng.solve(z0, lb, ub, params=s.params(), opts=o)
      eng.close()
      return hashlib.sha1(b"".join(np.ascontiguousarray(r[k]).tobytes() for k in ("z", "lam", "cost", "kkt", "status", "iters"))).hexdigest(), r

    for form in ({"MYRIAD_FUSED_WAVES": "1"}, {"MYRIAD_FUSED_WAVES": "2"}, {"MYRIAD_SOLVE_MODE": "lane"}):
      ref, r0 = run(dict(form, MYRIAD_REG_FILL="zero", MYRIAD_STACK_FILL="zero"))
      for pat in ("nan", "random"):
        got, r1 = run(dict(form, MYRIAD_REG_FILL=pat, MYRIAD_STACK_FILL=pat))
        assert got == ref, (form, N, lim, "registers / stack " + pat, r0["status"], r1["status"], r0["iters"], r1["iters"], r0["cost"], r1["cost"])
    for form in ({"MYRIAD_FUSED_WAVES": "1"}, {"MYRIAD_FUSED_WAVES": "2"}):
      ref, r0 = run(form)
      for pz in ("nan", "big", "random"):
        got, r1 = run(dict(form, MYRIAD_POISON=pz))
        assert got == ref, (form, N, lim, "poison " + pz, r0["status"], r1["status"], r0["iters"], r1["iters"])
      assert run(form)[0] == ref
    w1, w2 = run({"MYRIAD_FUSED_WAVES": "1"})[1], run({"MYRIAD_FUSED_WAVES": "2"})[1]      # the two forms: the same steps, merit sums in another order
    assert np.array_equal(w1["status"], w2["status"]) and (lim > 10 or np.array_equal(w1["iters"], w2["iters"]))
    if lim <= 10:
      fin = np.isfinite(w1["z"])
      assert (np.abs(w1["z"] - w2["z"])[fin] / np.maximum(1.0, np.abs(w1["z"])[fin])).max(initial=0.0) <= 1e-7
