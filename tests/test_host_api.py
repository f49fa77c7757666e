"""CPU tests of the host-side mirror of the reference API: HParams derived fields, CLI parsing by enum member name,
guesses/bounds vs the oracle's restatement, error behaviour, sharding + gather with gloo (world_size 2)."""
import os
import sys

import numpy as np
import pytest

from myriad_amd.config import (Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule)
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer, hs_dense_from_blocks
from myriad_amd.useful_scripts import run_setup
from myriad_amd.batched import shard_range

CFG = Config(verbose=False, plot=False)


def test_hparams_post_init_pins():
  """SURVEY.md 8(b) pins of config.py:97-112."""
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, intervals=100)
  assert (hp.controls_per_interval, hp.num_steps, hp.stepsize, hp.state_size, hp.control_size) == (1, 100, 0.02, 4, 1)
  hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50)
  assert (hp.num_steps, hp.stepsize, hp.state_size) == (50, 0.2, 2)
  hp = HParams(system=SystemType.CANCERTREATMENT, max_iter=500)
  assert (hp.num_steps, hp.stepsize, hp.max_iter) == (100, 0.2, 500)
  hp = HParams(system=SystemType.SIMPLECASE, intervals=10)
  assert (hp.num_steps, hp.stepsize) == (1000, 0.001)
  assert hp.minibatch_size == 3
  assert HParams(nlpsolver=NLPSolverType.EXTRAGRADIENT).max_iter == 10000
  assert HParams().nlpsolver == NLPSolverType.IPOPT and HParams().quadrature_rule == QuadratureRule.TRAPEZOIDAL


def test_cli_parses_enums_by_member_name():
  """README.md:82-85 command lines."""
  hp, cfg = run_setup(["--system=CARTPOLE", "--optimizer=COLLOCATION", "--intervals=100"])
  assert hp.system == SystemType.CARTPOLE and hp.quadrature_rule == QuadratureRule.TRAPEZOIDAL   # README:83 is trapezoidal!
  hp, cfg = run_setup(["--system=VANDERPOL", "--optimizer=SHOOTING", "--intervals=1", "--controls_per_interval=50",
                       "--integration_method=RK4", "--nlpsolver=SQP", "--verbose=false", "--hidden_layers", "64", "64"])
  assert hp.integration_method == IntegrationMethod.RK4 and hp.nlpsolver == NLPSolverType.SQP
  assert cfg.verbose is False and hp.hidden_layers == (64, 64)
  with pytest.raises((SystemExit, KeyError)):
    run_setup(["--system=NOPE"])


@pytest.mark.parametrize("sysname,opt,quad,method,kw", [
  ("CARTPOLE", "COLLOCATION", "HERMITE_SIMPSON", "RK4", dict(intervals=100)),
  ("CARTPOLE", "COLLOCATION", "TRAPEZOIDAL", "HEUN", dict(intervals=100)),
  ("VANDERPOL", "SHOOTING", "TRAPEZOIDAL", "HEUN", dict(intervals=1, controls_per_interval=50)),
  ("CANCERTREATMENT", "SHOOTING", "TRAPEZOIDAL", "HEUN", dict(intervals=1, controls_per_interval=100)),
  ("SIMPLECASE", "SHOOTING", "TRAPEZOIDAL", "HEUN", dict(intervals=10, controls_per_interval=100)),
  ("SIMPLECASE", "SHOOTING", "TRAPEZOIDAL", "RK4", dict(intervals=3, controls_per_interval=4)),
])
def test_guess_and_bounds_match_oracle(sysname, opt, quad, method, kw):
  from oracle import myriad_oracle as O
  hp = HParams(system=SystemType[sysname], optimizer=OptimizerType[opt], quadrature_rule=QuadratureRule[quad],
               integration_method=IntegrationMethod[method], **kw)
  o = get_optimizer(hp, CFG, hp.system())
  tr = O.make_transcription(O.SYSTEMS[sysname](), opt, hp.intervals, hp.controls_per_interval, quad, method)
  np.testing.assert_allclose(o.guess, tr.guess, rtol=0, atol=1e-15)
  assert np.array_equal(o.bounds, tr.bounds)
  x, u = o.unravel(o.guess)
  assert x.shape == (tr.x_rows, tr.ns) and u.shape == (tr.u_rows, tr.nu)
  for attr in ("hp", "cfg", "objective", "parametrized_objective", "constraints", "parametrized_constraints", "bounds",
               "guess", "unravel", "require_adj", "x_guess", "u_guess", "x_bounds", "u_bounds", "solve", "solve_with_params"):
    assert hasattr(o, attr), attr


def test_error_behaviour_matches_reference():
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.FBSM)
  with pytest.raises(NotImplementedError):
    get_optimizer(hp, CFG, hp.system())
  from myriad_amd.nlp_solvers import solve
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=4)
  hp.nlpsolver = "bogus"
  with pytest.raises(ValueError):        # nlp_solvers/__init__.py:59-61
    solve(hp, CFG, {'optimizer': None, 'guess': None, 'bounds': None, 'unravel': None})


def test_cartpole_params_from_mapping_takes_abs():
  s = SystemType.CARTPOLE()
  np.testing.assert_allclose(s.params_from_mapping({'g': -9.0, 'm1': 2.0}), [9.0, 2.0, 0.3, 0.5])   # cartpole.py:90-93


def test_block_to_dense_roundtrip_matches_oracle_helper():
  from oracle import myriad_oracle as O
  rng = np.random.default_rng(0)
  N, ns, nu = 3, 4, 1
  blk = rng.standard_normal((N, 5 * ns * ns + 5 * ns * nu))
  assert np.array_equal(hs_dense_from_blocks(blk, N, ns, nu), O.hs_dense_from_blocks(blk, N, ns, nu))


def test_shard_range_covers_everything():
  for total, world in [(4096, 8), (4096, 3), (5, 8), (0, 2)]:
    spans = [shard_range(total, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == total
    assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _gloo_worker(rank, world, port, q):
  import torch
  import torch.distributed as dist
  from myriad_amd.batched import shard_range, gather_solutions
  os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  total = 11                                     # ragged: 6 + 5
  counts = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
  lo, hi = shard_range(total, rank, world)
  idx = torch.arange(lo, hi, dtype=torch.float64)
  local = {"z": idx[:, None] * torch.ones(1, 7, dtype=torch.float64), "cost": idx * 2, "status": (idx % 3).to(torch.int32)}
  out = gather_solutions(local, counts)
  ok = (out["z"].shape == (total, 7) and torch.equal(out["z"][:, 0], torch.arange(total, dtype=torch.float64))
        and torch.equal(out["cost"], 2 * torch.arange(total, dtype=torch.float64))
        and torch.equal(out["status"], (torch.arange(total) % 3).to(torch.int32)))
  t = torch.tensor([1.0 + rank]); dist.all_reduce(t, op=dist.ReduceOp.MAX)      # max-over-ranks timing reduction of bench.py
  q.put((rank, bool(ok), float(t)))
  dist.destroy_process_group()


def test_gather_solutions_gloo_world2():
  """The N>1 path of bench.py / solve_batch on CPU: world_size 2, gloo, ragged shards."""
  import torch.multiprocessing as mp
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = 29500 + (os.getpid() % 2000)
  ps = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
  [p.start() for p in ps]
  res = sorted(q.get(timeout=120) for _ in range(2))
  [p.join(60) for p in ps]
  assert res == [(0, True, 2.0), (1, True, 2.0)]


class _StubEngine:
  """CPU stand-in with the DeviceEngine interface of bench.py: 'solves' by leaving z at the guess and reporting status 0
  for every instance except (global) instance 3, so that the multi-rank bookkeeping of bench.run() is checkable."""

  def __init__(self, N, T, dev, B):
    self.m, self.ngrad, self.jblk = 2 * N * 4, 2 * N + 1, N * 100
    self.B = B

  def solve(self, B, z, lb, ub, lam, cost, status, iters, kkt):
    assert B == self.B == z.shape[0]
    status.zero_(); iters.fill_(7); cost.fill_(1.0); lam.zero_()
    status[(z[:, 0] == self.bad_x0)] = 1

  def eval(self, B, z, fv, gv, cv, jv):
    cv.zero_()

  def timer_reset(self):
    pass

  def timers(self):
    return (1.0, 1), (2.0, 1)


def _bench_worker(rank, world, port, scaling, batch, q):
  import argparse
  import torch
  import torch.distributed as dist
  import bench
  os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  a = argparse.Namespace(gpus=world, steps=2, warmup=1, batch=batch, scaling=scaling, intervals=4, cpu_budget=0.0)
  # the instance to fail: global index 3 of the strong-scaling workload (rank 0's shard) -- its x0[0] identifies it
  _StubEngine.bad_x0 = float(bench.build_workload(batch, 4, 2019)[1][3, 0])
  out = bench.run(a, rank, world, torch.device("cpu"), lambda N, T, dev, B: _StubEngine(N, T, dev, B))
  q.put((rank, out))
  dist.destroy_process_group()


@pytest.mark.parametrize("scaling,batch", [("weak", 6), ("strong", 11)])
def test_bench_run_world2_gloo_with_stub_engine(scaling, batch):
  """bench.py's N>1 logic (sharding per --scaling, gather to rank 0, max-over-ranks time, summed convergence count, the
  JSON line) on two gloo ranks with a stub engine -- what the driver's 8-GPU run exercises over RCCL."""
  import torch.multiprocessing as mp
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = 31500 + (os.getpid() % 2000) + (7 if scaling == "weak" else 0)
  ps = [ctx.Process(target=_bench_worker, args=(r, 2, port, scaling, batch, q)) for r in range(2)]
  [p.start() for p in ps]
  res = dict(q.get(timeout=180) for _ in range(2))
  [p.join(60) for p in ps]
  assert res[1] is None
  out = res[0]
  total = batch * 2 if scaling == "weak" else batch
  assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 2 and out["warmup"] == 1
  assert out["config"]["global_batch"] == total
  assert out["config"]["per_gpu_batch"] == ([6, 6] if scaling == "weak" else [6, 5])
  assert "gloo group of 2 ranks" in out["config"]["parallelism"]
  # exactly one instance (global index 3, on rank 0 in both modes) is reported unconverged in every step
  assert out["converged_fraction"] == pytest.approx((total - 1) / total)
  assert out["value"] == pytest.approx(2 * (total - 1) / (out["ms_per_step"] * 2e-3), rel=1e-9)
  assert out["unit"] == "solves/s" and out["higher_is_better"] is True and out["dtype"] == "f64"
  assert "cpu_baseline" not in out          # rank 0 at N=1 only


def _both_worker(rank, world, port, q):
  import argparse
  import torch
  import torch.distributed as dist
  import bench
  os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  a = argparse.Namespace(gpus=world, steps=2, warmup=1, batch=8, scaling="weak", intervals=4, cpu_budget=0.0, no_other_configs=True)
  _StubEngine.bad_x0 = float("nan")
  out = bench.both_scalings(a, rank, world, torch.device("cpu"), lambda N, T, dev, B: _StubEngine(N, T, dev, B))
  q.put((rank, out))
  dist.destroy_process_group()


def test_bench_prints_the_strong_split_beside_the_weak_line_world2_gloo():
  """N > 1: the contract's line (weak: --batch per GPU) carries the other split of the same batch (strong: --batch in total, SURVEY.md 8(e)'s
  partitioning) measured in the same run -- the driver's SCALE file then holds both under one clock."""
  import torch.multiprocessing as mp
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = 35500 + (os.getpid() % 2000)
  ps = [ctx.Process(target=_both_worker, args=(r, 2, port, q)) for r in range(2)]
  [p.start() for p in ps]
  res = dict(q.get(timeout=180) for _ in range(2))
  [p.join(60) for p in ps]
  assert res[1] is None
  out = res[0]
  assert out["scaling"] == "weak" and out["config"]["global_batch"] == 16 and out["config"]["per_gpu_batch"] == [8, 8]
  o2 = out["other_scaling"]
  assert o2["scaling"] == "strong" and o2["global_batch"] == 8 and o2["per_gpu_batch"] == [4, 4]
  assert o2["unit"] == "solves/s" and o2["value"] > 0 and o2["steps"] == 2


def test_gather_to_one_rank_gloo_world2():
  import torch.multiprocessing as mp
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = 33500 + (os.getpid() % 2000)
  ps = [ctx.Process(target=_gather_dst_worker, args=(r, 2, port, q)) for r in range(2)]
  [p.start() for p in ps]
  res = sorted(q.get(timeout=120) for _ in range(2))
  [p.join(60) for p in ps]
  assert res == [(0, True), (1, True)]


def _gather_dst_worker(rank, world, port, q):
  import torch
  import torch.distributed as dist
  from myriad_amd.batched import shard_range, gather_solutions
  os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  total = 9
  counts = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
  lo, hi = shard_range(total, rank, world)
  idx = torch.arange(lo, hi, dtype=torch.float64)
  out = gather_solutions({"z": idx[:, None].repeat(1, 3), "status": idx.to(torch.int32)}, counts, dst=0)
  if rank == 0:
    ok = torch.equal(out["z"][:, 2], torch.arange(total, dtype=torch.float64)) and torch.equal(out["status"], torch.arange(total, dtype=torch.int32))
  else:
    ok = out == {}
  q.put((rank, bool(ok)))
  dist.destroy_process_group()


class _SecondStartEngine:
  """CPU stand-in for _lib.Engine: `solve` reports MAXITER for every instance whose control guess is all zero (the reference
  guess) and success otherwise -- enough to exercise TrajectoryOptimizer.device_solve / excitation_guess without a GPU."""
  ns, nu = 2, 1

  def __init__(self):
    self.calls = []

  def rollout(self, x0, us, steps, params=None):
    assert us.shape[1] in (steps + 1, 2 * steps + 1)
    xs = np.cumsum(np.concatenate([x0[:, None, :], 0.01 * np.ones((x0.shape[0], steps, x0.shape[1]))], axis=1), axis=1)
    xs[:, 3, 0] = 1e9                       # a rollout that leaves the box: the guess must be clipped into the bounds
    return xs, np.zeros(x0.shape[0])

  def solve(self, z0, lb, ub, params=None, opts=None):
    z0 = np.asarray(z0); B = z0.shape[0]
    self.calls.append(z0.copy())
    nx = z0.shape[1] - self.rows_u
    fail = np.all(z0[:, nx:] == 0.0, axis=1)
    return {"z": z0.copy(), "lam": np.zeros((B, 1)), "cost": np.where(fail, 9.0, 1.0), "status": fail.astype(np.int32),
            "iters": np.full(B, 10, np.int32), "kkt": np.zeros((B, 3))}


def test_restoration_is_the_librarys_business(monkeypatch):
  """Rounds 2-3 ran the elastic phase and the second starts in this Python host; round 4 moved them behind the C-ABI
  (myr_solve_opts.restoration, csrc/myriad_hip.hip: solve_restored; GPU tests: tests/test_gpu_elastic.py,
  tests/test_integration_stub.py).  What is left here: device_solve makes ONE engine call, and asks for a single attempt
  (restoration = 0) when the caller brought an explicit guess."""
  from myriad_amd.config import Config, HParams, IntegrationMethod, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = HParams(system=SystemType.PENDULUM, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               integration_method=IntegrationMethod.HEUN, intervals=6)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  seen = []

  class Opts:
    max_iter = 7; restoration = -1

  class Eng(_SecondStartEngine):
    def solve(self, z0, lb, ub, params=None, opts=None):
      seen.append(opts.restoration)
      return super().solve(z0, lb, ub, params=params, opts=opts)

    def default_opts(self):
      return Opts()

  eng = Eng(); eng.rows_u = opt._u_shape[0]
  opt._engine = eng
  z0, lb, ub = opt.batch_inputs(np.tile(opt.system.x_0, (3, 1)), None)
  r = opt.device_solve(z0, lb, ub, None, None)
  assert len(eng.calls) == 1 and seen == [-1] and list(r["status"]) == [1, 1, 1]      # the stub's statuses, untouched
  opt.device_solve(z0, lb, ub, None, Opts(), second_starts=False)
  assert seen == [-1, 0] and Opts.restoration == -1                                    # (the caller's options are not modified)


class _ShardEngine:
  """CPU stand-in for _lib.Engine that records which rows it was given and on which thread."""

  def __init__(self, tag, fail_on=None):
    self.tag, self.fail_on, self.calls = tag, fail_on, []

  def solve(self, z0, lb, ub, params=None, opts=None):
    import threading
    import time
    z0 = np.asarray(z0); B = z0.shape[0]
    assert lb.shape == z0.shape and ub.shape == z0.shape and (params is None or np.ndim(params) == 1 or len(params) == B)
    self.calls.append((z0[:, 0].copy(), threading.current_thread().name, None if params is None else np.array(params, copy=True)))
    time.sleep(0.05)
    if self.fail_on is not None and (z0[:, 0] == self.fail_on).any():
      raise RuntimeError("device fault")
    return {"z": z0 + 1.0, "lam": np.zeros((B, 2)), "cost": z0[:, 0] * 10.0, "status": np.zeros(B, np.int32),
            "iters": np.full(B, self.tag, np.int32), "kkt": np.zeros((B, 3))}

  def solve_x0(self, x0s, g0, g1, lb, ub, params=None, opts=None):
    """myr_solve_x0's shape: start states + templates (only the first point matters to this stand-in)."""
    x0s = np.asarray(x0s); B, ns = x0s.shape
    z0 = np.tile(np.asarray(g0), (B, 1)); z0[:, :ns] = x0s
    return self.solve(z0, np.broadcast_to(lb, z0.shape), np.broadcast_to(ub, z0.shape), params=params, opts=opts)


def test_fan_out_shards_a_batch_over_engines_in_order():
  """batched.fan_out_solve: contiguous shards, one thread per engine, results concatenated in instance order, per-instance
  parameters follow their rows, an engine listed twice gets both shards (serialised), errors reach the caller."""
  from myriad_amd.batched import fan_out_solve, shard_range
  B, n = 11, 4
  z0 = np.arange(B, dtype=np.float64)[:, None] * np.ones((1, n))
  lb, ub = -np.ones(n), np.ones((B, n))
  p = np.arange(B * 3, dtype=np.float64).reshape(B, 3)
  engs = [_ShardEngine(1), _ShardEngine(2), _ShardEngine(3)]
  res = fan_out_solve(engs, z0, lb, ub, params=p)
  assert np.array_equal(res["z"], z0 + 1.0) and np.array_equal(res["cost"], 10.0 * np.arange(B))
  for r, e in enumerate(engs):
    lo, hi = shard_range(B, r, 3)
    rows, thread, pp = e.calls[0]
    assert np.array_equal(rows, np.arange(lo, hi)) and thread == f"myriad-dev{r}" and np.array_equal(pp, p[lo:hi])
    assert (res["iters"][lo:hi] == r + 1).all()
  one = _ShardEngine(7)
  res = fan_out_solve([one, one], z0, lb, ub, params=p[0])          # `devices=[0, 0]`: two shards queue on one handle
  assert len(one.calls) == 2 and np.array_equal(res["z"], z0 + 1.0) and one.calls[0][2].shape == (3,)
  res = fan_out_solve(engs, z0[:2], lb, ub[:2])                       # fewer instances than engines
  assert res["z"].shape == (2, n)
  with pytest.raises(RuntimeError, match="device fault"):
    fan_out_solve([_ShardEngine(1), _ShardEngine(2, fail_on=9.0)], z0, lb, ub)


def test_solve_batch_fans_out_beneath_the_unchanged_api(monkeypatch):
  """TrajectoryOptimizer.solve_batch with `devices` set: the batch is split over one handle per listed device (stub engines
  here), the result dict keeps the reference's keys plus status / iters / start / attempts, in instance order."""
  from myriad_amd.config import Config, HParams, IntegrationMethod, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               integration_method=IntegrationMethod.RK4, intervals=4)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())

  class E(_ShardEngine):
    def default_opts(self):
      class O: max_iter = 0
      return O()
  e0, e1 = E(1), E(2)
  opt._engine = e0; opt._engines = {0: e0, 1: e1}
  opt.devices = [0, 1]; opt.min_shard = 2
  B = 6
  x0s = np.tile(opt.system.x_0, (B, 1)) + 0.01 * np.arange(B)[:, None]
  out = opt.solve_batch(x0s=x0s)
  assert len(e0.calls) == 1 and len(e1.calls) == 1 and np.allclose(e0.calls[0][0], x0s[:3, 0]) and np.allclose(e1.calls[0][0], x0s[3:, 0])
  assert out["x"].shape == (B, 9, 4) and list(out["iters"]) == [1, 1, 1, 2, 2, 2] and list(out["attempts"]) == [1] * B
  assert set(out) >= {"x", "u", "xs_and_us", "cost", "lambda", "status", "iters", "start", "attempts"}
  opt.devices = [0]
  e0.calls.clear(); e1.calls.clear()
  opt.solve_batch(x0s=x0s)
  assert len(e0.calls) == 1 and e0.calls[0][0].shape[0] == B and not e1.calls


def test_x0_form_takes_the_device_expansion(monkeypatch):
  """solve_batch without an explicit guess hands start states + the guess rule to the engine (myr_solve_x0: no [B][n] arrays on
  the host); the rule reproduces batch_inputs' arrays bit for bit.  MYRIAD_SOLVE_X0=0 and an explicit guess keep the array path."""
  from myriad_amd.config import Config, HParams, IntegrationMethod, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = HParams(system=SystemType.PENDULUM, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               integration_method=IntegrationMethod.HEUN, intervals=6)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  x0_calls = []

  class Eng(_SecondStartEngine):
    def solve_x0(self, x0s, g0, g1, lb, ub, params=None, opts=None):
      x0_calls.append((np.array(x0s), np.array(g0), np.array(g1), np.array(lb), np.array(ub)))
      B = x0s.shape[0]
      return {"z": np.zeros((B, g0.size)), "lam": np.zeros((B, 1)), "cost": np.ones(B), "status": np.zeros(B, np.int32),
              "iters": np.full(B, 10, np.int32), "kkt": np.zeros((B, 3)), "start": np.zeros(B, np.int32), "attempts": np.ones(B, np.int32),
              "restored": np.zeros(B, np.int32)}

    def solve(self, z0, lb, ub, params=None, opts=None):
      r = super().solve(z0, lb, ub, params=params, opts=opts)
      B = r["status"].shape[0]
      r.update(start=np.zeros(B, np.int32), attempts=np.ones(B, np.int32), restored=np.zeros(B, np.int32))
      return r

    def default_opts(self):
      class O: max_iter = 0; restoration = -1
      return O()

  eng = Eng(); eng.rows_u = opt._u_shape[0]
  opt._engine = eng
  rng = np.random.default_rng(0)
  x0s = opt.system.x_0[None] + 0.1 * rng.standard_normal((3, 2))
  res = opt.solve_batch(x0s=x0s)
  assert len(x0_calls) == 1 and np.array_equal(x0_calls[0][0], x0s) and not eng.calls
  z0, lb, ub = opt.batch_inputs(x0s, None)
  g0, g1 = x0_calls[0][1], x0_calls[0][2]
  rows, ns = opt._x_shape
  ze = np.tile(g0, (3, 1)); ze[:, :rows * ns] += np.tile(x0s, (1, rows)) * g1[None, :rows * ns]
  assert np.array_equal(ze, z0)                                    # the rule reproduces the reference guess bit for bit
  assert np.array_equal(x0_calls[0][3], opt.bounds[:, 0]) and np.array_equal(x0_calls[0][4], opt.bounds[:, 1])
  assert list(res["status"]) == [0, 0, 0] and list(res["attempts"]) == [1, 1, 1]
  monkeypatch.setenv("MYRIAD_SOLVE_X0", "0")
  opt.solve_batch(x0s=x0s)
  assert len(x0_calls) == 1 and len(eng.calls) == 1 and np.array_equal(eng.calls[0], z0)
  monkeypatch.delenv("MYRIAD_SOLVE_X0")
  opt.solve_batch(x0s=x0s, guess=z0[0])
  assert len(x0_calls) == 1 and len(eng.calls) == 2
