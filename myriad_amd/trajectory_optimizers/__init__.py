"""Host mirror of /root/reference/myriad/trajectory_optimizers/ (base.py:28-93, __init__.py:12-28,
collocation/hermite_simpson.py:16-81, collocation/trapezoidal.py:16-77, shooting.py:16-77,247-275).

The optimizer objects keep the reference's attribute names (hp, cfg, objective, constraints, bounds, guess, unravel,
x_guess, u_guess, x_bounds, u_bounds, require_adj) and its solve() / solve_with_params(params, guess) methods; the
callables evaluate on the GPU through the C-ABI.  New: solve_batch(x0s, params) for many instances at once."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import os

import numpy as np

from myriad_amd import _lib
from myriad_amd.config import Config, HParams, IntegrationMethod, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.utils import integrate_time_independent


def _state_control_bounds(system, x_rows, u_rows, trap_quirk=False):
  """hermite_simpson.py:55-81 / shooting.py:247-275 / trapezoidal.py:55-77."""
  ns = system.x_0.shape[0]
  nu = system.bounds.shape[0] - ns
  xb = np.zeros((x_rows, ns, 2))
  xb[:, :, :] = system.bounds[:-nu]
  xb[0, :, :] = system.x_0[:, None]
  if system.x_T is not None:
    if trap_quirk:                                     # trapezoidal.py:71 (quirk Q2)
      xb[-nu, :, :] = system.x_T[:, None]
    else:
      for i in range(len(system.x_T)):
        if system.x_T[i] is not None:
          xb[-1, i, :] = system.x_T[i]
  xb = xb.reshape((-1, 2))
  ub = np.empty((u_rows * nu, 2))
  for i in range(nu, 0, -1):                           # control-major blocks (quirk Q1)
    ub[(nu - i) * u_rows:(nu - i + 1) * u_rows] = system.bounds[-i]
  return xb, ub


def _rollout_guess(system, controls, interval_size, intervals, method, rows):
  """shooting.py:56-74 / trapezoidal.py:36-50."""
  rolled = None
  def roll():
    return integrate_time_independent(system.dynamics, system.x_0, controls, interval_size, intervals, method)[1]
  if system.x_T is not None:
    cols = []
    for i in range(len(system.x_T)):
      if system.x_T[i] is not None:
        cols.append(np.linspace(system.x_0[i], system.x_T[i], num=rows).reshape(-1, 1))
      else:
        rolled = roll() if rolled is None else rolled
        cols.append(rolled[:, i].reshape(-1, 1))
    return np.hstack(cols)
  return roll()


class TrajectoryOptimizer(object):
  """trajectory_optimizers/base.py:28-93."""
  require_adj = False

  def __init__(self, hp: HParams, cfg: Config, system, transcription: str, x_guess, u_guess, x_bounds, u_bounds):
    self.hp, self.cfg, self.system = hp, cfg, system
    self.transcription = transcription
    self.x_guess, self.u_guess = x_guess, u_guess
    self.x_bounds, self.u_bounds = x_bounds, u_bounds
    self.guess = np.concatenate([x_guess.ravel(), u_guess.ravel()])      # ravel_pytree((x_guess, u_guess))
    self.bounds = np.vstack((x_bounds, u_bounds))
    self._x_shape, self._u_shape = x_guess.shape, u_guess.shape
    self._engine: Optional[_lib.Engine] = None
    self._engines: Dict[int, _lib.Engine] = {}
    # devices a batch is fanned out over (one handle + one host thread each; `devices=[0, 0]` queues two shards on one GPU).
    # None = every visible device, except under a torch.distributed launch (one process per GPU: that rank's device)
    self.devices = None
    if cfg.verbose:                                                        # base.py:53-63
      print("hp opt type", hp.optimizer)
      print("hp quadrature rule", hp.quadrature_rule)
      print(f"guess.shape = {self.guess.shape}")
      print(f"bounds.shape = {self.bounds.shape}")
    if hp.system.name == "INVASIVEPLANT":                                  # base.py:65-67
      raise NotImplementedError("Discrete systems are not compatible with Trajectory trajectory_optimizers")

  # ---- device engine ------------------------------------------------------------------------------
  def _make_engine(self, device: int) -> _lib.Engine:
    eng = _lib.Engine(self.system.name, self.transcription, self.hp.intervals, self.system.T,
                      controls_per_interval=self.hp.controls_per_interval,
                      integration_method=self.hp.integration_method.name, device=device)
    scale = getattr(self.system, "var_scale", None)
    if scale is not None and os.environ.get("MYRIAD_VAR_SCALE", "1") != "0":
      s = scale()
      if s is not None and np.any(s != 1.0):
        eng.set_var_scale(s)
    return eng

  def _device_list(self):
    """Device ordinals of the fan-out.  OPT-IN: `self.devices` (a list of ordinals, or "all") or MYRIAD_DEVICES ("all" | "0,1,..");
    without either a batch runs on ONE device -- this rank's under a one-process-per-GPU launch (LOCAL_RANK), else device 0 --
    so that an existing caller never finds handles created on GPUs it did not ask for."""
    if self.devices is not None and self.devices != "all":
      return [int(d) for d in self.devices]
    env = os.environ.get("MYRIAD_DEVICES")
    if self.devices == "all" or env == "all":
      return list(range(max(1, _lib.device_count())))
    if env:
      return [int(d) for d in env.replace(",", " ").split()]
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
      return [int(os.environ.get("LOCAL_RANK", "0"))]
    return [0]

  def engines_for(self, B: int):
    """Handles a batch of B instances is sharded over: one per device of `_device_list()`, as many as leave every shard at
    least `min_shard` instances (a launch lasts as long as one solve: smaller shards do not finish sooner)."""
    devs = self._device_list()
    k = max(1, min(len(devs), B // self.min_shard if B >= self.min_shard else 1))
    out = []
    for d in devs[:k]:
      if d not in self._engines:
        self._engines[d] = self.engine if (d == self._primary_device() and self._engine is not None) else self._make_engine(d)
      out.append(self._engines[d])
    return out

  min_shard = 256

  def _primary_device(self) -> int:
    return self._device_list()[0]

  @property
  def engine(self) -> _lib.Engine:
    if self._engine is None:
      d = self._primary_device()
      self._engine = self._engines[d] if d in self._engines else self._make_engine(d)
      self._engines[d] = self._engine
    return self._engine

  def unravel(self, z):
    z = np.asarray(z)
    nx = int(np.prod(self._x_shape))
    return z[..., :nx].reshape(z.shape[:-1] + self._x_shape), z[..., nx:].reshape(z.shape[:-1] + self._u_shape)

  # ---- the reference's callables (evaluated by the hs_eval kernel) ---------------------------------
  def _full_grad(self, g):
    eng = self.engine
    if eng.ngrad == eng.n:
      return g
    out = np.zeros(g.shape[:-1] + (eng.n,))
    out[..., eng.x_rows * eng.ns:] = g
    return out

  def objective(self, variables):
    return float(self.engine.eval(variables, params=self.system.device_params(), want=("f",))["f"][0])

  def parametrized_objective(self, params, variables):
    return float(self.engine.eval(variables, params=self.system.params_from_mapping(params), want=("f",))["f"][0])

  def constraints(self, variables):
    return self.engine.eval(variables, params=self.system.device_params(), want=("c",))["c"][0]

  def parametrized_constraints(self, params, variables):
    return self.engine.eval(variables, params=self.system.params_from_mapping(params), want=("c",))["c"][0]

  def objective_grad(self, variables, params=None):
    p = self.system.device_params() if params is None else self.system.params_from_mapping(params)
    return self._full_grad(self.engine.eval(variables, params=p, want=("gradf",))["gradf"])[0]

  def constraints_jac(self, variables, params=None):
    """Dense Jacobian (what jax.jacrev(constraints) returns) assembled on the host from the kernel's stage blocks."""
    p = self.system.device_params() if params is None else self.system.params_from_mapping(params)
    blk = self.engine.eval(variables, params=p, want=("jblk",))["jblk"][0].reshape(self.hp.intervals, -1)
    N, ns, nu = self.hp.intervals, self.engine.ns, self.engine.nu
    if self.transcription == "HERMITE_SIMPSON":
      return hs_dense_from_blocks(blk, N, ns, nu)
    if self.transcription == "TRAPEZOIDAL":
      return trap_dense_from_blocks(blk, N, ns, nu)
    mc = 2 if self.hp.integration_method == IntegrationMethod.RK4 else 1     # RK4 control rows per step (shooting.py:31)
    return shoot_dense_from_blocks(blk, N, mc * self.hp.controls_per_interval, ns, nu)

  # ---- Lagrangian products (collocation; experiments/e2e_sysid.py:113-141, nlp_solvers/extra_gradient.py:21-33) ----
  def lagrangian(self, variables, lmbdas, params=None):
    """fun(x) + lmbda @ constraint_fun(x)"""
    p = self.system.device_params() if params is None else self.system.params_from_mapping(params)
    r = self.engine.eval(variables, params=p, want=("f", "c"))
    return float(r["f"][0] + np.asarray(lmbdas, dtype=np.float64) @ r["c"][0])

  def lagrangian_grad(self, variables, lmbdas, params=None):
    """jax.grad(lagrangian, argnums=0): grad f + J^T lmbda, applied matrix-free on the device."""
    p = self.system.device_params() if params is None else self.system.params_from_mapping(params)
    return self.engine.vjp(variables, lmbdas, params=p, add_gradf=True)[0]

  def constraints_vjp(self, variables, lmbdas, params=None):
    p = self.system.device_params() if params is None else self.system.params_from_mapping(params)
    return self.engine.vjp(variables, lmbdas, params=p, add_gradf=False)[0]

  def constraints_jvp(self, variables, v, params=None):
    p = self.system.device_params() if params is None else self.system.params_from_mapping(params)
    return self.engine.jvp(variables, v, params=p)[0]

  def extragradient_step(self, variables, lmbdas, eta_x, eta_v, nsteps=1, params=None):
    """`nsteps` iterations of the reference's `step(x, lmbda)` (extra_gradient.py:25-33)."""
    p = self.system.device_params() if params is None else self.system.params_from_mapping(params)
    z, lam = self.engine.exgd(variables, lmbdas, self.bounds[:, 0], self.bounds[:, 1], eta_x, eta_v, nsteps, params=p)
    return z[0], lam[0]

  # ---- what stands in for IPOPT's restoration phase ---------------------------------------------------------------------
  # The batched SQP has no feasibility-restoration phase of its own.  An instance that ends without a KKT point goes through
  # an ELASTIC PHASE (the system's elastic twin, where the library carries one) and then through SECOND STARTS from excitation
  # guesses -- inside the library's solve call (include/myriad_hip.h: myr_solve_opts.restoration, csrc/myriad_hip.hip:
  # solve_restored), so that every binding of the C-ABI gets it; rounds 2-3 had this logic here, in the Python host.
  # MYRIAD_ELASTIC=0 / MYRIAD_SECOND_STARTS=0 switch the two halves off; `start`, `attempts`, `restored` of a result say which
  # start produced each instance (myr_solve_info).

  def _solve_sharded(self, z0, lb, ub, params, opts):
    """One device call per handle of `engines_for(B)`, concurrently (myriad_amd.batched.fan_out_solve)."""
    from myriad_amd.batched import fan_out_solve
    B = 1 if np.ndim(z0) == 1 else np.shape(z0)[0]
    return fan_out_solve(self.engines_for(B), z0, lb, ub, params=params, opts=opts)

  def _solve_sharded_x0(self, x0s, rule, params, opts):
    """The first attempt of a batch that differs in its start states only: guess and bounds are expanded on the device
    (myr_solve_x0) -- B x ns doubles go over PCIe instead of three [B][n] arrays, and no host packing."""
    from myriad_amd.batched import fan_out_solve_x0
    g0, g1 = rule
    return fan_out_solve_x0(self.engines_for(x0s.shape[0]), x0s, g0, g1, self.bounds[:, 0], self.bounds[:, 1], params=params, opts=opts)

  def x0_rule(self):
    """(g0, g1), each [n], when the reference's guess of this transcription is an affine function of the start state --
    z0 = g0 + g1 * tile(x0) on the state rows (include/myriad_hip.h: myr_solve_x0) -- else None (a rollout guess)."""
    return None

  def _linspace_rule(self):
    """linspace(x_0, x_T, rows) per instance as x0 * (1 - lin) + x_T * lin: the arithmetic of batch_inputs, rounding included."""
    rows, ns = self._x_shape
    lin = np.linspace(0.0, 1.0, rows)
    xT = np.asarray(self.system.x_T, dtype=np.float64)
    g0 = np.zeros(self.guess.size); g1 = np.zeros(self.guess.size)
    g0[:rows * ns] = (xT[None, :] * lin[:, None]).ravel()
    g1[:rows * ns] = np.repeat(1 - lin, ns)
    return g0, g1

  def _expand_x0(self, x0s, rule):
    """Host-side expansion of the myr_solve_x0 inputs (for the instances the second starts take up again)."""
    g0, g1 = rule
    rows, ns = self._x_shape
    z0 = np.tile(g0, (x0s.shape[0], 1))
    z0[:, :rows * ns] += (np.tile(x0s, (1, rows)) * g1[None, :rows * ns])
    lb, ub = self._batch_bounds(x0s)
    return z0, lb, ub

  def device_solve(self, z0, lb, ub, params, opts, second_starts=True, x0_form=None):
    """The device solve, fanned out over `engines_for(B)`.  `second_starts=False` (what solve_with_params / solve_batch pass when the
    caller gave an explicit guess) asks the library for ONE attempt from that guess (restoration = 0); otherwise its default applies:
    elastic phase + second starts for the instances the first attempt leaves without a KKT point.  The result says which start
    produced each instance: `start` = 0 for the caller's point, c for the excitation guess with c cycles; `attempts` = device solves
    the instance went through (its `iters` are summed over them); `restored` = 1 when it comes out of the elastic phase."""
    if opts is None:
      opts = self.engine.default_opts()
    import copy
    o = type(opts).from_buffer_copy(opts) if hasattr(type(opts), "from_buffer_copy") else copy.copy(opts)
    if not second_starts:
      o.restoration = 0
    if x0_form is not None:       # (x0s, rule): guess and bounds are expanded on the device
      return self._solve_sharded_x0(x0_form[0], x0_form[1], params, o)
    return self._solve_sharded(z0, lb, ub, params, o)

  # ---- solve ---------------------------------------------------------------------------------------
  def _opt_inputs(self, params=None, guess=None) -> Dict:
    """base.py:69-93: `solve` passes the default-parameter callables, `solve_with_params` closures over
    parametrized_objective / parametrized_constraints (base.py:81-93), so EVERY solver branch sees `params`."""
    if params is None:
      objective, constraints = self.objective, self.constraints
    else:
      objective = lambda variables: self.parametrized_objective(params, variables)
      constraints = lambda variables: self.parametrized_constraints(params, variables)
    return {'objective': objective, 'guess': self.guess if guess is None else np.asarray(guess),
            'constraints': constraints, 'bounds': self.bounds, 'unravel': self.unravel,
            # descriptor for the device solver, which owns the transcription (SURVEY.md 8(b) inner boundary)
            'optimizer': self, 'params_map': params, 'explicit_guess': guess is not None,
            'params': self.system.device_params() if params is None else self.system.params_from_mapping(params)}

  def solve(self) -> Dict[str, np.ndarray]:
    from myriad_amd.nlp_solvers import solve
    return solve(self.hp, self.cfg, self._opt_inputs())

  def solve_with_params(self, params, guess=None) -> Dict[str, np.ndarray]:
    from myriad_amd.nlp_solvers import solve
    return solve(self.hp, self.cfg, self._opt_inputs(params, guess))

  def batch_inputs(self, x0s, params=None):
    """Per-instance (z0, lb, ub) for start states x0s [B,ns], by the reference's own guess / bounds rules."""
    raise NotImplementedError

  def solve_batch(self, x0s=None, params=None, guess=None, max_iter=None, second_starts=None) -> Dict[str, np.ndarray]:
    """EXTENSION (the reference solves one instance per call): B independent instances -- random x0 and/or parameter
    sweeps -- in one call, sharded over `self.devices` (default: every visible GPU; one handle + host thread per device).
    x0s [B,ns] replaces x_0 per instance (bounds row 0 and the guess rule); params [B,np] are per-instance model
    parameters in device order (system.param_names).  Second starts (see device_solve) apply to the reference's guess only:
    with an explicit `guess` the first attempt is returned unless second_starts=True."""
    eng = self.engine
    if x0s is None:
      B = 1 if params is None or np.ndim(params) == 1 else np.shape(params)[0]
      x0s = np.tile(self.system.x_0, (B, 1))
    x0s = np.asarray(x0s, dtype=np.float64)
    p = self.system.device_params() if params is None else np.asarray(params, dtype=np.float64)
    o = eng.default_opts()
    o.max_iter = self.hp.max_iter if max_iter is None else max_iter
    ss = (guess is None) if second_starts is None else bool(second_starts)
    rule = self.x0_rule() if (guess is None and os.environ.get("MYRIAD_SOLVE_X0", "1") != "0") else None
    if rule is not None:       # start states only: guess and bounds are expanded on the device (myr_solve_x0)
      res = self.device_solve(None, None, None, p, o, second_starts=ss, x0_form=(x0s, rule))
    else:
      z0, lb, ub = self.batch_inputs(x0s, p)
      if guess is not None:
        z0 = np.broadcast_to(np.asarray(guess, dtype=np.float64), z0.shape).copy()
      res = self.device_solve(z0, lb, ub, p, o, second_starts=ss)
    x, u = self.unravel(res["z"])
    B = res["status"].shape[0]
    info = {k: res[k] if k in res else (np.ones(B, np.int32) if k == "attempts" else np.zeros(B, np.int32)) for k in ("start", "attempts", "restored")}
    return {'x': x, 'u': u, 'xs_and_us': res["z"], 'cost': res["cost"], 'lambda': res["lam"],
            'status': res["status"], 'iters': res["iters"], 'kkt': res["kkt"], **info}

  def _batch_bounds(self, x0s):
    B, ns = x0s.shape
    lb = np.tile(self.bounds[:, 0], (B, 1)); ub = np.tile(self.bounds[:, 1], (B, 1))
    lb[:, :ns] = x0s; ub[:, :ns] = x0s
    return lb, ub

  def _batch_state_guess(self, x0s, p, rows, controls_rows):
    """shooting.py:56-74 / trapezoidal.py:36-50 per instance: linspace(x0, x_T) where x_T is given, otherwise a coarse
    rollout with zero controls -- done for the whole batch by the rollout kernel."""
    B, ns = x0s.shape
    xT = self.system.x_T
    if xT is not None and all(v is not None for v in xT):
      lin = np.linspace(0.0, 1.0, rows)[None, :, None]
      return x0s[:, None, :] * (1 - lin) + np.asarray(xT, dtype=np.float64)[None, None, :] * lin
    nu = self.u_guess.shape[1]
    xs, _ = self.engine.rollout(x0s, np.zeros((B, controls_rows, nu)), rows - 1, params=p)
    if xT is not None:
      for i, v in enumerate(xT):
        if v is not None:
          xs[:, :, i] = x0s[:, i, None] + (v - x0s[:, i, None]) * np.linspace(0.0, 1.0, rows)[None, :]
    return xs

class HermiteSimpsonCollocationOptimizer(TrajectoryOptimizer):
  """collocation/hermite_simpson.py:16-81 (guess :37-48, bounds :55-81)."""

  def __init__(self, hp: HParams, cfg: Config, system):
    K = 2 * hp.intervals + 1
    ns = system.x_0.shape[0]
    nu = system.bounds.shape[0] - ns
    u_guess = np.zeros((K, nu))
    x_guess = np.linspace(system.x_0, system.x_T, num=K) if system.x_T is not None else np.ones((K, ns)) * 0.1
    xb, ub = _state_control_bounds(system, K, K)
    super().__init__(hp, cfg, system, "HERMITE_SIMPSON", x_guess, u_guess, xb, ub)

  def x0_rule(self):
    if self.system.x_T is not None:
      return self._linspace_rule()
    g0 = np.zeros(self.guess.size)
    g0[:self._x_shape[0] * self._x_shape[1]] = 0.1             # hermite_simpson.py:41: ones * 0.1, whatever x_0 is
    return g0, np.zeros(self.guess.size)

  def batch_inputs(self, x0s, params=None):
    """hermite_simpson.py:41,59 applied per instance."""
    x0s = np.asarray(x0s, dtype=np.float64)
    B, K = x0s.shape[0], 2 * self.hp.intervals + 1
    ns = x0s.shape[1]
    if self.system.x_T is not None:
      lin = np.linspace(0.0, 1.0, K)[None, :, None]
      xs = x0s[:, None, :] * (1 - lin) + self.system.x_T[None, None, :] * lin
    else:
      xs = np.ones((B, K, ns)) * 0.1
    z0 = np.concatenate([xs.reshape(B, -1), np.zeros((B, self.u_guess.size))], axis=1)
    lb, ub = self._batch_bounds(x0s)
    return z0, lb, ub


class TrapezoidalCollocationOptimizer(TrajectoryOptimizer):
  """collocation/trapezoidal.py:16-77 (eval: hs_eval_kernel<.., EVAL_TRAP>, solve: TrapCore in csrc/os_solver.h)."""

  def __init__(self, hp: HParams, cfg: Config, system):
    N = hp.intervals
    ns = system.x_0.shape[0]
    nu = system.bounds.shape[0] - ns
    h = system.T / N
    u_guess = np.zeros((N + 1, nu))
    x_guess = _rollout_guess(system, u_guess, h, N, hp.integration_method, N + 1)
    xb, ub = _state_control_bounds(system, N + 1, N + 1, trap_quirk=True)
    super().__init__(hp, cfg, system, "TRAPEZOIDAL", x_guess, u_guess, xb, ub)

  def x0_rule(self):
    xT = self.system.x_T
    return self._linspace_rule() if (xT is not None and all(v is not None for v in xT)) else None      # else: a rollout guess

  def batch_inputs(self, x0s, params=None):
    N = self.hp.intervals
    xs = self._batch_state_guess(x0s, params, N + 1, N + 1)
    z0 = np.concatenate([xs.reshape(x0s.shape[0], -1), np.zeros((x0s.shape[0], self.u_guess.size))], axis=1)
    lb, ub = self._batch_bounds(x0s)
    return z0, lb, ub


class MultipleShootingOptimizer(TrajectoryOptimizer):
  """shooting.py:16-77,247-275 (eval: hs_eval_kernel<.., EVAL_TRAP>, solve: TrapCore in csrc/os_solver.h)."""

  def __init__(self, hp: HParams, cfg: Config, system, key=None):
    I, cpi = hp.intervals, hp.controls_per_interval
    ns = system.x_0.shape[0]
    nu = system.bounds.shape[0] - ns
    mc = 2 if hp.integration_method == IntegrationMethod.RK4 else 1
    u_guess = np.zeros((mc * I * cpi + 1, nu))
    x_guess = _rollout_guess(system, u_guess[::mc * cpi], system.T / I, I, hp.integration_method, I + 1)
    assert len(x_guess) == I + 1
    xb, ub = _state_control_bounds(system, I + 1, mc * I * cpi + 1)
    super().__init__(hp, cfg, system, "SHOOTING", x_guess, u_guess, xb, ub)
    self._mc = mc

  def x0_rule(self):
    xT = self.system.x_T
    return self._linspace_rule() if (xT is not None and all(v is not None for v in xT)) else None      # else: a rollout guess

  def batch_inputs(self, x0s, params=None):
    I = self.hp.intervals
    if self.system.x_T is not None and all(v is not None for v in self.system.x_T):
      xs = self._batch_state_guess(x0s, params, I + 1, I + 1)
    else:
      # coarse rollout: `intervals` steps of size T/intervals under zero controls (shooting.py:66-74); needs an engine whose
      # step count is `intervals`, i.e. controls_per_interval = 1
      from myriad_amd import _lib
      eng = _lib.Engine(self.system.name, "SHOOTING", I, self.system.T, controls_per_interval=1,
                        integration_method=self.hp.integration_method.name)
      B = x0s.shape[0]
      xs, _ = eng.rollout(x0s, np.zeros((B, self._mc * I + 1, self.u_guess.shape[1])), I, params=params)
      eng.close()
      if self.system.x_T is not None:
        for i, v in enumerate(self.system.x_T):
          if v is not None:
            xs[:, :, i] = x0s[:, i, None] + (v - x0s[:, i, None]) * np.linspace(0.0, 1.0, I + 1)[None, :]
    z0 = np.concatenate([xs.reshape(x0s.shape[0], -1), np.zeros((x0s.shape[0], self.u_guess.size))], axis=1)
    lb, ub = self._batch_bounds(x0s)
    return z0, lb, ub


def get_optimizer(hp: HParams, cfg: Config, system):
  """trajectory_optimizers/__init__.py:12-28."""
  if hp.optimizer == OptimizerType.COLLOCATION:
    if hp.quadrature_rule == QuadratureRule.TRAPEZOIDAL:
      optimizer = TrapezoidalCollocationOptimizer(hp, cfg, system)
    elif hp.quadrature_rule == QuadratureRule.HERMITE_SIMPSON:
      optimizer = HermiteSimpsonCollocationOptimizer(hp, cfg, system)
    else:
      raise KeyError
  elif hp.optimizer == OptimizerType.SHOOTING:
    optimizer = MultipleShootingOptimizer(hp, cfg, system)
  elif hp.optimizer == OptimizerType.FBSM:
    from myriad_amd.trajectory_optimizers.forward_backward_sweep import FBSM
    optimizer = FBSM(hp, cfg, system)
  else:
    raise KeyError
  return optimizer


def hs_dense_from_blocks(blk: np.ndarray, N: int, ns: int, nu: int) -> np.ndarray:
  """Dense (2 N ns) x (K (ns+nu)) Jacobian from the eval kernel's stage blocks (include/myriad_hip.h, myr_eval)."""
  K = 2 * N + 1
  J = np.zeros((2 * N * ns, K * (ns + nu)))
  szs = [ns * ns] * 3 + [ns * nu] * 3 + [ns * ns] * 2 + [ns * nu] * 2
  cuts = np.cumsum(szs)[:-1]
  for k in range(N):
    pr = np.split(blk[k], cuts)
    rd = slice(k * ns, (k + 1) * ns)
    ri = slice(N * ns + k * ns, N * ns + (k + 1) * ns)
    cx = lambda q: slice(q * ns, (q + 1) * ns)
    cu = lambda q: slice(K * ns + q * nu, K * ns + (q + 1) * nu)
    s, m, e = 2 * k, 2 * k + 1, 2 * k + 2
    J[rd, cx(s)] = pr[0].reshape(ns, ns); J[rd, cx(m)] = pr[1].reshape(ns, ns); J[rd, cx(e)] = pr[2].reshape(ns, ns)
    J[rd, cu(s)] = pr[3].reshape(ns, nu); J[rd, cu(m)] = pr[4].reshape(ns, nu); J[rd, cu(e)] = pr[5].reshape(ns, nu)
    J[ri, cx(s)] = pr[6].reshape(ns, ns); J[ri, cx(e)] = pr[7].reshape(ns, ns); J[ri, cx(m)] = np.eye(ns)
    J[ri, cu(s)] = pr[8].reshape(ns, nu); J[ri, cu(e)] = pr[9].reshape(ns, nu)
  return J


def trap_dense_from_blocks(blk: np.ndarray, N: int, ns: int, nu: int) -> np.ndarray:
  """Dense (N ns) x ((N+1)(ns+nu)) trapezoidal Jacobian from the per-interval blocks Cxs, Cxe, Cus, Cue."""
  J = np.zeros((N * ns, (N + 1) * (ns + nu)))
  cuts = np.cumsum([ns * ns, ns * ns, ns * nu])
  for k in range(N):
    a, b, c, d = np.split(blk[k], cuts)
    r = slice(k * ns, (k + 1) * ns)
    J[r, k * ns:(k + 1) * ns] = a.reshape(ns, ns)
    J[r, (k + 1) * ns:(k + 2) * ns] = b.reshape(ns, ns)
    J[r, (N + 1) * ns + k * nu:(N + 1) * ns + (k + 1) * nu] = c.reshape(ns, nu)
    J[r, (N + 1) * ns + (k + 1) * nu:(N + 1) * ns + (k + 2) * nu] = d.reshape(ns, nu)
  return J


def shoot_dense_from_blocks(blk: np.ndarray, I: int, cpi: int, ns: int, nu: int) -> np.ndarray:
  """Dense (I ns) x ((I+1) ns + (I cpi + 1) nu) shooting Jacobian from the per-interval blocks Jx (ns x ns),
  Ju (ns x (cpi+1) nu); d c_k / d x_{k+1} = -I is implied.  `cpi` = control rows per interval minus one (mc * cpi for RK4)."""
  n = (I + 1) * ns + (I * cpi + 1) * nu
  J = np.zeros((I * ns, n))
  for k in range(I):
    r = slice(k * ns, (k + 1) * ns)
    J[r, k * ns:(k + 1) * ns] = blk[k][:ns * ns].reshape(ns, ns)
    J[r, (k + 1) * ns:(k + 2) * ns] = -np.eye(ns)
    u0 = (I + 1) * ns + k * cpi * nu
    J[r, u0:u0 + (cpi + 1) * nu] = blk[k][ns * ns:].reshape(ns, (cpi + 1) * nu)
  return J
