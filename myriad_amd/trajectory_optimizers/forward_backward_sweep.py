"""Host mirror of /root/reference/myriad/trajectory_optimizers/forward_backward_sweep.py:20-116 (class FBSM) and of
IndirectMethodOptimizer (trajectory_optimizers/base.py:106-141): same constructor, attributes and `solve()` result
({'x','u','adj'}), with the sweeps on the GPU (`myr_fbsm`, csrc/fbsm.h).  EXTENSION: `solve_batch` runs B instances
(parameter / start-state sweeps) in one call.

Terminal state conditions go through the secant `sequencesolver` (:118-158, a host loop over device sweeps); discrete
systems (:33-41; INVASIVEPLANT) run the direct recurrences of utils.py:184-188 in `fbsm_discrete_kernel`."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from myriad_amd import _lib
from myriad_amd.config import Config, HParams
from myriad_amd.systems import IndirectFHCS


class IndirectMethodOptimizer(object):
  """trajectory_optimizers/base.py:106-141."""
  require_adj: bool = True

  def __init__(self, hp: HParams, cfg: Config, bounds, guess, unravel):
    self.hp, self.cfg, self.bounds, self.guess, self.unravel = hp, cfg, bounds, guess, unravel

  def solve(self):
    raise NotImplementedError

  def stopping_criterion(self, x_iter, u_iter, adj_iter, delta: float = 0.001) -> bool:
    """base.py:128-141 (kept for callers; the device applies the same rule inside the sweep loop)."""
    (x, old_x), (u, old_u), (adj, old_adj) = x_iter, u_iter, adj_iter
    stop_x = np.abs(x).sum(axis=0) * delta - np.abs(x - old_x).sum(axis=0)
    stop_u = np.abs(u).sum(axis=0) * delta - np.abs(u - old_u).sum(axis=0)
    stop_adj = np.abs(adj).sum(axis=0) * delta - np.abs(adj - old_adj).sum(axis=0)
    return bool(np.min(np.hstack((stop_u, stop_x, stop_adj))) < 0)


class FBSM(IndirectMethodOptimizer):
  """Forward-Backward Sweep Method (Lenhart & Workman), forward_backward_sweep.py:20-116."""

  def __init__(self, hp: HParams, cfg: Config, system: IndirectFHCS):
    if not isinstance(system, IndirectFHCS):
      raise NotImplementedError("FBSM needs adjoint dynamics: an IndirectFHCS system (tests/test_smoke.py:34-37)")
    self.system = system
    self.discrete = bool(getattr(system, "discrete", False))
    # sweep grid (:31-35): fbsm_intervals steps of T/N, or one step per time unit for a discrete system
    self.N, self.h = (int(system.T), 1) if self.discrete else (hp.fbsm_intervals, system.T / hp.fbsm_intervals)
    self.t_interval = np.linspace(0, system.T, num=self.N + 1).reshape(-1, 1)
    ns = system.x_0.shape[0]
    nu = system.bounds.shape[0] - ns
    # the three trajectories the sweeps iterate on (:38-46): zero except the state's first and the adjoint's last row
    rows = {"x": self.N + 1, "u": self.N + (0 if self.discrete else 1), "adj": self.N + 1}
    cols = {"x": ns, "u": nu, "adj": ns}
    tr = {k: np.zeros((rows[k], cols[k])) for k in rows}
    tr["x"][0] = system.x_0
    if system.adj_T is not None:
      tr["adj"][-1] = system.adj_T
    self.x_guess, self.u_guess, self.adj_guess = tr["x"], tr["u"], tr["adj"]
    cuts = np.cumsum([tr[k].size for k in ("x", "u")])
    shapes = [tr[k].shape for k in ("x", "u", "adj")]

    def unravel(v):
      return tuple(part.reshape(sh) for part, sh in zip(np.split(v, cuts), shapes))

    self.x_bounds, self.u_bounds = system.bounds[:-1], system.bounds[-1:]                        # :52-55
    # a pinned terminal state (:58-70, at most one) is met by the secant iteration of sequencesolver() over adj(T) of that state
    pinned = [] if system.x_T is None else [i for i, v in enumerate(system.x_T) if v is not None]
    if len(pinned) > 1:
      raise NotImplementedError("Multiple states with terminal condition not supported yet")
    self.terminal_cdtion = len(pinned) == 1
    if self.terminal_cdtion:
      self.term_cdtion_state = pinned[0]
      self.term_value = float(system.x_T[pinned[0]])
    super().__init__(hp, cfg, np.vstack((self.x_bounds, self.u_bounds)), np.concatenate([tr[k].ravel() for k in ("x", "u", "adj")]), unravel)
    self._engine: Optional[_lib.Engine] = None

  @property
  def engine(self) -> _lib.Engine:
    if self._engine is None:
      # the handle's transcription is irrelevant for myr_fbsm; any valid one creates the system slot
      self._engine = _lib.Engine(self.system.name, "HERMITE_SIMPSON", 1, self.system.T, max_batch=1)
    return self._engine

  def _clip_bounds(self):
    """Per control: the bounds row the system's optim_characterization clips with, and max|bounds[-1]| for the bang-bang
    ones.  Reference rules: bounds[-1] for all single-control systems except SIMPLECASE, which uses bounds[0] -- the
    STATE row, a quirk (simple_case.py:59-62) -- and GLUCOSE, which does not clip (glucose.py:122-126); BEARPOPULATIONS
    clips its two controls with bounds[-2] and bounds[-1] (bear_populations.py:133-142)."""
    b = self.system.bounds
    nu = b.shape[0] - self.system.x_0.shape[0]
    name = self.system.name
    if name == "SIMPLECASE":
      lo, hi = [b[0, 0]], [b[0, 1]]
    elif name == "GLUCOSE":
      lo, hi = [-np.inf], [np.inf]
    elif name == "INVASIVEPLANT":                                # invasive_plant.py:90: bounds[-1] for every control
      lo, hi = [b[-1, 0]] * nu, [b[-1, 1]] * nu
    else:
      lo, hi = [b[-nu + c, 0] for c in range(nu)], [b[-nu + c, 1] for c in range(nu)]
    return np.array(lo, dtype=np.float64), np.array(hi, dtype=np.float64), float(np.max(np.abs(b[-1])))

  def solve_batch(self, x0s=None, params=None, max_sweeps: int = 10000) -> Dict[str, np.ndarray]:
    if x0s is None:
      B = 1 if params is None or np.ndim(params) == 1 else np.shape(params)[0]
      x0s = np.tile(self.system.x_0, (B, 1))
    p = self.system.device_params() if params is None else np.asarray(params, dtype=np.float64)
    lo, hi, bang = self._clip_bounds()
    r = self.engine.fbsm(np.asarray(x0s, dtype=np.float64), self.N, lo, hi, params=p, adj_T=self.system.adj_T, max_sweeps=max_sweeps, bang=bang,
                         discrete=self.discrete)
    return {'x': r['x'], 'u': r['u'], 'adj': r['adj'], 'sweeps': r['sweeps']}

  def _solve_with_adj_T(self, adj_T, max_sweeps):
    lo, hi, bang = self._clip_bounds()
    r = self.engine.fbsm(self.system.x_0[None], self.N, lo, hi, params=self.system.device_params(), adj_T=adj_T,
                         max_sweeps=max_sweeps, bang=bang, discrete=self.discrete)
    return r['x'][0], r['u'][0], r['adj'][0]

  def sequencesolver(self, max_sweeps: int = 10000, max_secant: int = 100) -> Dict[str, np.ndarray]:
    """:118-158: secant method on a = adj(T)[term_cdtion_state] until the terminal state hits its value (|V| <= 1e-10).
    `reinitiate(a)` (:72-86) resets all guesses, so every evaluation is one fresh device sweep sequence with that adj_T.
    `max_secant` guards the reference's uncapped loop."""
    base = np.zeros(self.system.x_0.shape[0]) if self.system.adj_T is None else np.asarray(self.system.adj_T, dtype=np.float64)

    def V(a):
      aT = base.copy(); aT[self.term_cdtion_state] = a
      x, u, adj = self._solve_with_adj_T(aT, max_sweeps)
      return x[-1, self.term_cdtion_state] - self.term_value, (x, u, adj)

    a, b = self.system.guess_a, self.system.guess_b
    Va, sol = V(a)
    Vb, _ = V(b)
    count = 0
    while abs(Va) > 1e-10 and count < max_secant:
      if abs(Va) > abs(Vb):
        a, b = b, a
        Va, Vb = Vb, Va
      d = Va * (b - a) / (Vb - Va)
      b, Vb = a, Va
      a = a - d
      Va, sol = V(a)
      count += 1
    self.x_guess, self.u_guess, self.adj_guess = sol
    self.secant_iterations = count
    return {'x': self.x_guess, 'u': self.u_guess, 'adj': self.adj_guess}

  def solve(self) -> Dict[str, np.ndarray]:
    """:88-116 -- {'x': [N+1,ns], 'u': [N+1,nu] ([N,nu] for a discrete system), 'adj': [N+1,ns]}; the guesses are
    updated like the reference's."""
    if self.terminal_cdtion:
      return self.sequencesolver()
    r = self.solve_batch()
    self.x_guess, self.u_guess, self.adj_guess = r['x'][0], r['u'][0], r['adj'][0]
    return {'x': self.x_guess, 'u': self.u_guess, 'adj': self.adj_guess}
