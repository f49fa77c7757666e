"""Host mirror of the hot-path helpers of /root/reference/myriad/utils.py.
The numerics run on the GPU through the C-ABI (myr_rollout); only array plumbing happens here."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from myriad_amd import _lib


def _engine_for(hp, system, device=0):
  from myriad_amd.config import OptimizerType, QuadratureRule
  if hp.optimizer == OptimizerType.COLLOCATION:
    tr = "HERMITE_SIMPSON" if hp.quadrature_rule == QuadratureRule.HERMITE_SIMPSON else "TRAPEZOIDAL"
  else:
    tr = "SHOOTING"
  return _lib.Engine(system.name, tr, hp.intervals, system.T, controls_per_interval=hp.controls_per_interval,
                     integration_method=hp.integration_method.name, device=device)


def get_state_trajectory_and_cost(hp, system, start_state, us, params=None, engine=None) -> Tuple[np.ndarray, float]:
  """utils.py:258-298: integrate [x; cost] of the TRUE dynamics under `us` with hp.integration_method over
  hp.intervals*hp.controls_per_interval steps.  Batched when start_state is [B,ns] / us is [B,rows,nu]."""
  if getattr(system, "discrete", False):
    # DEVIATION: the reference integrates the next-state map of a discrete system as if it were an ODE right-hand side
    # (utils.py:262-298 has no discrete branch), which has no meaning; here the recurrence x_{i+1} = dynamics(x_i, u_i)
    # is applied over the rows of `us` and the running cost summed (host loop: int(T) steps).
    x = np.asarray(start_state, dtype=np.float64)
    xs, c = [x], 0.0
    for u_t in np.asarray(us, dtype=np.float64):
      c += system.cost(x, u_t)
      x = system.dynamics(x, u_t)
      xs.append(x)
    return np.stack(xs), float(c)
  eng = engine or _engine_for(hp, system)
  num_steps = hp.intervals * hp.controls_per_interval
  x0 = np.asarray(start_state, dtype=np.float64)
  us = np.asarray(us, dtype=np.float64)
  single = x0.ndim == 1
  if us.ndim == 1:
    us = us[:, None]
  p = system.device_params() if params is None else params
  xs, cost = eng.rollout(x0, us, num_steps, params=p)
  if engine is None:
    eng.close()
  return (xs[0], float(cost[0])) if single else (xs, cost)


def get_defect(system, learned_xs) -> Optional[np.ndarray]:
  """utils.py:313-324."""
  if system.x_T is None:
    return None
  last = np.asarray(learned_xs)[-1]
  return np.array([last[i] - system.x_T[i] for i in range(len(system.x_T)) if system.x_T[i] is not None])


def integrate_time_independent(dynamics, x_0, interval_us, h, N, integration_method):
  """utils.py:80-131 on the host, for the handful of coarse steps the reference's initial guesses take
  (shooting.py:56-74, trapezoidal.py:36-50).  `interval_us` indexing clamps like jnp (quirk Q6)."""
  name = integration_method if isinstance(integration_method, str) else integration_method.name
  x = np.asarray(x_0, dtype=np.float64)
  us = np.asarray(interval_us, dtype=np.float64)
  g = lambda i: us[min(i, len(us) - 1)]
  out = [x]
  for i in range(N):
    if name == "EULER":
      x = x + h * dynamics(x, g(i))
    elif name == "HEUN":
      k1 = dynamics(x, g(i)); k2 = dynamics(x + h * k1, g(i + 1)); x = x + h / 2 * (k1 + k2)
    elif name == "MIDPOINT":
      x_mid = x + h * dynamics(x, g(i)); x = x + h * dynamics(x_mid, (g(i) + g(i + 1)) / 2)
    elif name == "RK4":
      u1, u2, u3 = g(2 * i), g(2 * i + 1), g(2 * i + 2)
      k1 = dynamics(x, u1); k2 = dynamics(x + h * k1 / 2, u2); k3 = dynamics(x + h * k2 / 2, u2); k4 = dynamics(x + h * k3, u3)
      x = x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    else:
      raise KeyError(name)
    out.append(x)
  return x, np.stack(out)
