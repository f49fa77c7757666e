"""Host-side mirror of myriad.systems for the hot path (reference: /root/reference/myriad/systems/).

Only problem DATA lives here (x_0, x_T, T, bounds, parameters, ids); the dynamics/cost arithmetic of the hot
path is the generated device code in csrc/systems_gen.h.  The small numpy `dynamics`/`cost` methods below exist
for host-side plumbing the reference also does on the host (initial-guess rollouts of a handful of steps,
shooting.py:56-74) -- never for the solve itself.
"""
from __future__ import annotations

from enum import Enum
from typing import Optional

import numpy as np


class FiniteHorizonControlSystem:
  """systems/base.py:12-111."""
  name = "BASE"
  param_names = ()
  terminal_cost = False
  discrete = False

  def __init__(self, x_0, x_T, T, bounds, terminal_cost=False, discrete=False):
    self.x_0 = np.asarray(x_0, dtype=np.float64)
    self.x_T = None if x_T is None else np.asarray(x_T, dtype=np.float64)
    self.T = float(T)
    self.bounds = np.asarray(bounds, dtype=np.float64)
    self.terminal_cost = terminal_cost
    self.discrete = discrete

  def dynamics(self, x_t, u_t, t=None):
    raise NotImplementedError

  def parametrized_dynamics(self, params, x_t, u_t, t=None):
    return self.dynamics(x_t, u_t)

  def cost(self, x_t, u_t, t=None):
    raise NotImplementedError

  def parametrized_cost(self, params, x_t, u_t, t=None):
    return self.cost(x_t, u_t, t)

  def terminal_cost_fn(self, x_T, u_T, T=None):
    return 0

  # --- device-side description -------------------------------------------------------------------
  def device_params(self) -> np.ndarray:
    """Parameter vector in the order the generated device code expects (csrc/systems_gen.h)."""
    return np.array([getattr(self, k) for k in self.param_names], dtype=np.float64)

  def params_from_mapping(self, params) -> np.ndarray:
    """`params` mapping of the reference's parametrized_dynamics -> device parameter vector."""
    p = self.device_params()
    for i, k in enumerate(self.param_names):
      if k in params:
        p[i] = float(params[k])
    return p


class IndirectFHCS(FiniteHorizonControlSystem):
  """systems/base.py:125-182: adjoint terminal value and secant guesses; adj_ODE / optim_characterization live on the
  device (csrc/fbsm.h) for the systems on the path."""
  adj_T = None
  guess_a = None
  guess_b = None


class CartPole(FiniteHorizonControlSystem):
  """systems/classical_control/cartpole.py:50-111."""
  name = "CARTPOLE"
  param_names = ("g", "m1", "m2", "length")

  def __init__(self, g: float = 9.81, m1: float = 1., m2: float = .3, length: float = 0.5):
    self.m1, self.m2, self.length, self.g = m1, m2, length, g
    self.u_max, self.d_max, self.d = 20, 2.0, 1.0
    super().__init__(x_0=[0., 0., 0., 0.], x_T=[self.d, np.pi, 0., 0.], T=2.0,
                     bounds=[[-self.d_max, self.d_max], [-2 * np.pi, 2 * np.pi], [-5., 5.], [-10., 10.],
                             [-self.u_max, self.u_max]])

  def params_from_mapping(self, params):
    # parametrized_dynamics takes |params| (cartpole.py:90-93)
    return np.abs(super().params_from_mapping(params))

  def dynamics(self, x_t, u_t, t=None):
    x, theta, dx, dtheta = x_t
    u = float(np.squeeze(u_t))
    s, c = np.sin(theta), np.cos(theta)
    ddx = (self.length * self.m2 * s * dtheta ** 2 + u + self.m2 * self.g * c * s) / (self.m1 + self.m2 * (1 - c ** 2))
    ddtheta = -((self.length * self.m2 * c * dtheta ** 2 + u * c + (self.m1 + self.m2) * self.g * s)
                / (self.length * self.m1 + self.length * self.m2 * (1 - c ** 2)))
    return np.array([dx, dtheta, ddx, ddtheta])

  def cost(self, x_t, u_t, t=None):
    return float(np.squeeze(u_t)) ** 2


class VanDerPol(FiniteHorizonControlSystem):
  """systems/miscellaneous/van_der_pol.py:29-63."""
  name = "VANDERPOL"
  param_names = ("a",)

  def __init__(self, a=1.):
    self.a = a
    super().__init__(x_0=[0., 1.], x_T=np.zeros(2), T=10.0, bounds=[[-4., 4.], [-4., 4.], [-0.75, 1.0]])

  def dynamics(self, x_t, u_t, t=None):
    x0, x1 = x_t
    return np.array([self.a * (1. - x1 ** 2) * x0 - x1 + float(np.squeeze(u_t)), x0])

  def cost(self, x_t, u_t, t=None):
    return float(np.dot(x_t, x_t)) + float(np.squeeze(u_t)) ** 2


class CancerTreatment(IndirectFHCS):
  """systems/lenhart/cancer_treatment.py:40-91."""
  name = "CANCERTREATMENT"
  param_names = ("r", "a", "delta")

  def __init__(self, r=0.3, a=3., delta=0.45, x_0=0.975, T=20):
    super().__init__(x_0=[x_0], x_T=None, T=T, bounds=[[1e-3, 1.], [0., 2.]])
    self.r, self.a, self.delta = r, a, delta

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x_t = np.asarray(x_t, dtype=np.float64)
    return self.r * x_t * np.log(1 / x_t) - np.squeeze(u_t) * self.delta * x_t

  def cost(self, x_t, u_t, t=None):
    return float(np.squeeze(self.a * np.asarray(x_t) ** 2 + np.squeeze(u_t) ** 2))


class SimpleCase(IndirectFHCS):
  """systems/lenhart/simple_case.py:25-62."""
  name = "SIMPLECASE"
  param_names = ("A", "B", "C")

  def __init__(self, A=1., B=1., C=4., x_0=1., T=1.):
    super().__init__(x_0=[x_0], x_T=None, T=T, bounds=[[-np.inf, np.inf], [-np.inf, np.inf]])
    self.A, self.B, self.C = A, B, C

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    return -0.5 * np.asarray(x_t, dtype=np.float64) ** 2 + self.C * np.squeeze(u_t)

  def cost(self, x_t, u_t, t=None):
    return float(np.squeeze(-self.A * np.asarray(x_t) + self.B * np.squeeze(u_t) ** 2))


class SystemType(Enum):
  """systems/__init__.py:29-53: an enum of system classes; calling a member instantiates the system.
  Members outside the hot path (17 further systems) are listed in DESIGN.md as out of scope."""
  CARTPOLE = CartPole
  VANDERPOL = VanDerPol
  SIMPLECASE = SimpleCase
  CANCERTREATMENT = CancerTreatment

  def __call__(self, *args, **kwargs):
    return self.value(*args, **kwargs)
