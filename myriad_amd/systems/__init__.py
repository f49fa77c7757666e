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
    # a terminal state with unpinned components stays a list with None entries (predator_prey.py:47), as in the reference
    self.x_T = None if x_T is None else (list(x_T) if any(v is None for v in x_T) else np.asarray(x_T, dtype=np.float64))
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

  def var_scale(self) -> np.ndarray:
    """Variable scales [ns+nu] for the device solver (myr_set_var_scale): each state by the magnitude of its start /
    terminal value when that is at least 8, rounded to a power of two so that the change of variables is exact in floating
    point; controls unscaled.  All ones for the BASELINE systems.  (The reference leaves scaling to IPOPT's
    gradient-based `nlp_scaling_method`; the batched SQP has no such pass.)"""
    ns = self.x_0.shape[0]
    s = np.ones(self.bounds.shape[0])
    for i in range(ns):
      a = abs(float(self.x_0[i]))
      if self.x_T is not None and self.x_T[i] is not None and np.isfinite(float(self.x_T[i])):
        a = max(a, abs(float(self.x_T[i])))
      if a >= 8.0:
        s[i] = 2.0 ** np.round(np.log2(a))
    return s

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


# ---- further autonomous systems without terminal cost.  Constructor signatures, x_0 / x_T / T / bounds
# and parameter names are the reference's; dynamics and cost are numpy restatements for host-side use (plots, checks),
# the device has its own generated code (tools/gen_systems.py).
class Bioreactor(IndirectFHCS):
  """systems/lenhart/bioreactor.py:37-83."""
  name = "BIOREACTOR"
  param_names = ("K", "G", "D")

  def __init__(self, K=2., G=1., D=1., M=1., x_0=(.5, .1), T=2.):
    super().__init__(x_0=[x_0[0]], x_T=None, T=T, bounds=[[0., 1.], [0., M]])
    self.K, self.G, self.D, self.M = K, G, D, M

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x = np.asarray(x_t, dtype=np.float64); u = float(np.squeeze(u_t))
    return np.array([self.G * u * x[0] - self.D * x[0] ** 2])

  def cost(self, x_t, u_t, t=None):
    return float(-self.K * np.asarray(x_t)[0] + np.squeeze(u_t))


class Glucose(IndirectFHCS):
  """systems/lenhart/glucose.py:41-104."""
  name = "GLUCOSE"
  param_names = ("a", "b", "c", "A", "l")

  def __init__(self, a=1., b=1., c=1., A=2., l=.5, x_0=(.75, 0.), T=.2):
    super().__init__(x_0=[x_0[0], x_0[1]], x_T=None, T=T, bounds=[[0., 1.], [0., 1.], [0., 0.01]])
    self.a, self.b, self.c, self.A, self.l = a, b, c, A, l

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x0, x1 = x_t
    return np.array([-self.a * x0 - self.b * x1, -self.c * x1 + float(np.squeeze(u_t))])

  def cost(self, x_t, u_t, t=None):
    return float(100_000 * (self.A * (x_t[0] - self.l) ** 2 + np.squeeze(u_t) ** 2))


class MouldFungicide(IndirectFHCS):
  """systems/lenhart/mould_fungicide.py:26-70."""
  name = "MOULDFUNGICIDE"
  param_names = ("r", "M", "A")

  def __init__(self, r=0.3, M=10., A=10., x_0=1.0, T=5):
    super().__init__(x_0=[x_0], x_T=None, T=T, bounds=[[0., 5.], [0., 5.]])
    self.r, self.M, self.A = r, M, A

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x = np.asarray(x_t, dtype=np.float64)
    return self.r * (self.M - x) - np.squeeze(u_t) * x

  def cost(self, x_t, u_t, t=None):
    return float(np.squeeze(self.A * np.asarray(x_t) ** 2 + np.squeeze(u_t) ** 2))


class SimpleCaseWithBounds(IndirectFHCS):
  """systems/lenhart/simple_case_with_bounds.py:24-55."""
  name = "SIMPLECASEWITHBOUNDS"
  param_names = ("A", "C")

  def __init__(self, A=1., C=4., M_1=-1., M_2=2., x_0=1., T=1.):
    super().__init__(x_0=[x_0], x_T=None, T=T, bounds=[[0., 3.], [M_1, M_2]])
    self.A, self.C, self.M_1, self.M_2 = A, C, M_1, M_2

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    return -0.5 * np.asarray(x_t, dtype=np.float64) ** 2 + self.C * np.squeeze(u_t)

  def cost(self, x_t, u_t, t=None):
    return float(np.squeeze(-self.A * np.asarray(x_t) + np.squeeze(u_t) ** 2))


class HIVTreatment(IndirectFHCS):
  """systems/lenhart/hiv_treatment.py:33-111."""
  name = "HIVTREATMENT"
  param_names = ("s", "m_1", "m_2", "m_3", "r", "T_max", "k", "N", "A")

  def __init__(self, s=10., m_1=.02, m_2=.5, m_3=4.4, r=.03, T_max=1500., k=.000024, N=300., x_0=(800., .04, 1.5), A=.05, T=20.):
    super().__init__(x_0=[x_0[0], x_0[1], x_0[2]], x_T=None, T=T, bounds=[[0., 1600.], [0., 100.], [0., 100.], [0., 1.]])
    self.s, self.m_1, self.m_2, self.m_3, self.r, self.T_max, self.k, self.N, self.A = s, m_1, m_2, m_3, r, T_max, k, N, A

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x0, x1, x2 = x_t
    u = float(np.squeeze(u_t))
    return np.array([self.s / (1 + x2) - self.m_1 * x0 + self.r * x0 * (1 - (x0 + x1) / self.T_max) - u * self.k * x0 * x2,
                     u * self.k * x0 * x2 - self.m_2 * x1,
                     self.N * self.m_2 * x1 - self.m_3 * x2])

  def cost(self, x_t, u_t, t=None):
    return float(-self.A * x_t[0] + (1 - np.squeeze(u_t)) ** 2)


class EpidemicSEIRN(IndirectFHCS):
  """systems/lenhart/epidemic_seirn.py:41-95."""
  name = "EPIDEMICSEIRN"
  param_names = ("A", "b", "d", "c", "e", "g", "a")

  def __init__(self, A=.1, b=.525, d=.5, c=.0001, e=.5, g=.1, a=.2, x_0=(1000., 100., 50., 15.), T=20.):
    super().__init__(x_0=[x_0[0], x_0[1], x_0[2], float(np.sum(x_0))], x_T=None, T=T,
                     bounds=[[-np.inf, np.inf]] * 4 + [[0., 0.9]])
    self.A, self.b, self.d, self.c, self.e, self.g, self.a = A, b, d, c, e, g, a

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x0, x1, x2, x3 = x_t
    u = float(np.squeeze(u_t))
    return np.array([self.b * x3 - self.d * x0 - self.c * x0 * x2 - u * x0,
                     self.c * x0 * x2 - (self.e + self.d) * x1,
                     self.e * x1 - (self.g + self.a + self.d) * x2,
                     (self.b - self.d) * x3 - self.a * x2])

  def cost(self, x_t, u_t, t=None):
    return float(self.A * x_t[2] + np.squeeze(u_t) ** 2)


class SEIR(FiniteHorizonControlSystem):
  """systems/miscellaneous/seir.py:44-95 (constants fixed in the constructor, box bounds on the states)."""
  name = "SEIR"
  param_names = ("A", "b", "d", "c", "e", "g", "a")

  def __init__(self):
    self.b, self.d, self.c, self.e, self.g, self.a = 0.525, 0.5, 0.0001, 0.5, 0.1, 0.2
    self.S_0, self.E_0, self.I_0, self.R_0 = 1000.0, 100.0, 50.0, 15.0
    self.N_0 = self.S_0 + self.E_0 + self.I_0 + self.R_0
    self.A, self.M = 0.1, 1000
    super().__init__(x_0=[self.S_0, self.E_0, self.I_0, self.N_0], x_T=None, T=20,
                     bounds=[[0., 2000.], [0., 250.], [0., 250.], [0., 3000.], [0., 1.]])

  dynamics = EpidemicSEIRN.dynamics
  cost = EpidemicSEIRN.cost


class BearPopulations(IndirectFHCS):
  """systems/lenhart/bear_populations.py:38-110 (two controls)."""
  name = "BEARPOPULATIONS"
  param_names = ("r", "K", "m_p", "m_f", "c_p", "c_f")

  def __init__(self, r=.1, K=.75, m_p=.5, m_f=.5, c_p=10_000, c_f=10, x_0=(.4, .2, 0.), T=25):
    super().__init__(x_0=[x_0[0], x_0[1], x_0[2]], x_T=None, T=T, bounds=[[0., 2.], [0., 2.], [0., 2.], [0., .2], [0., .2]])
    self.r, self.K, self.m_p, self.m_f, self.c_p, self.c_f = r, K, m_p, m_f, float(c_p), float(c_f)

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    k, k2 = self.r / self.K, self.r / self.K ** 2
    x0, x1, _ = x_t
    u0, u1 = u_t
    return np.array([self.r * x0 - k * x0 ** 2 + k * self.m_f * (1 - x0 / self.K) * x1 ** 2 - u0 * x0,
                     self.r * x1 - k * x1 ** 2 + k * self.m_p * (1 - x1 / self.K) * x0 ** 2 - u1 * x1,
                     k * (1 - self.m_p) * x0 ** 2 + k * (1 - self.m_f) * x1 ** 2 + k2 * self.m_f * x0 * x1 ** 2 + k2 * self.m_p * x0 ** 2 * x1])

  def cost(self, x_t, u_t, t=None):
    return float(x_t[2] + self.c_p * u_t[0] ** 2 + self.c_f * u_t[1] ** 2)


class Pendulum(FiniteHorizonControlSystem):
  """systems/classical_control/pendulum.py:51-120."""
  name = "PENDULUM"
  param_names = ("g", "m", "length")

  def __init__(self, g: float = 10., m: float = 1., length: float = 1.):
    self.g, self.m, self.length = g, m, length
    self.max_speed, self.max_torque, self.ctrl_penalty = 8., 2., 0.001
    super().__init__(x_0=[0., 0.], x_T=[np.pi, 0.], T=15,
                     bounds=[[-np.pi, np.pi], [-self.max_speed, self.max_speed], [-self.max_torque, self.max_torque]])

  def dynamics(self, x_t, u_t, t=None):
    u = float(np.clip(np.squeeze(u_t), -self.max_torque, self.max_torque))
    theta = ((x_t[0] + np.pi) % (2 * np.pi)) - np.pi
    dot_theta = float(np.clip(x_t[1], -self.max_speed, self.max_speed))
    return np.array([dot_theta, (-3. * self.g / (2. * self.length) * np.sin(theta) + 3. * u / (self.m * self.length ** 2)) * 0.05])

  def cost(self, x_t, u_t, t=None):
    theta = ((x_t[0] + np.pi) % (2 * np.pi)) - np.pi
    return float(theta ** 2 + 0.1 * x_t[1] ** 2 + self.ctrl_penalty * np.squeeze(u_t) ** 2)


class MountainCar(FiniteHorizonControlSystem):
  """systems/classical_control/mountain_car.py:55-101."""
  name = "MOUNTAINCAR"
  param_names = ("power", "gravity")

  def __init__(self, power=0.0015, gravity=0.0025):
    self.min_action, self.max_action = -1.0, 1.0
    self.min_position, self.max_position, self.max_speed = -1.2, 0.6, 0.07
    self.power, self.gravity = power, gravity
    super().__init__(x_0=[-0.1, 0.], x_T=[0.45, 0.], T=300.,
                     bounds=[[self.min_position, self.max_position], [-self.max_speed, self.max_speed], [self.min_action, self.max_action]])

  def dynamics(self, x_t, u_t, t=None):
    force = float(np.clip(np.squeeze(u_t), self.min_action, self.max_action))
    return np.array([x_t[1], force * self.power - self.gravity * x_t[0]])

  def cost(self, x_t, u_t, t=None):
    return float(10. * np.squeeze(u_t) ** 2)


class RocketLanding(FiniteHorizonControlSystem):
  """systems/miscellaneous/rocket_landing.py:55-120."""
  name = "ROCKETLANDING"
  param_names = ("g", "m", "length")

  def __init__(self, g: float = 9.8, m: float = 100_000, length: float = 50, width: float = 10):
    self.g, self.m, self.length, self.width = g, float(m), float(length), width
    self.min_thrust, self.max_thrust = 880 * 1000, 1 * 2210 * 1000
    self.I = 1 / 12 * self.m * self.length ** 2
    self.max_gimble = 20 * 0.01745329
    self.min_gimble = -self.max_gimble
    self.min_percent_thrust, self.max_percent_thrust = 0.4, 1.
    super().__init__(x_0=[0., 0., 1000., -80., -np.pi / 2., 0.], x_T=[0.] * 6, T=16.,
                     bounds=[[-250., 150.], [-250., 150.], [0., 1000.], [-250., 150.], [-2 * np.pi, 2 * np.pi], [-250., 150.],
                             [self.min_percent_thrust, self.max_percent_thrust], [self.min_gimble, self.max_gimble]])

  def dynamics(self, x_t, u_t, t=None):
    theta, thrust, ang = x_t[4], u_t[0], u_t[1]
    F_x = self.max_thrust * thrust * np.sin(ang + theta)
    F_y = self.max_thrust * thrust * np.cos(ang + theta)
    Tq = -self.length / 2 * self.max_thrust * thrust * np.sin(ang)
    return np.array([x_t[1], F_x / self.m, x_t[3], F_y / self.m - self.g, x_t[5], Tq / self.I])

  def cost(self, x_t, u_t, t=None):
    return float(u_t[0] ** 2 + u_t[1] ** 2 + 2 * x_t[5] ** 2)


class Bacteria(IndirectFHCS):
  """systems/lenhart/bacteria.py:33-86 (terminal cost -C x(T))."""
  name = "BACTERIA"
  param_names = ("r", "A", "B", "C")

  def __init__(self, r=1., A=1., B=12., C=1., x_0=1.):
    super().__init__(x_0=[x_0], x_T=None, T=1, bounds=[[0., 10.], [0., 2.]], terminal_cost=True)
    self.adj_T = np.array([C])
    self.r, self.A, self.B, self.C = r, A, B, C

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x = np.asarray(x_t, dtype=np.float64); u = np.squeeze(u_t)
    return self.r * x + self.A * u * x - self.B * u ** 2 * np.exp(-x)

  def cost(self, x_t, u_t, t=None):
    return float(np.squeeze(u_t) ** 2)

  def terminal_cost_fn(self, x_T, u_T, T=None):
    return float(-self.C * np.squeeze(x_T))


class Tumour(FiniteHorizonControlSystem):
  """systems/miscellaneous/tumour.py:52-108 (zero running cost, terminal cost p(T))."""
  name = "TUMOUR"
  param_names = ("xi", "b", "d", "G", "mu")

  def __init__(self, xi=0.084, b=5.85, d=0.00873, G=0.15, mu=0.02):
    self.xi, self.b, self.d, self.G, self.mu = xi, b, d, G, mu
    p_ = ((b - mu) / d) ** (3 / 2)
    super().__init__(x_0=[p_ / 2, p_ / 4, 0.], x_T=None, T=1.2, bounds=[[0., p_], [0., p_], [0., 15.], [0., 75.]], terminal_cost=True)

  def dynamics(self, x_t, u_t, t=None):
    p, q, y = x_t
    u = float(np.squeeze(u_t))
    return np.array([-self.xi * p * np.log(p / q), q * (self.b - (self.mu + self.d * p ** (2 / 3) + self.G * u)), u])

  def cost(self, x_t, u_t, t=None):
    return 0.

  def terminal_cost_fn(self, x_T, u_T, T=None):
    return float(x_T[0])


class Harvest(IndirectFHCS):
  """systems/lenhart/harvest.py:27-62 (running cost with explicit time)."""
  name = "HARVEST"
  param_names = ("A", "k", "m")

  def __init__(self, A=5., k=10., m=.2, M=1., x_0=.4, T=10.):
    super().__init__(x_0=[x_0], x_T=None, T=T, bounds=[[-np.inf, np.inf], [0., M]])
    self.A, self.k, self.m, self.M = A, k, m, M

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    return -(self.m + np.squeeze(u_t)) * np.asarray(x_t, dtype=np.float64)

  def cost(self, x_t, u_t, t=None):
    t = 0.0 if t is None else float(t)
    return float(np.squeeze(-1 * self.A * (self.k * t / (t + 1)) * np.asarray(x_t) * np.squeeze(u_t) + np.squeeze(u_t) ** 2))


class TimberHarvest(IndirectFHCS):
  """systems/lenhart/timber_harvest.py:36-85 (discounted running cost)."""
  name = "TIMBERHARVEST"
  param_names = ("r", "k")

  def __init__(self, r=0., k=1., x_0=100., T=5.):
    super().__init__(x_0=[x_0], x_T=None, T=T, bounds=[[0., 20_000.], [0., 1.]])
    self.r, self.k = r, k

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    return np.array([self.k * x_t[0] * float(np.squeeze(u_t))])

  def cost(self, x_t, u_t, t=None):
    t = 0.0 if t is None else float(t)
    return float(-np.exp(-self.r * t) * x_t[0] * (1 - np.squeeze(u_t)))


class PredatorPrey(IndirectFHCS):
  """systems/lenhart/predator_prey.py:40-137: terminal cost x_0(T) and ONE pinned terminal state (x_T = [None, None, B]).  As in
  the reference only the shooting optimiser and the FBSM secant solver accept it; the collocation optimisers raise
  TypeError on the None entries (trapezoidal.py:71, hermite_simpson.py:41)."""
  name = "PREDATORPREY"
  param_names = ("d_1", "d_2", "A")

  def __init__(self, d_1=.1, d_2=.1, A=1., B=5., guess_a=-.52, guess_b=.5, M=1., x_0=(10., 1., 0.), T=10.):
    super().__init__(x_0=[x_0[0], x_0[1], x_0[2]], x_T=[None, None, B], T=T, bounds=[[0., 11.], [0., 11.], [0., 5.], [0, M]],
                     terminal_cost=True)
    self.adj_T = np.array([1., 0., 0.])
    self.d_1, self.d_2, self.A, self.guess_a, self.guess_b, self.M = d_1, d_2, A, guess_a, guess_b, M

  def dynamics(self, x_t, u_t, v_t=None, t=None):
    x0, x1, _ = x_t
    u = float(np.squeeze(u_t))
    return np.array([(1 - x1) * x0 - self.d_1 * x0 * u, (x0 - 1) * x1 - self.d_2 * x1 * u, u])

  def cost(self, x_t, u_t, t=None):
    return float(self.A * 0.5 * np.squeeze(u_t) ** 2)

  def terminal_cost_fn(self, x_T, u_T, T=None):
    return float(x_T[0])


class InvasivePlant(IndirectFHCS):
  """systems/lenhart/invasive_plant.py:11-94: a DISCRETE-time system (five foci, one removal ratio each).  The direct
  optimisers refuse discrete systems with NotImplementedError (trajectory_optimizers/base.py:66-67); its solver is the
  discrete Forward-Backward Sweep (csrc/fbsm.h, fbsm_discrete_kernel)."""
  name = "INVASIVEPLANT"
  param_names = ("B", "k", "eps")

  def __init__(self, B=1., k=1., eps=.01, x_0=(.5, 1., 1.5, 2., 10.), T=10.):
    super().__init__(x_0=list(x_0), x_T=None, T=T, bounds=[[-np.inf, np.inf]] * 5 + [[0., 1.]] * 5, discrete=True)
    self.adj_T = np.ones(5)                                      # :60
    self.B, self.k, self.eps = B, k, eps

  def dynamics(self, x_t, u_t, v_t=None, t=None):               # :68-72: the NEXT state, not a derivative
    x_t = np.asarray(x_t, dtype=np.float64)
    return (x_t + x_t * self.k / (self.eps + x_t)) * (1 - np.asarray(u_t, dtype=np.float64))

  def cost(self, x_t, u_t, t=None):                              # :74-75
    return float(self.B * (np.asarray(u_t, dtype=np.float64) ** 2).sum())


class SystemType(Enum):
  """systems/__init__.py:29-53: an enum of system classes; calling a member instantiates the system.
  All 21 members of the reference's enum are present."""
  CARTPOLE = CartPole
  VANDERPOL = VanDerPol
  SEIR = SEIR
  TUMOUR = Tumour
  MOUNTAINCAR = MountainCar
  PENDULUM = Pendulum
  SIMPLECASE = SimpleCase
  MOULDFUNGICIDE = MouldFungicide
  BACTERIA = Bacteria
  SIMPLECASEWITHBOUNDS = SimpleCaseWithBounds
  CANCERTREATMENT = CancerTreatment
  EPIDEMICSEIRN = EpidemicSEIRN
  HARVEST = Harvest
  HIVTREATMENT = HIVTreatment
  BEARPOPULATIONS = BearPopulations
  GLUCOSE = Glucose
  TIMBERHARVEST = TimberHarvest
  BIOREACTOR = Bioreactor
  PREDATORPREY = PredatorPrey
  INVASIVEPLANT = InvasivePlant
  ROCKETLANDING = RocketLanding

  def __call__(self, *args, **kwargs):
    return self.value(*args, **kwargs)
