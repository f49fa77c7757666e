"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (torch fp64 + SciPy) of the reference's trajectory-optimisation
hot path, used ONLY by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg as the checker for the HIP kernels in myriad_amd/csrc/.
Nothing under myriad_amd/ may import this module.

PARITY PINNING STATUS: **parity unpinned at the JAX / IPOPT boundary.**
The reference (nikihowe/myriad) is pure Python on JAX + cyipopt; neither is
installed in the build container or on the GPU box, so the reference itself
cannot be executed to generate vectors.  What IS pinned:
  * the reference's only numeric test, tests/tests.py:19-43 (RK4 of y'=y over
    99 steps equals e to 6 decimals) -- see tests/test_oracle.py;
  * every restated function below follows the cited reference lines one to one
    (jnp -> torch, jax.jacrev/grad -> torch.func.jacrev/grad, lax.scan -> loop);
  * the SciPy SLSQP / trust-constr drivers are the reference's own
    NLPSolverType.SLSQP / TRUST code paths (myriad/nlp_solvers/__init__.py:31-55)
    called with the same keyword set.
The IPOPT branch (nlp_solvers/__init__.py:56-58) is not reproducible here.

All citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

torch.set_default_dtype(torch.float64)
DT = torch.float64


def _t(a) -> torch.Tensor:
  return torch.as_tensor(a, dtype=DT)


# --------------------------------------------------------------------------------------
# L1  systems  (myriad/systems/**)
# --------------------------------------------------------------------------------------
class System:
  """myriad/systems/base.py:12-111 FiniteHorizonControlSystem (fields + method slots)."""
  name = "BASE"
  x_0: np.ndarray
  x_T: Optional[np.ndarray]
  T: float
  bounds: np.ndarray
  terminal_cost: bool = False
  param_names: Tuple[str, ...] = ()

  # vectorised over leading dims: x[..., ns], u[..., nu] -> [..., ns]
  def dynamics(self, x, u):
    raise NotImplementedError

  def cost(self, x, u, t=None):
    raise NotImplementedError

  def terminal_cost_fn(self, x_T, u_T, T=None):                # systems/base.py:101-111
    return 0.0

  def params(self) -> np.ndarray:
    return np.zeros(0)

  @property
  def ns(self):
    return int(self.x_0.shape[0])

  @property
  def nu(self):
    return int(self.bounds.shape[0] - self.x_0.shape[0])


class CartPole(System):
  """myriad/systems/classical_control/cartpole.py:50-111."""
  name = "CARTPOLE"
  param_names = ("g", "m1", "m2", "length")

  def __init__(self, g=9.81, m1=1., m2=.3, length=0.5):
    self.g, self.m1, self.m2, self.length = g, m1, m2, length
    self.u_max, self.d_max, self.d = 20, 2.0, 1.0            # cartpole.py:57-59
    self.x_0 = np.array([0., 0., 0., 0.])                    # :62
    self.x_T = np.array([self.d, np.pi, 0., 0.])             # :63
    self.T = 2.0                                             # :64
    self.bounds = np.array([[-self.d_max, self.d_max],       # :65-71
                            [-2 * np.pi, 2 * np.pi],
                            [-5., 5.],
                            [-10., 10.],
                            [-self.u_max, self.u_max]])

  def params(self):
    return np.array([self.g, self.m1, self.m2, self.length])

  def dynamics(self, x, u):                                  # cartpole.py:76-87
    theta, dx, dtheta = x[..., 1], x[..., 2], x[..., 3]
    u0 = u[..., 0]
    s, c = torch.sin(theta), torch.cos(theta)
    ddx = ((self.length * self.m2 * s * dtheta ** 2 + u0 + self.m2 * self.g * c * s)
           / (self.m1 + self.m2 * (1 - c ** 2)))
    ddtheta = -((self.length * self.m2 * c * dtheta ** 2 + u0 * c + (self.m1 + self.m2) * self.g * s)
                / (self.length * self.m1 + self.length * self.m2 * (1 - c ** 2)))
    return torch.stack([dx, dtheta, ddx, ddtheta], dim=-1)

  def cost(self, x, u, t=None):                              # cartpole.py:106-108
    return u[..., 0] ** 2


class VanDerPol(System):
  """myriad/systems/miscellaneous/van_der_pol.py:29-63."""
  name = "VANDERPOL"
  param_names = ("a",)

  def __init__(self, a=1.):
    self.a = a
    self.x_0 = np.array([0., 1.])
    self.x_T = np.zeros(2)
    self.T = 10.0
    self.bounds = np.array([[-4., 4.], [-4., 4.], [-0.75, 1.0]])

  def params(self):
    return np.array([self.a])

  def dynamics(self, x, u):                                  # van_der_pol.py:46-50
    x0, x1 = x[..., 0], x[..., 1]
    return torch.stack([self.a * (1. - x1 ** 2) * x0 - x1 + u[..., 0], x0], dim=-1)

  def cost(self, x, u, t=None):                              # van_der_pol.py:59-60
    return (x * x).sum(-1) + u[..., 0] ** 2


class CancerTreatment(System):
  """myriad/systems/lenhart/cancer_treatment.py:40-91."""
  name = "CANCERTREATMENT"
  param_names = ("r", "a", "delta")

  def __init__(self, r=0.3, a=3., delta=0.45, x_0=0.975, T=20):
    self.r, self.a, self.delta = r, a, delta
    self.x_0 = np.array([x_0])
    self.x_T = None
    self.T = float(T)
    self.bounds = np.array([[1e-3, 1.], [0., 2.]])

  def params(self):
    return np.array([self.r, self.a, self.delta])

  def dynamics(self, x, u):                                  # cancer_treatment.py:62-65
    return self.r * x * torch.log(1 / x) - u * self.delta * x

  def cost(self, x, u, t=None):                              # cancer_treatment.py:75-76
    return (self.a * x ** 2 + u ** 2)[..., 0]

  # IndirectFHCS members (numpy; used by fbsm())
  adj_T = None

  def np_dynamics(self, x, u):                               # cancer_treatment.py:62-65
    return self.r * x * np.log(1 / x) - u * self.delta * x

  def adj_ODE(self, adj, x, u, t=None):                      # cancer_treatment.py:84-86
    return adj * (self.r + self.delta * u - self.r * np.log(1 / x)) - 2 * self.a * x

  def optim_characterization(self, adj, x, t=None):          # cancer_treatment.py:88-91 (clips with the CONTROL bounds)
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], 0.5 * adj * self.delta * x))


class SimpleCase(System):
  """myriad/systems/lenhart/simple_case.py:25-62."""
  name = "SIMPLECASE"
  param_names = ("A", "B", "C")

  def __init__(self, A=1., B=1., C=4., x_0=1., T=1.):
    self.A, self.B, self.C = A, B, C
    self.x_0 = np.array([x_0])
    self.x_T = None
    self.T = float(T)
    self.bounds = np.array([[-np.inf, np.inf], [-np.inf, np.inf]])

  def params(self):
    return np.array([self.A, self.B, self.C])

  def dynamics(self, x, u):                                  # simple_case.py:46-50
    return -0.5 * x ** 2 + self.C * u

  def cost(self, x, u, t=None):                              # simple_case.py:52-53
    return (-self.A * x + self.B * u ** 2)[..., 0]

  # IndirectFHCS members (numpy; used by fbsm())
  adj_T = None

  def np_dynamics(self, x, u):                               # simple_case.py:46-50
    return -0.5 * x ** 2 + self.C * u

  def adj_ODE(self, adj, x, u, t=None):                      # simple_case.py:55-57 (maximisation-convention adjoint)
    return -self.A + x * adj

  def optim_characterization(self, adj, x, t=None):          # simple_case.py:59-62 (clips with bounds[0], the STATE row)
    return np.minimum(self.bounds[0, 1], np.maximum(self.bounds[0, 0], (self.C * adj) / (2 * self.B)))


# ---- further autonomous systems without terminal cost ------------------------------------------
class Bioreactor(System):
  """myriad/systems/lenhart/bioreactor.py:37-83."""
  name = "BIOREACTOR"
  param_names = ("K", "G", "D")

  def __init__(self, K=2., G=1., D=1., M=1., x_0=(.5, .1), T=2.):
    self.K, self.G, self.D, self.M = K, G, D, M
    self.x_0 = np.array([x_0[0]]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[0., 1.], [0., M]])

  def params(self):
    return np.array([self.K, self.G, self.D])

  def dynamics(self, x, u):                                  # bioreactor.py:60-69
    return torch.stack([self.G * u[..., 0] * x[..., 0] - self.D * x[..., 0] ** 2], dim=-1)

  def cost(self, x, u, t=None):                              # bioreactor.py:82-83
    return -self.K * x[..., 0] + u[..., 0]


  # IndirectFHCS members (numpy; used by fbsm()); x, adj: [..., ns], u: [..., nu]
  adj_T = None

  def np_dynamics(self, x, u):
    return self.G * u * x - self.D * x ** 2

  def adj_ODE(self, adj, x, u, t=None):                      # bioreactor.py:90-94
    return -self.K - self.G * u * adj + 2 * self.D * x * adj

  def optim_characterization(self, adj, x, t=None):          # bioreactor.py:96-101 (bang-bang)
    temp = -1 + self.G * adj[:, :1] * x[:, :1]
    bmax = np.max(np.abs(self.bounds[-1]))
    char = np.sign(temp) * 2 * bmax + bmax
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))

class Glucose(System):
  """myriad/systems/lenhart/glucose.py:41-104."""
  name = "GLUCOSE"
  param_names = ("a", "b", "c", "A", "l")

  def __init__(self, a=1., b=1., c=1., A=2., l=.5, x_0=(.75, 0.), T=.2):
    self.a, self.b, self.c, self.A, self.l = a, b, c, A, l
    self.x_0 = np.array([x_0[0], x_0[1]]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[0., 1.], [0., 1.], [0., 0.01]])

  def params(self):
    return np.array([self.a, self.b, self.c, self.A, self.l])

  def dynamics(self, x, u):                                  # glucose.py:74-84
    return torch.stack([-self.a * x[..., 0] - self.b * x[..., 1], -self.c * x[..., 1] + u[..., 0]], dim=-1)

  def cost(self, x, u, t=None):                              # glucose.py:103-104
    return 100_000 * (self.A * (x[..., 0] - self.l) ** 2 + u[..., 0] ** 2)


  adj_T = None

  def np_dynamics(self, x, u):
    return np.stack([-self.a * x[..., 0] - self.b * x[..., 1], -self.c * x[..., 1] + u[..., 0]], axis=-1)

  def adj_ODE(self, adj, x, u, t=None):                      # glucose.py:114-120
    return np.stack([-2 * self.A * (x[..., 0] - self.l) + adj[..., 0] * self.a, adj[..., 0] * self.b + adj[..., 1] * self.c], axis=-1)

  def optim_characterization(self, adj, x, t=None):          # glucose.py:122-126 (not clipped)
    return (-adj[:, 1] / 2).reshape(-1, 1)

class MouldFungicide(System):
  """myriad/systems/lenhart/mould_fungicide.py:26-70."""
  name = "MOULDFUNGICIDE"
  param_names = ("r", "M", "A")

  def __init__(self, r=0.3, M=10., A=10., x_0=1.0, T=5):
    self.r, self.M, self.A = r, M, A
    self.x_0 = np.array([x_0]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[0., 5.], [0., 5.]])

  def params(self):
    return np.array([self.r, self.M, self.A])

  def dynamics(self, x, u):                                  # mould_fungicide.py:54-58
    return self.r * (self.M - x) - u * x

  def cost(self, x, u, t=None):                              # mould_fungicide.py:69-70
    return (self.A * x ** 2 + u ** 2)[..., 0]


  adj_T = None

  def np_dynamics(self, x, u):
    return self.r * (self.M - x) - u * x

  def adj_ODE(self, adj, x, u, t=None):                      # mould_fungicide.py:72-74
    return adj * (self.r + u) - 2 * self.A * x

  def optim_characterization(self, adj, x, t=None):          # mould_fungicide.py:76-79
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], 0.5 * adj * x))

class SimpleCaseWithBounds(System):
  """myriad/systems/lenhart/simple_case_with_bounds.py:24-55."""
  name = "SIMPLECASEWITHBOUNDS"
  param_names = ("A", "C")

  def __init__(self, A=1., C=4., M_1=-1., M_2=2., x_0=1., T=1.):
    self.A, self.C, self.M_1, self.M_2 = A, C, M_1, M_2
    self.x_0 = np.array([x_0]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[0., 3.], [M_1, M_2]])

  def params(self):
    return np.array([self.A, self.C])

  def dynamics(self, x, u):                                  # simple_case_with_bounds.py:47-51
    return -0.5 * x ** 2 + self.C * u

  def cost(self, x, u, t=None):                              # simple_case_with_bounds.py:53-55
    return (-self.A * x + u ** 2)[..., 0]


  adj_T = None

  def np_dynamics(self, x, u):
    return -0.5 * x ** 2 + self.C * u

  def adj_ODE(self, adj, x, u, t=None):                      # simple_case_with_bounds.py:57-59
    return -self.A + x * adj

  def optim_characterization(self, adj, x, t=None):          # simple_case_with_bounds.py:61-65
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], (self.C * adj) / 2))

class HIVTreatment(System):
  """myriad/systems/lenhart/hiv_treatment.py:33-111."""
  name = "HIVTREATMENT"
  param_names = ("s", "m_1", "m_2", "m_3", "r", "T_max", "k", "N", "A")

  def __init__(self, s=10., m_1=.02, m_2=.5, m_3=4.4, r=.03, T_max=1500., k=.000024, N=300., x_0=(800., .04, 1.5), A=.05, T=20.):
    self.s, self.m_1, self.m_2, self.m_3, self.r, self.T_max, self.k, self.N, self.A = s, m_1, m_2, m_3, r, T_max, k, N, A
    self.x_0 = np.array([x_0[0], x_0[1], x_0[2]]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[0., 1600.], [0., 100.], [0., 100.], [0., 1.]])

  def params(self):
    return np.array([self.s, self.m_1, self.m_2, self.m_3, self.r, self.T_max, self.k, self.N, self.A])

  def dynamics(self, x, u):                                  # hiv_treatment.py:72-84
    x0, x1, x2, u0 = x[..., 0], x[..., 1], x[..., 2], u[..., 0]
    return torch.stack([
      self.s / (1 + x2) - self.m_1 * x0 + self.r * x0 * (1 - (x0 + x1) / self.T_max) - u0 * self.k * x0 * x2,
      u0 * self.k * x0 * x2 - self.m_2 * x1,
      self.N * self.m_2 * x1 - self.m_3 * x2], dim=-1)

  def cost(self, x, u, t=None):                              # hiv_treatment.py:110-111
    return -self.A * x[..., 0] + (1 - u[..., 0]) ** 2


  adj_T = None

  def np_dynamics(self, x, u):
    x0, x1, x2, u0 = x[..., 0], x[..., 1], x[..., 2], u[..., 0]
    return np.stack([self.s / (1 + x2) - self.m_1 * x0 + self.r * x0 * (1 - (x0 + x1) / self.T_max) - u0 * self.k * x0 * x2,
                     u0 * self.k * x0 * x2 - self.m_2 * x1, self.N * self.m_2 * x1 - self.m_3 * x2], axis=-1)

  def adj_ODE(self, adj, x, u, t=None):                      # hiv_treatment.py:117-125
    x0, x1, x2, u0, a0, a1, a2 = x[..., 0], x[..., 1], x[..., 2], u[..., 0], adj[..., 0], adj[..., 1], adj[..., 2]
    return np.stack([
      -self.A + a0 * (self.m_1 - self.r * (1 - (x0 + x1) / self.T_max) + self.r * x0 / self.T_max + u0 * self.k * x2) - a1 * u0 * self.k * x2,
      a0 * self.r * x0 / self.T_max + a1 * self.m_2 - a2 * self.N * self.m_2,
      a0 * (self.s / (1 + x2) ** 2 + u0 * self.k * x0) - a1 * u0 * self.k * x0 + a2 * self.m_3], axis=-1)

  def optim_characterization(self, adj, x, t=None):          # hiv_treatment.py:127-131
    char = (1 + 0.5 * self.k * x[:, 0] * x[:, 2] * (adj[:, 1] - adj[:, 0])).reshape(-1, 1)
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))

class EpidemicSEIRN(System):
  """myriad/systems/lenhart/epidemic_seirn.py:41-95."""
  name = "EPIDEMICSEIRN"
  param_names = ("A", "b", "d", "c", "e", "g", "a")

  def __init__(self, A=.1, b=.525, d=.5, c=.0001, e=.5, g=.1, a=.2, x_0=(1000., 100., 50., 15.), T=20.):
    self.A, self.b, self.d, self.c, self.e, self.g, self.a = A, b, d, c, e, g, a
    self.x_0 = np.array([x_0[0], x_0[1], x_0[2], float(np.sum(x_0))]); self.x_T = None; self.T = float(T)   # :45-50
    self.bounds = np.array([[-np.inf, np.inf]] * 4 + [[0., 0.9]])

  def params(self):
    return np.array([self.A, self.b, self.d, self.c, self.e, self.g, self.a])

  def dynamics(self, x, u):                                  # epidemic_seirn.py:78-92
    x0, x1, x2, x3, u0 = x[..., 0], x[..., 1], x[..., 2], x[..., 3], u[..., 0]
    return torch.stack([
      self.b * x3 - self.d * x0 - self.c * x0 * x2 - u0 * x0,
      self.c * x0 * x2 - (self.e + self.d) * x1,
      self.e * x1 - (self.g + self.a + self.d) * x2,
      (self.b - self.d) * x3 - self.a * x2], dim=-1)

  def cost(self, x, u, t=None):                              # epidemic_seirn.py:94-95
    return self.A * x[..., 2] + u[..., 0] ** 2


  adj_T = None

  def np_dynamics(self, x, u):
    x0, x1, x2, x3, u0 = x[..., 0], x[..., 1], x[..., 2], x[..., 3], u[..., 0]
    return np.stack([self.b * x3 - self.d * x0 - self.c * x0 * x2 - u0 * x0, self.c * x0 * x2 - (self.e + self.d) * x1,
                     self.e * x1 - (self.g + self.a + self.d) * x2, (self.b - self.d) * x3 - self.a * x2], axis=-1)

  def adj_ODE(self, adj, x, u, t=None):                      # epidemic_seirn.py:97-105 (the (d - d) factor is the reference's)
    x0, x2, u0, a0, a1, a2, a3 = x[..., 0], x[..., 2], u[..., 0], adj[..., 0], adj[..., 1], adj[..., 2], adj[..., 3]
    return np.stack([a0 * (self.d + self.c * x2 + u0) - a1 * self.c * x2, a1 * (self.e + self.d) - a2 * self.e,
                     -self.A + a0 * self.c * x0 - a1 * self.c * x0 + a2 * (self.g + self.a + self.d) + a3 * self.a,
                     -self.b * a0 + a3 * (self.d - self.d)], axis=-1)

  def optim_characterization(self, adj, x, t=None):          # epidemic_seirn.py:107-111
    char = (adj[:, 0] * x[:, 0] / 2).reshape(-1, 1)
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))

class SEIR(EpidemicSEIRN):
  """myriad/systems/miscellaneous/seir.py:44-95: the same field with fixed constants and box bounds on the states."""
  name = "SEIR"

  def __init__(self):
    super().__init__()                                        # constants of seir.py:46-58 equal the SEIRN defaults
    self.x_0 = np.array([1000.0, 100.0, 50.0, 1165.0])        # S_0, E_0, I_0, N_0 = S+E+I+R (:52-56)
    self.bounds = np.array([[0., 2000.], [0., 250.], [0., 250.], [0., 3000.], [0., 1.]])   # :72-78


class BearPopulations(System):
  """myriad/systems/lenhart/bear_populations.py:38-110 (two controls)."""
  name = "BEARPOPULATIONS"
  param_names = ("r", "K", "m_p", "m_f", "c_p", "c_f")

  def __init__(self, r=.1, K=.75, m_p=.5, m_f=.5, c_p=10_000, c_f=10, x_0=(.4, .2, 0.), T=25):
    self.r, self.K, self.m_p, self.m_f, self.c_p, self.c_f = r, K, m_p, m_f, float(c_p), float(c_f)
    self.x_0 = np.array([x_0[0], x_0[1], x_0[2]]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[0., 2.], [0., 2.], [0., 2.], [0., .2], [0., .2]])

  def params(self):
    return np.array([self.r, self.K, self.m_p, self.m_f, self.c_p, self.c_f])

  def dynamics(self, x, u):                                  # bear_populations.py:72-86
    k, k2 = self.r / self.K, self.r / self.K ** 2
    x0, x1, u0, u1 = x[..., 0], x[..., 1], u[..., 0], u[..., 1]
    return torch.stack([
      self.r * x0 - k * x0 ** 2 + k * self.m_f * (1 - x0 / self.K) * x1 ** 2 - u0 * x0,
      self.r * x1 - k * x1 ** 2 + k * self.m_p * (1 - x1 / self.K) * x0 ** 2 - u1 * x1,
      k * (1 - self.m_p) * x0 ** 2 + k * (1 - self.m_f) * x1 ** 2 + k2 * self.m_f * x0 * x1 ** 2 + k2 * self.m_p * (x0 ** 2) * x1],
      dim=-1)

  def cost(self, x, u, t=None):                              # bear_populations.py:109-110
    return x[..., 2] + self.c_p * u[..., 0] ** 2 + self.c_f * u[..., 1] ** 2

  adj_T = None

  def np_dynamics(self, x, u):
    k, k2 = self.r / self.K, self.r / self.K ** 2
    x0, x1, u0, u1 = x[..., 0], x[..., 1], u[..., 0], u[..., 1]
    return np.stack([self.r * x0 - k * x0 ** 2 + k * self.m_f * (1 - x0 / self.K) * x1 ** 2 - u0 * x0,
                     self.r * x1 - k * x1 ** 2 + k * self.m_p * (1 - x1 / self.K) * x0 ** 2 - u1 * x1,
                     k * (1 - self.m_p) * x0 ** 2 + k * (1 - self.m_f) * x1 ** 2 + k2 * self.m_f * x0 * x1 ** 2 + k2 * self.m_p * (x0 ** 2) * x1], axis=-1)

  def adj_ODE(self, adj, x, u, t=None):                      # bear_populations.py:117-131
    k, k2 = self.r / self.K, self.r / self.K ** 2
    x0, x1, u0, u1, a0, a1, a2 = x[..., 0], x[..., 1], u[..., 0], u[..., 1], adj[..., 0], adj[..., 1], adj[..., 2]
    return np.stack([
      a0 * (2 * k * x0 + k2 * self.m_f * x1 ** 2 + u0 - self.r) - a1 * (2 * k * self.m_p * (1 - x1 / self.K) * x0)
      + a2 * (2 * k * (self.m_p - 1) * x0 - k2 * self.m_f * x1 ** 2 - 2 * k2 * self.m_p * x0 * x1),
      a1 * (2 * k * x1 + k2 * self.m_p * x0 ** 2 + u1 - self.r) - a0 * (2 * k * self.m_f * (1 - x0 / self.K) * x1)
      + a2 * (2 * k * (self.m_f - 1) * x1 - 2 * k2 * self.m_f * x0 * x1 - k2 * self.m_p * x0 ** 2),
      -np.ones_like(a0)], axis=-1)

  def optim_characterization(self, adj, x, t=None):          # bear_populations.py:133-142
    c0 = np.minimum(self.bounds[-2, 1], np.maximum(self.bounds[-2, 0], (adj[:, 0] * x[:, 0] / (2 * self.c_p)).reshape(-1, 1)))
    c1 = np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], (adj[:, 1] * x[:, 1] / (2 * self.c_f)).reshape(-1, 1)))
    return np.hstack((c0, c1))


def _angle_normalize(x):
  """pendulum.py:17-18 (Python / jnp `%`: result has the sign of the divisor -> torch.remainder)."""
  return torch.remainder(x + math.pi, 2 * math.pi) - math.pi




class Pendulum(System):
  """myriad/systems/classical_control/pendulum.py:51-120 (gym-style clips and angle normalisation kept)."""
  name = "PENDULUM"
  param_names = ("g", "m", "length")

  def __init__(self, g=10., m=1., length=1.):
    self.g, self.m, self.length = g, m, length
    self.max_speed, self.max_torque, self.ctrl_penalty = 8., 2., 0.001
    self.x_0 = np.array([0., 0.]); self.x_T = np.array([np.pi, 0.]); self.T = 15.0
    self.bounds = np.array([[-np.pi, np.pi], [-self.max_speed, self.max_speed], [-self.max_torque, self.max_torque]])

  def params(self):
    return np.array([self.g, self.m, self.length])

  def dynamics(self, x, u):                                  # pendulum.py:94-108
    uu = torch.clamp(u[..., 0], -self.max_torque, self.max_torque)
    theta = _angle_normalize(x[..., 0])
    dot_theta = torch.clamp(x[..., 1], -self.max_speed, self.max_speed)
    ddt = (-3. * self.g / (2. * self.length) * torch.sin(theta) + 3. * uu / (self.m * self.length ** 2)) * 0.05
    return torch.stack([dot_theta, ddt], dim=-1)

  def cost(self, x, u, t=None):                              # pendulum.py:114-120
    return _angle_normalize(x[..., 0]) ** 2 + 0.1 * x[..., 1] ** 2 + self.ctrl_penalty * u[..., 0] ** 2


class MountainCar(System):
  """myriad/systems/classical_control/mountain_car.py:55-101 (hill_function(x) = x^2/2, :11-13)."""
  name = "MOUNTAINCAR"
  param_names = ("power", "gravity")

  def __init__(self, power=0.0015, gravity=0.0025):
    self.power, self.gravity = power, gravity
    self.x_0 = np.array([-0.1, 0.]); self.x_T = np.array([0.45, 0.]); self.T = 300.0
    self.bounds = np.array([[-1.2, 0.6], [-0.07, 0.07], [-1.0, 1.0]])

  def params(self):
    return np.array([self.power, self.gravity])

  def dynamics(self, x, u):                                  # mountain_car.py:83-89; d/dx (x^2/2) = x
    force = torch.clamp(u[..., 0], -1.0, 1.0)
    return torch.stack([x[..., 1], force * self.power - self.gravity * x[..., 0]], dim=-1)

  def cost(self, x, u, t=None):                              # mountain_car.py:100-101
    return 10. * u[..., 0] ** 2


class RocketLanding(System):
  """myriad/systems/miscellaneous/rocket_landing.py:55-120 (six states, two controls)."""
  name = "ROCKETLANDING"
  param_names = ("g", "m", "length")

  def __init__(self, g=9.8, m=100_000, length=50, width=10):
    self.g, self.m, self.length, self.width = g, float(m), float(length), width
    self.max_thrust = 1 * 2210 * 1000
    self.I = 1 / 12 * self.m * self.length ** 2
    mg = 20 * 0.01745329
    self.x_0 = np.array([0., 0., 1000., -80., -np.pi / 2., 0.]); self.x_T = np.zeros(6); self.T = 16.0
    self.bounds = np.array([[-250., 150.], [-250., 150.], [0., 1000.], [-250., 150.], [-2 * np.pi, 2 * np.pi], [-250., 150.],
                            [0.4, 1.], [-mg, mg]])

  def params(self):
    return np.array([self.g, self.m, self.length])

  def dynamics(self, x, u):                                  # rocket_landing.py:99-117
    theta, thrust, ang = x[..., 4], u[..., 0], u[..., 1]
    F_x = self.max_thrust * thrust * torch.sin(ang + theta)
    F_y = self.max_thrust * thrust * torch.cos(ang + theta)
    Tq = -self.length / 2 * self.max_thrust * thrust * torch.sin(ang)
    return torch.stack([x[..., 1], F_x / self.m, x[..., 3], F_y / self.m - self.g, x[..., 5], Tq / self.I], dim=-1)

  def cost(self, x, u, t=None):                              # rocket_landing.py:119-120
    return u[..., 0] ** 2 + u[..., 1] ** 2 + 2 * x[..., 5] ** 2


# ---- systems with a (linear) terminal cost ---------------------------------------------------------------------------
class Bacteria(System):
  """myriad/systems/lenhart/bacteria.py:33-86."""
  name = "BACTERIA"
  param_names = ("r", "A", "B", "C")
  terminal_cost = True

  def __init__(self, r=1., A=1., B=12., C=1., x_0=1.):
    self.r, self.A, self.B, self.C = r, A, B, C
    self.x_0 = np.array([x_0]); self.x_T = None; self.T = 1.0
    self.bounds = np.array([[0., 10.], [0., 2.]])

  def params(self):
    return np.array([self.r, self.A, self.B, self.C])

  def dynamics(self, x, u):                                  # bacteria.py:59-63
    return self.r * x + self.A * u * x - self.B * u ** 2 * torch.exp(-x)

  def cost(self, x, u, t=None):                              # bacteria.py:77-78
    return (u ** 2)[..., 0]

  def terminal_cost_fn(self, x_T, u_T, T=None):              # bacteria.py:84-86
    return -self.C * x_T.squeeze()


  def np_dynamics(self, x, u):
    return self.r * x + self.A * u * x - self.B * u ** 2 * np.exp(-x)

  @property
  def adj_T(self):                                           # bacteria.py:50
    return np.array([self.C])

  def adj_ODE(self, adj, x, u, t=None):                      # bacteria.py:88-90
    return -adj * (self.r + self.A * u + self.B * u ** 2 * np.exp(-x))

  def optim_characterization(self, adj, x, t=None):          # bacteria.py:92-95
    char = adj * self.A * x / (2 * (1 + self.B * adj * np.exp(-x)))
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))

class PredatorPrey(System):
  """myriad/systems/lenhart/predator_prey.py:40-137 (x_T = [None, None, B]: only the third state is pinned)."""
  name = "PREDATORPREY"
  param_names = ("d_1", "d_2", "A")
  terminal_cost = True

  def __init__(self, d_1=.1, d_2=.1, A=1., B=5., guess_a=-.52, guess_b=.5, M=1., x_0=(10., 1., 0.), T=10.):
    self.d_1, self.d_2, self.A, self.M, self.guess_a, self.guess_b = d_1, d_2, A, M, guess_a, guess_b
    self.x_0 = np.array([x_0[0], x_0[1], x_0[2]]); self.x_T = [None, None, B]; self.T = float(T)
    self.bounds = np.array([[0., 11.], [0., 11.], [0., 5.], [0., M]])
    self.adj_T = np.array([1., 0., 0.])                        # :66

  def params(self):
    return np.array([self.d_1, self.d_2, self.A])

  def dynamics(self, x, u):                                  # predator_prey.py:85-96
    x0, x1, u0 = x[..., 0], x[..., 1], u[..., 0]
    return torch.stack([(1 - x1) * x0 - self.d_1 * x0 * u0, (x0 - 1) * x1 - self.d_2 * x1 * u0, u0], dim=-1)

  def cost(self, x, u, t=None):                              # predator_prey.py:113-114
    return self.A * 0.5 * u[..., 0] ** 2

  def terminal_cost_fn(self, x_T, u_T, T=None):              # predator_prey.py:120-122
    return x_T[..., 0]

  def np_dynamics(self, x, u):
    x0, x1, u0 = x[..., 0], x[..., 1], u[..., 0]
    return np.stack([(1 - x1) * x0 - self.d_1 * x0 * u0, (x0 - 1) * x1 - self.d_2 * x1 * u0, u0], axis=-1)

  def adj_ODE(self, adj, x, u, t=None):                      # predator_prey.py:124-130
    a0, a1, x0, x1, u0 = adj[..., 0], adj[..., 1], x[..., 0], x[..., 1], u[..., 0]
    return np.stack([a0 * (x1 - 1 + self.d_1 * u0) - a1 * x1, a0 * x0 + a1 * (1 - x0 + self.d_2 * u0), np.zeros_like(a0)], axis=-1)

  def optim_characterization(self, adj, x, t=None):          # predator_prey.py:132-137
    char = ((adj[:, 0] * self.d_1 * x[:, 0] + adj[:, 1] * self.d_2 * x[:, 1] - adj[:, 2]) / self.A).reshape(-1, 1)
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))


class InvasivePlant:
  """myriad/systems/lenhart/invasive_plant.py:11-94 -- DISCRETE-time: `dynamics` returns the next state and `adj_ODE` the
  previous adjoint.  Only the discrete FBSM applies (the direct optimisers refuse it, trajectory_optimizers/base.py:66-67),
  so it is not in SYSTEMS."""
  name = "INVASIVEPLANT"
  param_names = ("B", "k", "eps")
  discrete = True

  def __init__(self, B=1., k=1., eps=.01, x_0=(.5, 1., 1.5, 2., 10.), T=10.):
    self.B, self.k, self.eps = B, k, eps
    self.x_0 = np.array(x_0, dtype=np.float64); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[-np.inf, np.inf]] * 5 + [[0., 1.]] * 5)
    self.adj_T = np.ones(5)                                    # :60

  def params(self):
    return np.array([self.B, self.k, self.eps])

  def np_dynamics(self, x, u):                               # :68-72
    return (x + x * self.k / (self.eps + x)) * (1 - u)

  def adj_ODE(self, adj, x, u, t=None):                      # :77-81
    return adj * (1 - u) * (1 + self.eps * self.k / (self.eps + x) ** 2)

  def optim_characterization(self, adj, x, t=None):          # :83-90 (rows shifted: adj[1:], x[:-1])
    sa, sx = adj[1:, :], x[:-1, :]
    char = 0.5 * sa / self.B * (sx + sx * self.k / (self.eps + sx))
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))


class Tumour(System):
  """myriad/systems/miscellaneous/tumour.py:52-108 (zero running cost; the objective is the terminal tumour volume)."""
  name = "TUMOUR"
  param_names = ("xi", "b", "d", "G", "mu")
  terminal_cost = True

  def __init__(self, xi=0.084, b=5.85, d=0.00873, G=0.15, mu=0.02):
    self.xi, self.b, self.d, self.G, self.mu = xi, b, d, G, mu
    p_ = ((b - mu) / d) ** (3 / 2)                           # :62 asymptotically stable focus
    self.x_0 = np.array([p_ / 2, p_ / 4, 0.]); self.x_T = None; self.T = 1.2
    self.bounds = np.array([[0., p_], [0., p_], [0., 15.], [0., 75.]])

  def params(self):
    return np.array([self.xi, self.b, self.d, self.G, self.mu])

  def dynamics(self, x, u):                                  # tumour.py:80-86
    pp, q, u0 = x[..., 0], x[..., 1], u[..., 0]
    return torch.stack([-self.xi * pp * torch.log(pp / q), q * (self.b - (self.mu + self.d * pp ** (2 / 3) + self.G * u0)), u0], dim=-1)

  def cost(self, x, u, t=None):                              # tumour.py:100-101
    return torch.zeros_like(u[..., 0])

  def terminal_cost_fn(self, x_T, u_T, T=None):              # tumour.py:106-108
    return x_T[..., 0]


# ---- systems whose running cost depends on time ------------------------------------------------------------------------
class Harvest(System):
  """myriad/systems/lenhart/harvest.py:27-62."""
  name = "HARVEST"
  param_names = ("A", "k", "m")

  def __init__(self, A=5., k=10., m=.2, M=1., x_0=.4, T=10.):
    self.A, self.k, self.m, self.M = A, k, m, M
    self.x_0 = np.array([x_0]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[-np.inf, np.inf], [0., M]])

  def params(self):
    return np.array([self.A, self.k, self.m])

  def dynamics(self, x, u):                                  # harvest.py:54-58
    return -(self.m + u) * x

  def cost(self, x, u, t=None):                              # harvest.py:61-62
    t = torch.zeros((), dtype=DT) if t is None else t
    return -1 * self.A * (self.k * t / (t + 1)) * x[..., 0] * u[..., 0] + u[..., 0] ** 2


  adj_T = None

  def np_dynamics(self, x, u):
    return -(self.m + u) * x

  def adj_ODE(self, adj, x, u, t=None):                      # harvest.py:64-66
    return adj * (self.m + u) - self.A * (self.k * t / (t + 1)) * u

  def optim_characterization(self, adj, x, t=None):          # harvest.py:68-71
    char = 0.5 * x * (self.A * (self.k * t / (t + 1)) - adj)
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))

class TimberHarvest(System):
  """myriad/systems/lenhart/timber_harvest.py:36-85."""
  name = "TIMBERHARVEST"
  param_names = ("r", "k")

  def __init__(self, r=0., k=1., x_0=100., T=5.):
    self.r, self.k = r, k
    self.x_0 = np.array([x_0]); self.x_T = None; self.T = float(T)
    self.bounds = np.array([[0., 20_000.], [0., 1.]])

  def params(self):
    return np.array([self.r, self.k])

  def dynamics(self, x, u):                                  # timber_harvest.py:62-69
    return torch.stack([self.k * x[..., 0] * u[..., 0]], dim=-1)

  def cost(self, x, u, t=None):                              # timber_harvest.py:84-85
    t = torch.zeros((), dtype=DT) if t is None else t
    return -torch.exp(-self.r * t) * x[..., 0] * (1 - u[..., 0])


  adj_T = None

  def np_dynamics(self, x, u):
    return self.k * x * u

  def adj_ODE(self, adj, x, u, t=None):                      # timber_harvest.py:87-92
    return u * (np.exp(-self.r * t) - self.k * adj) - np.exp(-self.r * t)

  def optim_characterization(self, adj, x, t=None):          # timber_harvest.py:94-103 (bang-bang)
    temp = x[:, :1] * (self.k * adj[:, :1] - np.exp(-self.r * t[:, :1]))
    bmax = np.max(np.abs(self.bounds[-1]))
    char = np.sign(temp) * 2 * bmax + bmax
    return np.minimum(self.bounds[-1, 1], np.maximum(self.bounds[-1, 0], char))

class NodeCartPole(CartPole):
  """myriad/systems/neural_ode/node_system.py:14-42 over CARTPOLE: dynamics = net.apply(params, [x;u]) with the MLP of
  myriad/neural_ode/create_node.py:110-117 (Linear+sigmoid per hidden layer, Linear out; Haiku y = x @ w + b);
  cost of the true system.  `params`: {'linear': {'w','b'}, 'linear_1': ..., 'linear_2': ...}."""
  name = "NODE_CARTPOLE"

  def __init__(self, params):
    super().__init__()
    self.mlp = [( _t(params[k]["w"]), _t(params[k]["b"])) for k in ("linear", "linear_1", "linear_2")]

  def dynamics(self, x, u):
    h = torch.cat([x, u], dim=-1)
    for w, b in self.mlp[:-1]:
      h = torch.sigmoid(h @ w + b)
    w, b = self.mlp[-1]
    return h @ w + b


class Elastic(System):
  """The ELASTIC TWIN of a system (no counterpart in the reference; include/myriad_hip.h MYR_SYS_*_ELASTIC): x' = f(x,u) + s with
  ns free slack controls s appended to u, running cost g + rho/2 |s|^2.  The checker of the twins' device code."""

  def __init__(self, base: System, rho: float = 1.0):
    self.base, self.rho = base, float(rho)
    self.name = base.name + "_ELASTIC"
    self.param_names = tuple(base.param_names) + ("rho",)
    self.x_0, self.x_T, self.T = base.x_0, base.x_T, base.T
    self.bounds = np.vstack([base.bounds, np.tile([[-np.inf, np.inf]], (base.ns, 1))])
    self._nu = base.nu

  def params(self):
    return np.concatenate([self.base.params(), [self.rho]])

  def dynamics(self, x, u):
    return self.base.dynamics(x, u[..., :self._nu]) + u[..., self._nu:]

  def cost(self, x, u, t=None):
    return self.base.cost(x, u[..., :self._nu], t) + 0.5 * self.rho * (u[..., self._nu:] ** 2).sum(-1)


SYSTEMS = {c.name: c for c in (CartPole, VanDerPol, CancerTreatment, SimpleCase, Bioreactor, Glucose, MouldFungicide,
                                SimpleCaseWithBounds, HIVTreatment, EpidemicSEIRN, SEIR, BearPopulations, Pendulum, MountainCar,
                                RocketLanding, Bacteria, Tumour, Harvest, TimberHarvest, PredatorPrey)}


# --------------------------------------------------------------------------------------
# L2  integrators  (myriad/utils.py:22-134)
# --------------------------------------------------------------------------------------
def _step(dyn, method: str, x, us, idx, h, t=None, ts=None):
  """One step of utils.py:31-54 / :90-112 (time argument optional)."""
  def f(xx, uu, tt):
    return dyn(xx, uu, tt) if ts is not None else dyn(xx, uu)
  def g(i):  # jnp gather semantics: out-of-range indices clamp (quirk Q6: tests/tests.py:37 relies on it)
    return us[min(i, us.shape[0] - 1)]
  tt = ts[idx] if ts is not None else None
  if method == "EULER":
    return x + h * f(x, g(idx), tt)
  if method == "HEUN":
    k1 = f(x, g(idx), tt)
    k2 = f(x + h * k1, g(idx + 1), None if tt is None else tt + h)
    return x + h / 2 * (k1 + k2)
  if method == "MIDPOINT":
    x_mid = x + h * f(x, g(idx), tt)
    u_mid = (g(idx) + g(idx + 1)) / 2
    return x + h * f(x_mid, u_mid, None if tt is None else tt + h / 2)
  if method == "RK4":
    u1, u2, u3 = g(2 * idx), g(2 * idx + 1), g(2 * idx + 2)
    k1 = f(x, u1, tt)
    k2 = f(x + h * k1 / 2, u2, None if tt is None else tt + h / 2)
    k3 = f(x + h * k2 / 2, u2, None if tt is None else tt + h / 2)   # u2 for k2 AND k3 (quirk Q5)
    k4 = f(x + h * k3, u3, None if tt is None else tt + h)
    return x + h / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
  raise KeyError(method)


def integrate(dynamics_t, x_0, interval_us, h, N, ts, method: str):
  """utils.py:22-73 (time-dependent).  Returns (x_T, states[N+1, ...])."""
  x = x_0
  out = [x_0]
  for i in range(N):
    x = _step(dynamics_t, method, x, interval_us, i, h, ts=ts)
    out.append(x)
  return x, torch.stack(out)


def integrate_time_independent(dynamics, x_0, interval_us, h, N, method: str):
  """utils.py:80-131."""
  x = x_0
  out = [x_0]
  for i in range(N):
    x = _step(dynamics, method, x, interval_us, i, h)
    out.append(x)
  return x, torch.stack(out)


# --------------------------------------------------------------------------------------
# L3  transcriptions
# --------------------------------------------------------------------------------------
@dataclass
class Transcription:
  """What TrajectoryOptimizer (trajectory_optimizers/base.py:28-51) carries."""
  objective: Callable
  constraints: Callable
  bounds: np.ndarray
  guess: np.ndarray
  x_rows: int
  u_rows: int
  ns: int
  nu: int
  x_guess: np.ndarray = None
  u_guess: np.ndarray = None

  def unravel(self, z):
    nx = self.x_rows * self.ns
    return z[:nx].reshape(self.x_rows, self.ns), z[nx:].reshape(self.u_rows, self.nu)


def _bounds(system: System, x_rows: int, u_rows: int, trap_quirk=False) -> np.ndarray:
  """hermite_simpson.py:55-81 / shooting.py:247-275 / trapezoidal.py:55-77."""
  ns, nu = system.ns, system.nu
  xb = np.zeros((x_rows, ns, 2))
  xb[:, :, :] = system.bounds[:-nu]
  xb[0, :, :] = system.x_0[:, None]
  if system.x_T is not None:
    if trap_quirk:                       # trapezoidal.py:71 pins row [-control_shape] (quirk Q2)
      xb[-nu, :, :] = system.x_T[:, None]
    else:
      for i in range(len(system.x_T)):
        if system.x_T[i] is not None:
          xb[-1, i, :] = system.x_T[i]
  xb = xb.reshape(-1, 2)
  ub = np.empty((u_rows * nu, 2))
  for i in range(nu, 0, -1):             # control-major blocks (quirk Q1; harmless for nu=1)
    ub[(nu - i) * u_rows:(nu - i + 1) * u_rows] = system.bounds[-i]
  return np.vstack((xb, ub))


def hermite_simpson(system: System, intervals: int) -> Transcription:
  """collocation/hermite_simpson.py:16-351."""
  N = intervals
  K = 2 * N + 1
  h = system.T / N                                             # :27 interval_duration
  ns, nu = system.ns, system.nu
  u_guess = np.zeros((K, nu))                                  # :37
  if system.x_T is not None:
    x_guess = np.linspace(system.x_0, system.x_T, num=K)       # :41
  else:
    x_guess = np.ones((K, ns)) * 0.1                           # :43
  guess = np.concatenate([x_guess.ravel(), u_guess.ravel()])   # ravel_pytree :47
  bounds = _bounds(system, K, K)

  def split(z):                                                # :84-107
    xs = z[:K * ns].reshape(K, ns)
    us = z[K * ns:].reshape(K, nu)
    return xs[0:-1:2], xs[1::2], xs[2::2], us[0:-1:2], us[1::2], us[2::2]

  def objective(z):                                            # :243-257 with hs_cost :194-214
    xs_, xm, xe, us_, um, ue = split(z)
    t = torch.linspace(0., system.T, 2 * N + 1, dtype=DT)      # :252-256: times of the points (cost functions with t)
    return ((h / 6) * (system.cost(xs_, us_, t[0:-1:2]) + 4 * system.cost(xm, um, t[1::2]) + system.cost(xe, ue, t[2::2]))).sum()

  def constraints(z):                                          # :325-335
    xs_, xm, xe, us_, um, ue = split(z)
    fs, fm, fe = system.dynamics(xs_, us_), system.dynamics(xm, um), system.dynamics(xe, ue)
    defect = (xe - xs_) - (h / 6) * (fs + 4 * fm + fe)         # hs_defect :110-128
    interp = xm - 0.5 * (xs_ + xe) - (h / 8) * (fs - fe)       # hs_interpolation :153-170
    return torch.cat([defect.reshape(-1), interp.reshape(-1)])

  return Transcription(objective, constraints, bounds, guess, K, K, ns, nu, x_guess, u_guess)


def trapezoidal(system: System, intervals: int, method: str = "HEUN") -> Transcription:
  """collocation/trapezoidal.py:16-209."""
  N = intervals
  h = system.T / N
  ns, nu = system.ns, system.nu
  u_guess = np.zeros((N + 1, nu))                              # :34
  x_guess = _rollout_guess(system, u_guess, h, N, method, N + 1)   # :36-50
  guess = np.concatenate([x_guess.ravel(), u_guess.ravel()])
  bounds = _bounds(system, N + 1, N + 1, trap_quirk=True)

  def split(z):
    x = z[:(N + 1) * ns].reshape(N + 1, ns)
    u = z[(N + 1) * ns:].reshape(N + 1, nu)
    return x, u

  def objective(z):                                            # :115-128
    x, u = split(z)
    t = torch.linspace(0., system.T, N + 1, dtype=DT)          # :124
    c = ((h / 2) * (system.cost(x[:-1], u[:-1], t[:-1]) + system.cost(x[1:], u[1:], t[1:]))).sum()
    if system.terminal_cost:                                   # :126-127
      c = c + system.terminal_cost_fn(x[-1], u[-1])
    return c

  def constraints(z):                                          # :183-192, trapezoid_defect :151-163
    x, u = split(z)
    left = (h / 2) * (system.dynamics(x[:-1], u[:-1]) + system.dynamics(x[1:], u[1:]))
    right = x[1:] - x[:-1]
    return (left - right).reshape(-1)                          # sign opposite to HS

  return Transcription(objective, constraints, bounds, guess, N + 1, N + 1, ns, nu, x_guess, u_guess)


def _rollout_guess(system, controls_for_guess, interval_size, intervals, method, rows):
  """shooting.py:56-74 / trapezoidal.py:36-50: linspace where x_T given, else coarse rollout."""
  ns = system.ns
  def roll():
    _, xs = integrate_time_independent(system.dynamics, _t(system.x_0), _t(controls_for_guess),
                                       interval_size, intervals, method)
    return xs.numpy()
  if system.x_T is not None:
    cols = []
    rolled = None
    for i in range(len(system.x_T)):
      if system.x_T[i] is not None:
        cols.append(np.linspace(system.x_0[i], system.x_T[i], num=rows).reshape(-1, 1))
      else:
        rolled = roll() if rolled is None else rolled
        cols.append(rolled[:, i].reshape(-1, 1))
    return np.hstack(cols)
  return roll()


def shooting(system: System, intervals: int, controls_per_interval: int, method: str = "HEUN") -> Transcription:
  """shooting.py:16-278 (MultipleShootingOptimizer)."""
  I, cpi = intervals, controls_per_interval
  S = I * cpi                                                  # :26
  step = system.T / S                                          # :27
  interval_size = system.T / I                                 # :28
  ns, nu = system.ns, system.nu
  mc = 2 if method == "RK4" else 1                             # :31
  R_u = mc * S + 1
  u_guess = np.zeros((R_u, nu))                                # :47
  x_guess = _rollout_guess(system, u_guess[::mc * cpi], interval_size, I, method, I + 1)  # :56-74
  guess = np.concatenate([x_guess.ravel(), u_guess.ravel()])
  bounds = _bounds(system, I + 1, R_u)

  def split(z):
    return z[:(I + 1) * ns].reshape(I + 1, ns), z[(I + 1) * ns:].reshape(R_u, nu)

  def reorganize_controls(us):                                 # :100-130  -> (I, mc*cpi+1, nu)
    a = us[:-1].reshape(I, mc * cpi, nu)
    b = us[::mc * cpi][1:][:, None]
    return torch.cat([a, b], dim=1)

  def constraints(z):                                          # :230-241
    xs, us = split(z)
    ru = reorganize_controls(us).transpose(0, 1)               # (mc*cpi+1, I, nu): batch over I
    px, _ = integrate_time_independent(system.dynamics, xs[:-1], ru, step, cpi, method)
    return (px - xs[1:]).reshape(-1)

  def objective(z):                                            # :169-210, augmented_dynamics :80-92
    xs, us = split(z)
    ru = reorganize_controls(us).transpose(0, 1)
    def aug(x_and_c, u, t):
      x = x_and_c[..., :-1]
      return torch.cat([system.dynamics(x, u), system.cost(x, u, t)[..., None]], dim=-1)
    t = torch.linspace(0., system.T, S + 1, dtype=DT)          # :196
    ts = torch.cat([t[:-1].reshape(I, cpi), t[::cpi][1:].reshape(I, 1)], dim=1).transpose(0, 1)   # reorganize_times :132-142
    start = torch.cat([xs[:-1], torch.zeros(I, 1, dtype=DT)], dim=1)      # :198
    end, _ = integrate(aug, start, ru, step, cpi, ts, method)             # :199-203 (integrate_in_parallel with the step times)
    total = end[:, -1].sum()                                   # :205
    if system.terminal_cost:                                   # :206-208: on the INTEGRATED end state of the last interval
      total = total + system.terminal_cost_fn(end[-1, :-1], us[-1])
    return total

  return Transcription(objective, constraints, bounds, guess, I + 1, R_u, ns, nu, x_guess, u_guess)


def make_transcription(system: System, optimizer: str, intervals: int, controls_per_interval: int = 1,
                       quadrature_rule: str = "TRAPEZOIDAL", integration_method: str = "HEUN") -> Transcription:
  """trajectory_optimizers/__init__.py:12-28 get_optimizer dispatch."""
  if optimizer == "COLLOCATION":
    if quadrature_rule == "TRAPEZOIDAL":
      return trapezoidal(system, intervals, integration_method)
    if quadrature_rule == "HERMITE_SIMPSON":
      return hermite_simpson(system, intervals)
    raise KeyError(quadrature_rule)
  if optimizer == "SHOOTING":
    return shooting(system, intervals, controls_per_interval, integration_method)
  raise KeyError(optimizer)


# --------------------------------------------------------------------------------------
# L4  callbacks + solve  (myriad/nlp_solvers/__init__.py:18-98)
# --------------------------------------------------------------------------------------
class Callbacks:
  """fun / jac / constraints.fun / constraints.jac exactly as nlp_solvers/__init__.py:31-42 builds them
  (jax.grad -> torch.func.grad, jax.jacrev -> torch.func.jacrev)."""

  def __init__(self, tr: Transcription):
    self.tr = tr
    self._grad = torch.func.grad(tr.objective)
    self._jac = torch.func.jacrev(tr.constraints)

  def fun(self, z):
    return float(self.tr.objective(_t(z)))

  def grad(self, z):
    return self._grad(_t(z)).numpy()

  def cons(self, z):
    return self.tr.constraints(_t(z)).numpy()

  def jac(self, z):
    return self._jac(_t(z)).numpy()


def _rk4_fbsm(dyn, x_t1, u, u_next, v, v_next, h, t):
  """utils.py:166-175: RK4 step with the costates averaged at the half step (stage times t, t + h/2, t + h)."""
  um, vm = (u + u_next) / 2, (v + v_next) / 2
  k1 = dyn(x_t1, u, v, t)
  k2 = dyn(x_t1 + h * k1 / 2, um, vm, t + h / 2)
  k3 = dyn(x_t1 + h * k2 / 2, um, vm, t + h / 2)
  k4 = dyn(x_t1 + h * k3, u_next, v_next, t + h)
  return x_t1 + (h / 6) * (k1 + 2 * k2 + 2 * k3 + k4)


def integrate_fbsm(dyn, x_0, u, h, N, v=None, t=None, discrete=False):
  """utils.py:138-197: forward (h > 0) from index 0, backward (h < 0) from index N.  `discrete` (:184-188): the direct
  recurrences dyn(x, u[idx], v[idx], t[idx]) forward and dyn(x, u[idx], v[idx-1], t[idx-1]) backward (idx = N..1)."""
  if v is None:
    v = np.zeros_like(u)
  if t is None:
    t = np.zeros(N + 1)
  out = np.zeros((N + 1,) + np.shape(x_0))
  if discrete:
    if h >= 0:
      out[0] = x_0
      for i in range(N):
        out[i + 1] = dyn(out[i], u[i], v[i], t[i])
    else:
      out[N] = x_0
      for i in range(N, 0, -1):
        out[i - 1] = dyn(out[i], u[i], v[i - 1], t[i - 1])
    return out
  if h >= 0:
    out[0] = x_0
    for i in range(N):
      out[i + 1] = _rk4_fbsm(dyn, out[i], u[i], u[i + 1], v[i], v[i + 1], h, t[i])
  else:
    out[N] = x_0
    for i in range(N, 0, -1):
      out[i - 1] = _rk4_fbsm(dyn, out[i], u[i], u[i - 1], v[i], v[i - 1], h, t[i])
  return out


def fbsm(system, N: int = 1000, delta: float = 0.001, max_sweeps: int = 10000):
  """Forward-Backward Sweep (trajectory_optimizers/forward_backward_sweep.py:20-116) for systems without terminal
  state conditions; stopping rule of trajectory_optimizers/base.py:128-141.  Returns {'x','u','adj','sweeps'}.
  Systems provide np_dynamics(x, u), adj_ODE(adj, x, u, t) and optim_characterization(adj, x, t) on [N+1, .] arrays."""
  ns = system.x_0.shape[0]
  nu = system.bounds.shape[0] - ns
  h = system.T / N
  disc = bool(getattr(system, "discrete", False))
  if disc:                                                     # :33-35 (the `N` argument is ignored, as hp.fbsm_intervals is)
    N, h = int(system.T), 1
  t = np.linspace(0, system.T, N + 1)                          # :48
  x = np.vstack([system.x_0, np.zeros((N, ns))])               # :38
  u = np.zeros((N if disc else N + 1, nu))                     # :39-42
  adj_T = getattr(system, "adj_T", None)
  adj = np.zeros((N + 1, ns)) if adj_T is None else np.vstack([np.zeros((N, ns)), np.asarray(adj_T, dtype=np.float64)])   # :44-47
  f = lambda x_, u_, v_, t_: system.np_dynamics(x_, u_)
  a = lambda adj_, x_, u_, t_: system.adj_ODE(adj_, x_, u_, t_)
  n = 0
  while True:
    old_u, old_x, old_adj = u.copy(), x.copy(), adj.copy()
    x = integrate_fbsm(f, x[0], u, h, N, t=t, discrete=disc)               # :95-96
    adj = integrate_fbsm(a, adj[-1], x, -h, N, u, t=t, discrete=disc)      # :97-98
    u = 0.5 * (system.optim_characterization(adj, x, t[:, None]) + old_u)  # :100-102
    n += 1
    stop = np.hstack([np.abs(v).sum(0) * delta - np.abs(v - o).sum(0) for v, o in ((u, old_u), (x, old_x), (adj, old_adj))])
    if not (stop.min() < 0) or n >= max_sweeps:                            # base.py:141
      break
  return {"x": x, "u": u, "adj": adj, "sweeps": n}


def fbsm_secant(system, N: int = 1000, max_sweeps: int = 10000, max_secant: int = 100):
  """FBSM.sequencesolver (forward_backward_sweep.py:118-158): secant method on the terminal adjoint of the pinned state."""
  idx = [i for i, v in enumerate(system.x_T) if v is not None]
  assert len(idx) == 1
  idx, val = idx[0], float(system.x_T[idx[0]])
  base = np.asarray(system.adj_T, dtype=np.float64).copy()

  def V(a):
    aT = base.copy(); aT[idx] = a                              # reinitiate(a) :72-86
    s2 = _With(system, adj_T=aT)
    r = fbsm(s2, N, max_sweeps=max_sweeps)
    return r["x"][-1, idx] - val, r

  a, b = system.guess_a, system.guess_b
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
  sol["secant_iterations"] = count
  return sol


class _With:
  """view of a system with some attributes replaced (used to vary adj_T without mutating the system)"""
  def __init__(self, base, **kw):
    self.__dict__["_b"] = base
    self.__dict__["_kw"] = kw
  def __getattr__(self, k):
    kw = self.__dict__["_kw"]
    return kw[k] if k in kw else getattr(self.__dict__["_b"], k)


class Lagrangian:
  """lagrangian(x, lmbda) = fun(x) + lmbda @ constraint_fun(x) and the extragradient `step` exactly as
  nlp_solvers/extra_gradient.py:21-33 (and experiments/e2e_sysid.py:113-125) build them: gradients by reverse-mode
  autodiff through the restated transcription (jax.grad -> torch.func.grad)."""

  def __init__(self, tr: Transcription):
    self.tr = tr
    self._L = lambda x, lm: tr.objective(x) + lm @ tr.constraints(x)
    self._gx = torch.func.grad(self._L, argnums=0)
    self._gl = torch.func.grad(self._L, argnums=1)

  def value(self, x, lmbda):
    return float(self._L(_t(x), _t(lmbda)))

  def grad_x(self, x, lmbda):
    return self._gx(_t(x), _t(lmbda)).numpy()

  def grad_lmbda(self, x, lmbda):
    return self._gl(_t(x), _t(lmbda)).numpy()

  def jvp(self, x, v):
    """J(x) v by forward-mode autodiff of the constraints."""
    return torch.func.jvp(self.tr.constraints, (_t(x),), (_t(v),))[1].numpy()

  def step(self, x, lmbda, eta_x, eta_v):                       # extra_gradient.py:25-33
    lb, ub = self.tr.bounds[:, 0], self.tr.bounds[:, 1]
    x_bar = np.clip(x - eta_x * self.grad_x(x, lmbda), lb, ub)
    x_new = np.clip(x - eta_x * self.grad_x(x_bar, lmbda), lb, ub)
    lmbda_new = lmbda + eta_v * self.grad_lmbda(x_new, lmbda)
    return x_new, lmbda_new


def solve(tr: Transcription, nlpsolver: str = "SLSQP", max_iter: int = 1000, guess=None,
          extra_options: Optional[dict] = None, cb: Optional[Callbacks] = None):
  """nlp_solvers/__init__.py:18-98 restricted to the SciPy branches (:50-55).
  Returns the reference's result dict (+ raw scipy result under 'scipy')."""
  from scipy.optimize import minimize
  cb = cb or Callbacks(tr)
  opts = {"maxiter": max_iter}
  if extra_options:
    opts.update(extra_options)
  method = {"SLSQP": "SLSQP", "TRUST": "trust-constr"}[nlpsolver]
  sol = minimize(fun=cb.fun, x0=tr.guess if guess is None else guess, method=method, jac=cb.grad,
                 constraints=({"type": "eq", "fun": cb.cons, "jac": cb.jac}),
                 bounds=tr.bounds, options=opts)
  x, u = tr.unravel(sol["x"])
  res = {"x": x, "u": u, "xs_and_us": sol["x"], "cost": sol["fun"], "scipy": sol}
  if nlpsolver == "TRUST":
    res["lambda"] = sol["v"]                                   # :83-84
  return res


# --------------------------------------------------------------------------------------
# post-solve validation  (myriad/utils.py:258-324)
# --------------------------------------------------------------------------------------
def get_state_trajectory_and_cost(system: System, num_steps: int, method: str, start_state, us):
  """utils.py:258-298: integrate [x; cost] under `us` with hp.integration_method over num_steps."""
  step = system.T / num_steps
  us = _t(us)
  def aug(x_and_c, u, t):
    x = x_and_c[:-1]
    return torch.cat([system.dynamics(x, u), system.cost(x, u, t).reshape(1)])
  times = torch.linspace(0., system.T, num_steps + 1, dtype=DT)
  start = torch.cat([_t(start_state), torch.zeros(1, dtype=DT)])
  _, sc = integrate(aug, start, us, step, num_steps, times, method)
  cost = float(sc[-1, -1])
  if system.terminal_cost:                                     # utils.py:295-296
    cost += float(system.terminal_cost_fn(sc[-1, :-1], us[-1]))
  return sc[:, :-1].numpy(), cost


def get_defect(system: System, xs):
  """utils.py:313-324."""
  if system.x_T is None:
    return None
  return np.array([xs[-1][i] - system.x_T[i] for i in range(len(system.x_T)) if system.x_T[i] is not None])


# --------------------------------------------------------------------------------------
# dense Jacobian <-> stage blocks (the HIP eval kernel's output layout, include/myriad_hip.h)
# --------------------------------------------------------------------------------------
def hs_blocks_from_dense(J: np.ndarray, N: int, ns: int, nu: int) -> np.ndarray:
  """Extract the per-interval stage blocks [N, 5*ns*ns + 5*ns*nu] from a dense HS Jacobian,
  in the order Dxs,Dxm,Dxe,Dus,Dum,Due,Ixs,Ixe,Ius,Iue (each row-major)."""
  K = 2 * N + 1
  out = []
  for k in range(N):
    rd = slice(k * ns, (k + 1) * ns)
    ri = slice(N * ns + k * ns, N * ns + (k + 1) * ns)
    cx = lambda p: slice(p * ns, (p + 1) * ns)
    cu = lambda p: slice(K * ns + p * nu, K * ns + (p + 1) * nu)
    s, m, e = 2 * k, 2 * k + 1, 2 * k + 2
    blk = [J[rd, cx(s)], J[rd, cx(m)], J[rd, cx(e)], J[rd, cu(s)], J[rd, cu(m)], J[rd, cu(e)],
           J[ri, cx(s)], J[ri, cx(e)], J[ri, cu(s)], J[ri, cu(e)]]
    out.append(np.concatenate([b.ravel() for b in blk]))
  return np.stack(out)


def hs_dense_from_blocks(blk: np.ndarray, N: int, ns: int, nu: int) -> np.ndarray:
  """Inverse of hs_blocks_from_dense (adds the identity d interp / d x_m that the kernel does not store)."""
  K = 2 * N + 1
  n, m = K * (ns + nu), 2 * N * ns
  J = np.zeros((m, n))
  szs = [ns * ns] * 3 + [ns * nu] * 3 + [ns * ns] * 2 + [ns * nu] * 2
  for k in range(N):
    parts = np.split(blk[k], np.cumsum(szs)[:-1])
    rd = slice(k * ns, (k + 1) * ns)
    ri = slice(N * ns + k * ns, N * ns + (k + 1) * ns)
    cx = lambda p: slice(p * ns, (p + 1) * ns)
    cu = lambda p: slice(K * ns + p * nu, K * ns + (p + 1) * nu)
    s, mm, e = 2 * k, 2 * k + 1, 2 * k + 2
    J[rd, cx(s)] += parts[0].reshape(ns, ns); J[rd, cx(mm)] += parts[1].reshape(ns, ns)
    J[rd, cx(e)] += parts[2].reshape(ns, ns)
    J[rd, cu(s)] += parts[3].reshape(ns, nu); J[rd, cu(mm)] += parts[4].reshape(ns, nu)
    J[rd, cu(e)] += parts[5].reshape(ns, nu)
    J[ri, cx(s)] += parts[6].reshape(ns, ns); J[ri, cx(e)] += parts[7].reshape(ns, ns)
    J[ri, cx(mm)] += np.eye(ns)
    J[ri, cu(s)] += parts[8].reshape(ns, nu); J[ri, cu(e)] += parts[9].reshape(ns, nu)
  return J


# --------------------------------------------------------------------------------------
# workload generators shared by tests and bench
# --------------------------------------------------------------------------------------
def random_x0(system: System, B: int, seed: int = 2019, spread: float = 0.1) -> np.ndarray:
  """x0_b = clip(x_0 + spread * N(0, I), state bounds): the reference's start-state perturbation
  rule, utils.py:412-419 with hp.start_spread = 0.1 (config.py:80); numpy default_rng(seed)."""
  rng = np.random.default_rng(seed)
  x0 = system.x_0[None, :] + spread * rng.standard_normal((B, system.ns))
  return np.clip(x0, system.bounds[:system.ns, 0], system.bounds[:system.ns, 1])
