"""HParams / Config / enums with the reference's field names and defaults (/root/reference/myriad/config.py:12-129).
New: NLPSolverType.SQP, the MI355X batched SQP that replaces the IPOPT call."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Tuple

from myriad_amd.systems import SystemType


class OptimizerType(Enum):
  COLLOCATION = "COLLOCATION"
  SHOOTING = "SHOOTING"
  FBSM = "FBSM"


class SamplingApproach(Enum):
  UNIFORM = 'UNIFORM'
  TRUE_OPTIMAL = 'TRUE_OPTIMAL'
  RANDOM_WALK = 'RANDOM_WALK'
  CURRENT_OPTIMAL = 'CURRENT_OPTIMAL'


class NLPSolverType(Enum):
  SLSQP = "SLSQP"
  TRUST = "TRUST"
  IPOPT = "IPOPT"                  # no IPOPT here: routed to SQP (nlp_solvers/__init__.py), documented in INTEGRATION.md
  EXTRAGRADIENT = "EXTRAGRADIENT"
  SQP = "SQP"                      # NEW: batched interior-point SQP on the GPU (csrc/hs_solver.h)


class IntegrationMethod(Enum):
  EULER = "CONSTANT"
  HEUN = "LINEAR"
  MIDPOINT = "MIDPOINT"
  RK4 = "RK4"


class QuadratureRule(Enum):
  TRAPEZOIDAL = "TRAPEZOIDAL"
  HERMITE_SIMPSON = "HERMITE_SIMPSON"


@dataclass(eq=True, frozen=False)
class HParams:
  """config.py:60-112 -- same fields, same defaults, same derived fields."""
  seed: int = 2019
  system: SystemType = SystemType.CANCERTREATMENT
  optimizer: OptimizerType = OptimizerType.SHOOTING
  nlpsolver: NLPSolverType = NLPSolverType.IPOPT
  integration_method: IntegrationMethod = IntegrationMethod.HEUN
  quadrature_rule: QuadratureRule = QuadratureRule.TRAPEZOIDAL

  max_iter: int = 1000
  intervals: int = 1
  controls_per_interval: int = 100
  fbsm_intervals: int = 1000

  sampling_approach: SamplingApproach = SamplingApproach.RANDOM_WALK
  train_size: int = 100
  val_size: int = 3
  test_size: int = 3
  sample_spread: float = 0.05
  start_spread: float = 0.1
  noise_level: float = 0.01 * 0.
  to_smooth: bool = False
  learning_rate: float = 0.001
  minibatch_size: int = 16
  num_epochs: int = 10_001
  num_experiments: int = 1
  loss_recording_frequency: int = 10
  plot_progress_frequency: int = 10
  early_stop_threshold: int = 30
  early_stop_check_frequency: int = 20
  hidden_layers: Tuple[int, int] = (50, 50)
  num_unrolled: int = 5
  eta_x: float = 1e-1
  eta_lmbda: float = 1e-3
  adam_lr: float = 1e-4

  def __post_init__(self):
    if self.optimizer == OptimizerType.COLLOCATION:
      self.controls_per_interval = 1
    if self.nlpsolver == NLPSolverType.EXTRAGRADIENT:
      self.max_iter *= 10
    system = self.system()
    self.num_steps = self.intervals * self.controls_per_interval
    self.stepsize = system.T / self.num_steps
    self.key = self.seed            # reference: jax.random.PRNGKey(seed); numpy Generators are seeded from it here
    self.state_size = system.x_0.shape[0]
    self.control_size = system.bounds.shape[0] - self.state_size
    self.minibatch_size = min([self.minibatch_size, self.train_size, self.val_size, self.test_size])


@dataclass(eq=True, frozen=False)
class Config:
  """config.py:115-129."""
  verbose: bool = True
  jit: bool = True
  plot: bool = True
  pretty_plotting: bool = True
  load_params_if_saved: bool = True
  figsize: Tuple[float, float] = (8, 6)
  file_extension: str = 'png'
