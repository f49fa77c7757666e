"""Hyper-parameters, run configuration and the enums of the reference's `myriad/config.py` (:12-129): same member names and
values, same field names, defaults and derived fields -- declared here as tables.  New: NLPSolverType.SQP, the batched
interior-point SQP on the GPU that takes the place of the IPOPT call."""
from __future__ import annotations

from dataclasses import field, make_dataclass
from enum import Enum
from typing import Tuple

from myriad_amd.systems import SystemType


def _named(name, members):
  """Enum whose values are given as 'MEMBER' (value = the name) or 'MEMBER=value' words."""
  return Enum(name, [tuple(w.split("=")) if "=" in w else (w, w) for w in members.split()], module=__name__)


OptimizerType = _named("OptimizerType", "COLLOCATION SHOOTING FBSM")                                   # :12-16
SamplingApproach = _named("SamplingApproach", "UNIFORM TRUE_OPTIMAL RANDOM_WALK CURRENT_OPTIMAL")      # :19-27
# IPOPT is not in this image: that slot is routed to SQP (nlp_solvers/__init__.py); SQP itself is new
NLPSolverType = _named("NLPSolverType", "SLSQP TRUST IPOPT EXTRAGRADIENT SQP")                         # :30-43
IntegrationMethod = _named("IntegrationMethod", "EULER=CONSTANT HEUN=LINEAR MIDPOINT RK4")             # :46-50
QuadratureRule = _named("QuadratureRule", "TRAPEZOIDAL HERMITE_SIMPSON")                               # :53-57

_HPARAMS = [  # (field, type, default)                                                                  :60-93
  ("seed", int, 2019), ("system", SystemType, SystemType.CANCERTREATMENT), ("optimizer", OptimizerType, OptimizerType.SHOOTING),
  ("nlpsolver", NLPSolverType, NLPSolverType.IPOPT), ("integration_method", IntegrationMethod, IntegrationMethod.HEUN),
  ("quadrature_rule", QuadratureRule, QuadratureRule.TRAPEZOIDAL),
  # solver sizes
  ("max_iter", int, 1000), ("intervals", int, 1), ("controls_per_interval", int, 100), ("fbsm_intervals", int, 1000),
  # data generation / model learning (host-side experiments of the reference; carried for interface parity)
  ("sampling_approach", SamplingApproach, SamplingApproach.RANDOM_WALK), ("train_size", int, 100), ("val_size", int, 3),
  ("test_size", int, 3), ("sample_spread", float, 0.05), ("start_spread", float, 0.1), ("noise_level", float, 0.0),
  ("to_smooth", bool, False), ("learning_rate", float, 0.001), ("minibatch_size", int, 16), ("num_epochs", int, 10_001),
  ("num_experiments", int, 1), ("loss_recording_frequency", int, 10), ("plot_progress_frequency", int, 10),
  ("early_stop_threshold", int, 30), ("early_stop_check_frequency", int, 20), ("hidden_layers", Tuple[int, int], (50, 50)),
  ("num_unrolled", int, 5), ("eta_x", float, 1e-1), ("eta_lmbda", float, 1e-3), ("adam_lr", float, 1e-4),
]


def _derive(hp):
  """:95-112 -- the derived fields (collocation has one control per interval, extragradient gets ten times the
  iterations, step counts and sizes from a default instance of the system)."""
  if hp.optimizer == OptimizerType.COLLOCATION:
    hp.controls_per_interval = 1
  if hp.nlpsolver == NLPSolverType.EXTRAGRADIENT:
    hp.max_iter *= 10
  probe = hp.system()
  hp.num_steps = hp.intervals * hp.controls_per_interval
  hp.stepsize = probe.T / hp.num_steps
  hp.key = hp.seed                  # the reference keeps jax.random.PRNGKey(seed); numpy Generators are seeded from it here
  hp.state_size = probe.x_0.shape[0]
  hp.control_size = probe.bounds.shape[0] - hp.state_size
  hp.minibatch_size = min(hp.minibatch_size, hp.train_size, hp.val_size, hp.test_size)


HParams = make_dataclass("HParams", [(n, t, field(default=d)) for n, t, d in _HPARAMS], namespace={"__post_init__": _derive},
                         eq=True, frozen=False)
HParams.__module__ = __name__
HParams.__doc__ = "config.py:60-112 -- same fields, same defaults, same derived fields."

Config = make_dataclass("Config", [(n, t, field(default=d)) for n, t, d in [                            # :115-129
  ("verbose", bool, True), ("jit", bool, True), ("plot", bool, True), ("pretty_plotting", bool, True),
  ("load_params_if_saved", bool, True), ("figsize", Tuple[float, float], (8, 6)), ("file_extension", str, "png")]],
                        eq=True, frozen=False)
Config.__module__ = __name__
Config.__doc__ = "config.py:115-129."
