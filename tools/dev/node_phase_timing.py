# dev tool: config 5 (CARTPOLE + NODE) through a -DMYR_PHASE_TIMING build of the NODE translation unit (prints cycles per phase for
# trajectories 0..3): MYR_VARIANT_SYS=SysNODE_CARTPOLE tools/dev/build_variant.sh myriad_amd/csrc libnodetiming.so -DMYR_PHASE_TIMING
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("MYRIAD_VARIANT_LIB", "variants/libnodetiming.so"))
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x0 = np.clip(0.1 * np.random.default_rng(2019).standard_normal((B, 4)), -2, 2)
res = opt.solve_batch(x0s=x0, params=opt.system.device_params())
print("converged", float((res["status"] == 0).mean()), "kernel ms", opt.engine.kernel_time(_lib.K_SOLVE))
