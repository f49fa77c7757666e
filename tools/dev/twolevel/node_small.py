import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
for N in (2, 3, 5, 7):
  out = {}
  for tag, env in (("w4", {"MYRIAD_FUSED_WAVES": "4"}), ("w2", {"MYRIAD_FUSED_WAVES": "2"}), ("r2", {"MYRIAD_SOLVE_MODE": "wave1"})):
    for k in ("MYRIAD_FUSED_WAVES", "MYRIAD_SOLVE_MODE"): os.environ.pop(k, None)
    os.environ.update(env)
    hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=N, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
    x0 = np.clip(0.1 * np.random.default_rng(N).standard_normal((4, 4)), -2, 2)
    r = opt.solve_batch(x0s=x0, params=opt.system.device_params(), max_iter=200)
    out[tag] = r; opt.engine.close()
  for tag in ("w4", "w2"):
    a, b = out[tag], out["r2"]
    both = (a["status"] == 0) & (b["status"] == 0)
    print(f"N={N} {tag} vs round-2 kernel: status {a['status']} / {b['status']} iters {a['iters']} / {b['iters']} max|dz| on converged pairs {np.abs(a['xs_and_us'][both] - b['xs_and_us'][both]).max(initial=0.0):.2e} dcost {np.abs(a['cost'][both] - b['cost'][both]).max(initial=0.0):.2e}")
