# dev tool: NodeSystem under shooting -- time and convergence of small batches
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
for (intervals, cpi, B, mi) in [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]]:
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=intervals, controls_per_interval=cpi,
               hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
  rng = np.random.default_rng(2019)
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  opt.engine.kernel_time_reset()
  t0 = time.time()
  res = opt.solve_batch(x0s=x0, params=opt.system.device_params(), max_iter=mi)
  dt = time.time() - t0
  ms, n = opt.engine.kernel_time(_lib.K_SOLVE)
  print(f"intervals={intervals} cpi={cpi} B={B} max_iter={mi}: {dt:.2f} s wall, solve kernels {ms:.1f} ms x {n}, status {np.bincount(res['status']).tolist()}, iters median {np.median(res['iters'])} max {res['iters'].max()}, attempts max {res['attempts'].max()}", flush=True)
