#!/bin/bash
# exp50 (round 5): config 5 baseline of the round -- per-phase cycles incl. the network passes (timing build), iteration distribution at B = 128 / 1024
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp50
MYRIAD_VARIANT_LIB=variants/libnodetiming.so python tools/dev/node_phase_timing.py 128 > gpurun_out/exp50/phase_b128.txt 2>&1
MYRIAD_VARIANT_LIB=variants/libnodetiming.so python tools/dev/node_phase_timing.py 1024 > gpurun_out/exp50/phase_b1024.txt 2>&1
python - > gpurun_out/exp50/iters.txt 2>&1 <<'PY'
import numpy as np, json
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
for seed in (2019, 7, 11):
  for B in (128, 1024):
    x0 = np.clip(0.1 * np.random.default_rng(seed).standard_normal((B, 4)), -2, 2)
    opt.solve_batch(x0s=x0, params=opt.system.device_params()); ts = []
    for _ in range(3):
      opt.engine.kernel_time_reset(); r = opt.solve_batch(x0s=x0, params=opt.system.device_params()); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms * max(1, n))
    it = np.sort(r["iters"])
    print(json.dumps(dict(seed=seed, B=B, kernel_ms=float(np.median(ts)), converged=float((r["status"] == 0).mean()), it_sum=int(it.sum()), it_median=float(np.median(it)),
                          it_p90=float(np.percentile(it, 90)), it_p99=float(np.percentile(it, 99)), top8=it[-8:].tolist())))
PY
tail -12 gpurun_out/exp50/phase_b128.txt; tail -12 gpurun_out/exp50/phase_b1024.txt; cat gpurun_out/exp50/iters.txt
