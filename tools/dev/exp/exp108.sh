#!/bin/bash
# exp108: the backtracking factor of the network problem's line search: 0.7 and 0.35 against 0.5 (a rejected trial sets a <- FACTOR a; -DMYR_LS_FACTOR_MLP, fused kernel only)
# full step: kernel time, iteration distribution, convergence, optimum against the regular library (draws of config5_1gpu.jsonl)
cd /root/repo; O=gpurun_out/exp108; mkdir -p $O
for lib in myriad_amd/libmyriad_hip.so xv/libfac0.7.so xv/libfac0.35.so; do
  echo "== $lib"
  MYRIAD_HIP_LIB=$PWD/$lib timeout 600 python - <<'PY'
import os, sys, json, hashlib, numpy as np
sys.path.insert(0, os.getcwd())
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
rng = np.random.default_rng(2019)
for B in (128, 256, 300, 512, 1024, 2048):
  x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
  opt.solve_batch(x0s=x0, params=opt.system.device_params())
  ts = []
  for _ in range(3):
    opt.engine.kernel_time_reset(); res = opt.solve_batch(x0s=x0, params=opt.system.device_params()); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms)
  it = res["iters"]
  print(f"B={B}: {min(ts):.2f} ms = {B / min(ts):.1f} k solves/s, converged {(res['status'] == 0).mean():.3f}, iterations sum {it.sum()} median {np.median(it):.0f} p99 {np.percentile(it, 99):.0f} max {it.max()}, cost mean {res['cost'].mean():.9f}", flush=True)
PY
done 2>&1 | grep -v amdgpu.ids | tee $O/times.txt
