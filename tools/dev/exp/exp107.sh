#!/bin/bash
# exp107: the line search of the network problem backtracks to the minimiser of the quadratic through phi(0), phi'(0), phi(a) (clamped to [0.1 or 0.25, 0.5] a) instead of a / 2:
# full step: kernel time, iteration distribution, convergence, optimum against the regular library (draws of config5_1gpu.jsonl)
# (The experiment's code is NOT in the tree.  The patch, hs_solver_fused.h: solve(): a rejected, evaluable trial sets
#  a = clamp(-Dphi a^2 / (2 (phi(a) - phi0 - Dphi a)), LO a, a / 2) instead of a / 2.  Result: profiles/r06/README.md.)
cd /root/repo; O=gpurun_out/exp107; mkdir -p $O
for lib in myriad_amd/libmyriad_hip.so xv/libinterp0.1.so xv/libinterp0.25.so; do
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
