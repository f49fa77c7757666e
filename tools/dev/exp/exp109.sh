#!/bin/bash
# exp109: the trapezoidal form of the headline system (README:83's literal) needs 21 KB of LDS -- seven workgroups per CU by LDS, four by registers (452 VGPRs): the kernel
# compiled for TWO workgroups per SIMD (-DMYR_FUSED_OCC=2: 256 registers, 316 spilled) against the regular build
cd /root/repo; O=gpurun_out/exp109; mkdir -p $O
for lib in myriad_amd/libmyriad_hip.so xv/libocc2.so; do
  MYRIAD_HIP_LIB=$PWD/$lib MYRIAD_DEBUG_PTRS=1 timeout 300 python - <<'PY' 2>&1 | grep -E "B=|fused W" | cut -c1-150
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=100, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
rng = np.random.default_rng(2019)
for B in (4096, 8192, 1024):
  x0 = np.clip(opt.system.x_0[None] + 0.1 * rng.standard_normal((B, 4)), opt.system.bounds[:4, 0], opt.system.bounds[:4, 1])
  opt.solve_batch(x0s=x0); ts = []
  for _ in range(3):
    opt.engine.kernel_time_reset(); r = opt.solve_batch(x0s=x0); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms)
  print(f"{os.environ['MYRIAD_HIP_LIB'].split('/')[-1]} B={B}: {min(ts):.2f} ms in {n} launches = {B / min(ts):.0f} k solves/s, converged {(r['status'] == 0).mean():.3f}, iterations median {np.median(r['iters']):.0f}", flush=True)
PY
done | tee $O/times.txt
