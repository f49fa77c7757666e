#!/bin/bash
# exp111: workgroups per SIMD of the shooting wavefront kernel (-DMYR_SHOOT_MIN_WAVES = 1 / 2 (regular) / 4): configs 3 and 4, kernel time of a batch
cd /root/repo; O=gpurun_out/exp111; mkdir -p $O
for lib in myriad_amd/libmyriad_hip.so xv/libmw1.so xv/libmw4.so; do
  MYRIAD_HIP_LIB=$PWD/$lib timeout 300 python - <<'PY' 2>&1 | grep "config"
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
CFG = Config(verbose=False, plot=False)
rng = np.random.default_rng(2019)
lib = os.environ["MYRIAD_HIP_LIB"].split("/")[-1]
def run(tag, opt, **kw):
  opt.solve_batch(**kw); ts = []
  for _ in range(5):
    opt.engine.kernel_time_reset(); r = opt.solve_batch(**kw); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms * max(1, n))
  B = len(r["status"])
  print(f"{lib} config {tag}: kernels {min(ts):.2f} ms per batch of {B} = {B / min(ts):.0f} k solves/s, converged {(r['status'] == 0).mean():.4f}, iterations median {np.median(r['iters']):.0f} max {r['iters'].max()}", flush=True)
hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, CFG, hp.system())
run("3 VANDERPOL 1x50 B=8192", opt, x0s=np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((8192, 2)), -4, 4))
hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.SHOOTING, max_iter=500, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, CFG, hp.system())
params = np.stack([rng.uniform(0.1, 0.5, 2048), rng.uniform(1, 5, 2048), rng.uniform(0.2, 0.8, 2048)], axis=1)
run("4 CANCERTREATMENT 1x100 B=2048", opt, x0s=rng.uniform(0.5, 0.99, (2048, 1)), params=params)
PY
done | tee $O/times.txt
