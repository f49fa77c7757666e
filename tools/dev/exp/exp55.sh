#!/bin/bash
# exp55 (round 5): helper workgroups that attach dynamically -- small batches, batches between half and the whole device, the tail of B = 1024
cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY'
import os, numpy as np, hashlib, json
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
def run(B, nh, seed=2019):
  if nh is None: os.environ.pop("MYRIAD_NODE_HELPERS", None)
  else: os.environ["MYRIAD_NODE_HELPERS"] = str(nh)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
  x0 = np.clip(0.1 * np.random.default_rng(seed).standard_normal((B, 4)), -2, 2)
  opt.solve_batch(x0s=x0, params=opt.system.device_params()); ts = []; bits = set()
  for _ in range(3):
    opt.engine.kernel_time_reset(); r = opt.solve_batch(x0s=x0, params=opt.system.device_params()); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms * max(1, n))
    bits.add(hashlib.sha1(b"".join(np.ascontiguousarray(r[k]).tobytes() for k in ("xs_and_us", "lambda", "cost", "status", "iters"))).hexdigest()[:12])
  opt.engine.close()
  return dict(B=B, seed=seed, helpers=nh, kernel_ms=round(float(np.median(ts)), 3), solves_per_s=round(B / float(np.median(ts)) * 1e3), converged=float((r["status"] == 0).mean()), it_max=int(r["iters"].max()), bits=sorted(bits))
for B in (8, 64, 128, 200, 256, 300, 512, 1024):
  for nh in (0, None):
    print(json.dumps(run(B, nh)), flush=True)
for seed in (7, 11):
  for nh in (0, None):
    print(json.dumps(run(1024, nh, seed)), flush=True)
PY
