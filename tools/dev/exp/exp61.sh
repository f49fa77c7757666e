#!/bin/bash
# exp61 (round 5): the two-wavefront THROUGHPUT form of the network kernel (two trajectories per CU) against the four-wavefront form, by batch size; same optima?
cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY'
import os, numpy as np, hashlib, json
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
def run(B, env, seed=2019):
  for k in ("MYRIAD_FUSED_WAVES", "MYRIAD_NODE_HELPERS"): os.environ.pop(k, None)
  os.environ.update(env)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
  x0 = np.clip(0.1 * np.random.default_rng(seed).standard_normal((B, 4)), -2, 2)
  opt.solve_batch(x0s=x0, params=opt.system.device_params()); ts = []
  for _ in range(3):
    opt.engine.kernel_time_reset(); r = opt.solve_batch(x0s=x0, params=opt.system.device_params()); ms, n = opt.engine.kernel_time(_lib.K_SOLVE); ts.append(ms * max(1, n))
  plan = opt.engine.solve_plan(); opt.engine.close()
  return r, dict(B=B, seed=seed, env=env, kernel_ms=round(float(np.median(ts)), 3), solves_per_s=round(B / float(np.median(ts)) * 1e3), converged=float((r["status"] == 0).mean()), it_sum=int(r["iters"].sum()), it_max=int(r["iters"].max()), plan=plan)
for B in (300, 512, 1024, 2048):
  ref, l4 = run(B, {"MYRIAD_FUSED_WAVES": "4"}); print(json.dumps(l4), flush=True)
  r2, l2 = run(B, {"MYRIAD_FUSED_WAVES": "2"}); print(json.dumps(l2), flush=True)
  r2n, l2n = run(B, {"MYRIAD_FUSED_WAVES": "2", "MYRIAD_NODE_HELPERS": "0"}); print(json.dumps(l2n), flush=True)
  same = (ref["iters"] == r2["iters"]).mean(); dz = np.abs(ref["xs_and_us"] - r2["xs_and_us"]).max(); dc = np.abs(ref["cost"] - r2["cost"]).max()
  print(f"   W=2 against W=4: same iteration count on {same:.3f} of the batch, max |dz| {dz:.2e}, max |dcost| {dc:.2e}; W=2 helpers on/off identical bits: {np.array_equal(r2['xs_and_us'], r2n['xs_and_us'])}", flush=True)
for seed in (7, 11):
  for w in ("4", "2"):
    print(json.dumps(run(1024, {"MYRIAD_FUSED_WAVES": w}, seed)[1]), flush=True)
PY
