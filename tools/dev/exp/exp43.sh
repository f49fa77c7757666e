#!/bin/bash
# exp43: config 5 -- do the residuals after K1 iterations predict what is left?  (256 slots: one trajectory per CU)
cd $GRAFT_REPO_ROOT
MYRIAD_PARK_ITER=0 python - <<'PY'
import os, sys, numpy as np, heapq
from scipy.stats import spearmanr
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
rng = np.random.default_rng(2019)
B = 1024
x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
full = opt.solve_batch(x0s=x0, params=opt.system.device_params())["iters"].astype(float)
def makespan(L, slots=256):
  h = [0.0] * slots; heapq.heapify(h)
  for x in L:
    t = heapq.heappop(h); heapq.heappush(h, t + x)
  return max(h)
print("iters: median %g p90 %g p99 %g max %g; one phase makespan %.0f ideal %.1f" % (np.median(full), np.percentile(full, 90), np.percentile(full, 99), full.max(), makespan(full), full.sum() / 256))
for K1 in (8, 12, 16):
  r = opt.solve_batch(x0s=x0, params=opt.system.device_params(), max_iter=K1)
  k = r["kkt"]; rem = np.maximum(full - K1, 0)
  ph1 = makespan(np.minimum(full, K1))
  keys = {"feas": k[:, 0], "stat": k[:, 1], "compl": k[:, 2], "geo": np.sqrt(np.maximum(k[:, 1] * k[:, 2], 0))}
  line = ["K1=%d: phase 1 %.0f + ticket order %.0f, perfect %.0f" % (K1, ph1, ph1 + makespan(rem), ph1 + makespan(rem[np.argsort(-rem)]))]
  for n, v in keys.items():
    line.append("%s rho %+.2f -> %.0f (rev %.0f)" % (n, spearmanr(v, rem).correlation, ph1 + makespan(rem[np.argsort(-v)]), ph1 + makespan(rem[np.argsort(v)])))
  print("  ".join(line), flush=True)
PY
