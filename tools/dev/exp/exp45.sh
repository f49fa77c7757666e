#!/bin/bash
# exp45: config 3 (VANDERPOL single shooting, B = 8192): how ragged is the launch, and do the residuals after K1 iterations predict what is left?
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, numpy as np, heapq
from scipy.stats import spearmanr
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
rng = np.random.default_rng(2019)
hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()); B = 8192
x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4)
full = opt.solve_batch(x0s=x0)["iters"].astype(float)
def makespan(L, slots):
  h = [0.0] * slots; heapq.heapify(h)
  for x in L:
    t = heapq.heappop(h); heapq.heappush(h, t + x)
  return max(h)
print("iters median %g p90 %g p99 %g max %g" % (np.median(full), np.percentile(full, 90), np.percentile(full, 99), full.max()))
for slots in (1024, 2048, 4096):
  line = ["slots %d: one phase %.0f ideal %.1f" % (slots, makespan(full, slots), full.sum() / slots)]
  for K1 in (16, 24, 32):
    r = opt.solve_batch(x0s=x0, max_iter=K1)
    k = r["kkt"]; rem = np.maximum(full - K1, 0); ph1 = makespan(np.minimum(full, K1), slots)
    key = np.sqrt(np.maximum(k[:, 1] * k[:, 2], 0)); key = np.where(key > 0, key, np.maximum(k[:, 1], k[:, 2]))
    line.append("K1=%d rho %.2f: two-phase %.0f (perfect %.0f)" % (K1, spearmanr(key, rem).correlation, ph1 + makespan(rem[np.argsort(-key)], slots), ph1 + makespan(rem[np.argsort(-rem)], slots)))
  print("  ".join(line), flush=True)
PY
