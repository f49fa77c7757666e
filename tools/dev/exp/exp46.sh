#!/bin/bash
# exp46: which of the loop's scalars at the parking point say how long a solve still is?  (config 5 and the headline workload; MYRIAD_PARK_DUMP)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp46
python - <<'PY'
import os, numpy as np, heapq
from scipy.stats import spearmanr
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
NAMES = ["mu", "pen", "pen_over", "pen_cuts", "stall", "small_steps", "delta_last", "lm", "nhist", "hpos", "hist_mu", "hist_pen", "p_on", "p_ap", "p_ad", "p_mu", "p_ksig"]
def makespan(L, slots):
  h = [0.0] * slots; heapq.heapify(h)
  for x in L:
    t = heapq.heappop(h); heapq.heappush(h, t + x)
  return max(h)
def study(tag, solve, B, slots, K1s):
  os.environ["MYRIAD_PARK_ITER"] = "0"; os.environ.pop("MYRIAD_PARK_DUMP", None)
  full = solve(None)["iters"].astype(float)
  print(tag, "iters median %g p99 %g max %g; one phase %.0f ideal %.1f" % (np.median(full), np.percentile(full, 99), full.max(), makespan(full, slots), full.sum() / slots))
  for K1 in K1s:
    path = "gpurun_out/exp46/%s_k%d.bin" % (tag, K1)
    os.environ["MYRIAD_PARK_ITER"] = str(K1); os.environ["MYRIAD_PARK_DUMP"] = path
    r = solve(None)
    os.environ.pop("MYRIAD_PARK_DUMP", None)
    sc = np.fromfile(path).reshape(B, -1)
    kk = solve(K1)["kkt"]        # residuals at the parking point (whole solve stopped there)
    rem = np.maximum(full - K1, 0); ph1 = makespan(np.minimum(full, K1), slots)
    parked = full > K1
    feats = {n: sc[:, i] for i, n in enumerate(NAMES)}
    feats.update(feas=kk[:, 0], stat=kk[:, 1], compl=kk[:, 2], geo=np.sqrt(np.maximum(kk[:, 1] * kk[:, 2], 0)))
    feats["geo/mu"] = feats["geo"] / np.maximum(feats["mu"], 1e-300); feats["mu*geo"] = feats["mu"] * feats["geo"]
    feats["delta>0"] = (feats["delta_last"] > 0).astype(float); feats["max(mu,geo)"] = np.maximum(feats["mu"], feats["geo"])
    out = []
    for n, v in feats.items():
      if np.std(v[parked]) == 0: continue
      rho = spearmanr(v[parked], rem[parked]).correlation
      out.append((ph1 + makespan(rem[np.argsort(-v, kind="stable")], slots), n, rho, ph1 + makespan(rem[np.argsort(v, kind="stable")], slots)))
    out.sort()
    print("  K1=%d: ticket order %.0f, perfect %.0f;" % (K1, ph1 + makespan(rem, slots), ph1 + makespan(rem[np.argsort(-rem)], slots)), "  ".join("%s %.0f (rho %+.2f, rev %.0f)" % (n, m, rho, rv) for m, n, rho, rv in out[:7]), flush=True)
# config 5
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
B = 1024
x0 = np.clip(0.1 * np.random.default_rng(2019).standard_normal((B, 4)), -2, 2)
def nsolve(mi):      # (a fresh handle per call: MYRIAD_PARK_ITER is read when the handle is made)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
  r = opt.solve_batch(x0s=x0, params=opt.system.device_params(), **({} if mi is None else dict(max_iter=mi)))
  opt.engine.close()
  return r
study("node", nsolve, B, 256, (8, 12, 16))
# headline
import bench
from myriad_amd import _lib
N, B = 100, 4096
x0h, z0, lb, ub, T = bench.build_workload(B, N, 2019)
def hsolve(mi):
  eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
  o = eng.default_opts(); o.restoration = 0
  if mi is not None: o.max_iter = mi
  r = eng.solve(z0, lb, ub, opts=o); eng.close()
  return r
study("cartpole", hsolve, B, 1024, (10, 12))
PY
