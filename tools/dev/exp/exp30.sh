#!/bin/bash
# exp30: does the solver's state after K1 iterations predict how many are left?  (two-phase launch: K1 iterations for everybody, then longest-remaining first)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp30
python - > gpurun_out/exp30/out.txt 2>&1 <<'PY'
import numpy as np, torch, bench, heapq
from scipy.stats import spearmanr
N, B = 100, 4096
x0, z0h, lbh, ubh, T = bench.build_workload(B, N, seed=2019)
eng = bench.DeviceEngine(N, T, 0, B)
dev = torch.device("cuda", 0); f64 = dict(dtype=torch.float64, device=dev)
z0 = torch.from_numpy(np.ascontiguousarray(z0h)).to(dev); lb = torch.from_numpy(np.ascontiguousarray(lbh)).to(dev); ub = torch.from_numpy(np.ascontiguousarray(ubh)).to(dev)
lam = torch.empty(B, eng.m, **f64); kkt = torch.empty(B, 3, **f64); cost = torch.empty(B, **f64)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
def run(maxit):
  eng.opts.max_iter = maxit; eng.opts.restoration = 0
  z = z0.clone(); torch.cuda.synchronize()
  eng.solve(B, z, lb, ub, lam, cost, st, it, kkt); torch.cuda.synchronize()
  return it.cpu().numpy().astype(float), kkt.cpu().numpy().copy(), cost.cpu().numpy().copy()
full, _, _ = run(1000)
def makespan(L, slots=1024):
  h = [0.0] * slots; heapq.heapify(h)
  for x in L:
    t = heapq.heappop(h); heapq.heappush(h, t + x)
  return max(h)
print("one phase: makespan %.0f, ideal %.1f" % (makespan(full), full.sum() / 1024))
for K1 in (6, 8, 10, 12, 14):
  _, k, c = run(K1)
  rem = np.maximum(full - K1, 0.0)
  feats = {"feas": k[:, 0], "stat": k[:, 1], "compl": k[:, 2], "cost": c}
  ph1 = 4 * K1 if (full >= K1).all() else makespan(np.minimum(full, K1))
  line = ["K1=%d: phase 1 %.0f, remaining ideal %.1f, ticket order %.0f, perfect LPT %.0f" % (K1, ph1, rem.sum() / 1024, ph1 + makespan(rem), ph1 + makespan(rem[np.argsort(-rem)]))]
  for n, v in feats.items():
    r = spearmanr(v, rem).correlation
    line.append("%s rho %+.2f -> %.0f / rev %.0f" % (n, r, ph1 + makespan(rem[np.argsort(-v)]), ph1 + makespan(rem[np.argsort(v)])))
  print("  ".join(line), flush=True)
  np.savez("gpurun_out/exp30/k%d.npz" % K1, kkt=k, cost=c, full=full)
PY
cat gpurun_out/exp30/out.txt
