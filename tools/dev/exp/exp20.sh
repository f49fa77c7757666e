#!/bin/bash
# exp20: is the iteration count of a cart-pole solve predictable from its start?  (for a longest-first ticket order)
cd /root/repo; mkdir -p gpurun_out/exp20
python - > gpurun_out/exp20/predict.txt 2>&1 <<'PY'
import numpy as np, torch, bench
from scipy.stats import spearmanr
N, B = 100, 4096
x0, z0h, lbh, ubh, T = bench.build_workload(B, N, seed=2019)
eng = bench.DeviceEngine(N, T, 0, B)
dev = torch.device("cuda", 0); f64 = dict(dtype=torch.float64, device=dev)
z0 = torch.from_numpy(np.ascontiguousarray(z0h)).to(dev); lb = torch.from_numpy(np.ascontiguousarray(lbh)).to(dev); ub = torch.from_numpy(np.ascontiguousarray(ubh)).to(dev)
lam = torch.empty(B, eng.m, **f64); kkt = torch.empty(B, 3, **f64); cost = torch.empty(B, **f64)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
fv = torch.empty(B, **f64); gv = torch.empty(B, eng.ngrad, **f64); cv = torch.empty(B, eng.m, **f64); jv = torch.empty(B, eng.jblk, **f64)
eng.eval(B, z0, fv, gv, cv, jv); torch.cuda.synchronize()
c0inf = cv.abs().amax(dim=1).cpu().numpy(); c0l1 = cv.abs().sum(dim=1).cpu().numpy(); f0 = fv.cpu().numpy()
z = z0.clone(); torch.cuda.synchronize()
eng.solve(B, z, lb, ub, lam, cost, st, it, kkt); torch.cuda.synchronize()
i = it.cpu().numpy().astype(float)
print("iters: mean %.2f median %g p90 %g p99 %g max %g; >25: %d  >30: %d" % (i.mean(), np.median(i), np.percentile(i, 90), np.percentile(i, 99), i.max(), (i > 25).sum(), (i > 30).sum()))
nom = np.median(x0, axis=0)
feats = {"c0inf": c0inf, "c0l1": c0l1, "f0": f0, "cost*": cost.cpu().numpy(), "|dx|": np.linalg.norm(x0 - nom, axis=1)}
for k in range(x0.shape[1]):
  feats["x0[%d]" % k] = x0[:, k]; feats["|x0[%d]-nom|" % k] = np.abs(x0[:, k] - nom[k])
for k, v in feats.items():
  r = spearmanr(v, i).correlation
  print("%-14s spearman %+.3f" % (k, r))
# what an order would buy: list-scheduling makespan over 1024 slots, jobs in ticket order
import heapq
def makespan(order, slots=1024):
  h = [0.0] * slots; heapq.heapify(h)
  for j in order:
    t = heapq.heappop(h); heapq.heappush(h, t + i[j])
  return max(h)
print("makespan (iterations): ticket order %.0f, perfect longest-first %.0f, ideal %.1f" % (makespan(range(B)), makespan(np.argsort(-i)), i.sum() / 1024))
for k, v in feats.items():
  print("  longest-first by %-14s: %.0f   (reverse: %.0f)" % (k, makespan(np.argsort(-v)), makespan(np.argsort(v))))
np.savez("gpurun_out/exp20/iters.npz", x0=x0, iters=i, c0inf=c0inf, c0l1=c0l1, f0=f0)
PY
cat gpurun_out/exp20/predict.txt
