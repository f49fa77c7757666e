# dev tool: does the iteration count of the headline workload correlate with a cheap feature of x0?
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd import _lib
from bench import build_workload
B = 4096
x0, z0, lb, ub, T = build_workload(B, 100, 2019)
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, T, max_batch=B)
res = eng.solve(z0, lb, ub)
it = res["iters"].astype(float)
d = x0 - np.array([0., 0., 0., 0.])
feats = {"|d|": np.linalg.norm(d, axis=1), "x": d[:, 0], "th": d[:, 1], "v": d[:, 2], "w": d[:, 3], "|th|": abs(d[:, 1]), "|w|": abs(d[:, 3]),
         "|x|": abs(d[:, 0]), "|v|": abs(d[:, 2])}
for k, f in feats.items():
  print(k, "corr", round(float(np.corrcoef(f, it)[0, 1]), 3))
A = np.column_stack([d, d ** 2, np.ones(B)])
coef, *_ = np.linalg.lstsq(A, it, rcond=None)
pred = A @ coef
print("quadratic fit R^2", 1 - ((it - pred) ** 2).sum() / ((it - it.mean()) ** 2).sum())
top = np.argsort(-pred)[:B // 4]
print("of the 100 longest, in predicted top quarter:", np.isin(np.argsort(-it)[:100], top).mean())
print("iters hist", np.percentile(it, [50, 90, 99, 100]))
ev = eng.eval(z0, want=("f", "c", "gradf"))
c = ev["c"]
for k, f in {"c1": abs(c).sum(1), "cinf": abs(c).max(1), "c2": (c ** 2).sum(1), "f": ev["f"]}.items():
  r = float(np.corrcoef(f, it)[0, 1])
  top = np.argsort(-f)[:B // 4]
  print(k, "corr", round(r, 3), "100 longest in top quarter by this proxy:", np.isin(np.argsort(-it)[:100], top).mean())
