import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
system, intervals, cpi, method = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
import importlib; T = importlib.import_module("test_gpu_solve")
rng = np.random.default_rng(7); B = 96
x_0 = np.array(T._shoot_opt(system, intervals, cpi, method).system.x_0, float)
if system == "VANDERPOL": x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4)
else: x0 = x_0 * (1 + 0.05 * rng.standard_normal((B, len(x_0)))) + (0.02 * rng.standard_normal((B, len(x_0))) if system == "CARTPOLE" else 0.0)
out = {}
for mode in ("wave", "lane"):
  os.environ["MYRIAD_SOLVE_MODE"] = mode
  out[mode] = T._shoot_opt(system, intervals, cpi, method).solve_batch(x0s=x0)
w, l = out["wave"], out["lane"]
bad = np.nonzero(w["status"] != l["status"])[0]
print(sys.argv[1:], "status mismatch", bad, w["status"][bad], l["status"][bad], "iters", w["iters"][bad], l["iters"][bad], "cost", w["cost"][bad], l["cost"][bad])
ok = (w["status"] == 0) & (l["status"] == 0)
print(" conv", (w["status"] == 0).mean(), (l["status"] == 0).mean(), "same iters", (w["iters"][ok] == l["iters"][ok]).mean(), "cost rel", np.max(np.abs(w["cost"][ok] - l["cost"][ok]) / np.abs(l["cost"][ok])),
      "z", np.abs(w["xs_and_us"][ok] - l["xs_and_us"][ok]).max(), "lam", np.abs(w["lambda"][ok] - l["lambda"][ok]).max(), np.abs(l["lambda"][ok]).max())
