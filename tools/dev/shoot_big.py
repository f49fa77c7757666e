import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import importlib; T = importlib.import_module("test_gpu_solve")
from myriad_amd import _lib
system, intervals, cpi, method, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
rng = np.random.default_rng(7)
out = {}
for mode in ("wave", "lane"):
  os.environ["MYRIAD_SOLVE_MODE"] = mode
  opt = T._shoot_opt(system, intervals, cpi, method)
  x_0 = np.array(opt.system.x_0, float)
  if mode == "wave": x0 = x_0 * (1 + 0.05 * rng.standard_normal((B, len(x_0)))) + 0.02 * rng.standard_normal((B, len(x_0)))
  opt.solve_batch(x0s=x0); opt.engine.kernel_time_reset()
  out[mode] = opt.solve_batch(x0s=x0)
  ms, n = opt.engine.kernel_time(_lib.K_SOLVE)
  print(mode, "kernel ms %.2f" % ms, "solves/s %.0f" % (B / ms * 1e3), "converged", (out[mode]["status"] == 0).mean(), "iters med/max", np.median(out[mode]["iters"]), out[mode]["iters"].max())
w, l = out["wave"], out["lane"]; ok = (w["status"] == 0) & (l["status"] == 0)
print("cost rel", np.max(np.abs(w["cost"][ok] - l["cost"][ok]) / np.abs(l["cost"][ok])), "same iters", (w["iters"][ok] == l["iters"][ok]).mean())
