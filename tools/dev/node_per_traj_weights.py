import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_node as T
rng = np.random.default_rng(5); B = 10
x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
out = {}
for v in ("1", "0"):
  os.environ["MYRIAD_NODE_COOP"] = v
  hp, node, opt = T._setup(20)
  p = np.tile(opt.system.device_params(), (B, 1))
  out[v] = opt.solve_batch(x0s=x0, params=p)
a, b = out["1"], out["0"]
print("per-trajectory weights: status", a["status"], b["status"], "iters equal", (a["iters"] == b["iters"]).all(), "cost diff", np.abs(a["cost"] - b["cost"]).max())
