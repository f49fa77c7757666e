import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MYRIAD_SOLVE_MODE"] = sys.argv[1]
import numpy as np
from myriad_amd import _lib
if os.environ.get("MYR_LIB"): _lib.LIB_PATH = os.path.abspath(os.environ["MYR_LIB"])
from bench import build_workload
N = int(sys.argv[2]); B = int(sys.argv[3])
x0, z0, lb, ub, T = build_workload(B, N, 2019)
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
res = eng.solve(z0, lb, ub)
print(sys.argv[1], "status", np.bincount(res["status"], minlength=4), "iters", res["iters"][:8], "kkt", res["kkt"][:2], "cost", res["cost"][:3])
