# dev tool: time the solve kernel of an alternative library build on the bench workload
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from bench import build_workload
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 100
x0, z0, lb, ub, T = build_workload(B, N, 2019)
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
for _ in range(3):
  res = eng.solve(z0, lb, ub)
print(sys.argv[1], "converged", (res["status"] == 0).mean(), "kernel ms avg", eng.kernel_time(_lib.K_SOLVE))
