# dev tool: the bench workload, two solver launches (for PMC passes)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd import _lib
from bench import build_workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, z0, lb, ub, T = build_workload(B, 100, 2019)
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, T, max_batch=B)
for _ in range(2):
  res = eng.solve(z0, lb, ub)
print("converged", (res["status"] == 0).mean(), eng.kernel_time(_lib.K_SOLVE))
