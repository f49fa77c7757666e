# dev tool: per-iteration time of the wave solve kernel vs the number of resident wavefronts, identical instances
# (every trajectory takes the same iterations, so kernel time / iterations / rounds is the loaded iteration time)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd import _lib
from bench import build_workload
x0, z0, lb, ub, T = build_workload(8, 100, 2019)
for B in (64, 128, 256, 512, 768, 1024, 2048, 4096):
  eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, T, max_batch=B)
  Z = np.repeat(z0[:1], B, 0); L = np.repeat(lb[:1], B, 0); U = np.repeat(ub[:1], B, 0)
  eng.solve(Z, L, U)
  eng.kernel_time_reset()
  res = eng.solve(Z, L, U)
  ms, n = eng.kernel_time(_lib.K_SOLVE)
  it = int(res["iters"][0])
  rounds = max(1.0, B / 1024)
  print(f"B={B:5d} kernel {ms/n:8.3f} ms  iters {it}  ms/iter/round {ms/n/it/rounds:.4f}", flush=True)
  eng.close()
