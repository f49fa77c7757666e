# dev: how much of the B=4096 makespan is the tail?  Solve once, then re-solve with the instances ordered by their
# (now known) iteration counts, longest first / shortest first / interleaved.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd import _lib
from bench import build_workload
B = 4096
x0, z0, lb, ub, T = build_workload(B, 100, 2019)
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, T, max_batch=B)
def run(order, tag):
  for _ in range(2):
    eng.kernel_time_reset()
    res = eng.solve(z0[order], lb[order], ub[order])
  print(tag, "kernel ms", eng.kernel_time(_lib.K_SOLVE)[0], "converged", (res["status"] == 0).mean())
  return res
res = run(np.arange(B), "natural")
it = res["iters"]
run(np.argsort(-it, kind="stable"), "longest first")
run(np.argsort(it, kind="stable"), "shortest first")
o = np.argsort(-it, kind="stable"); top = o[:1024]; rest = np.random.default_rng(0).permutation(o[1024:])
run(np.concatenate([top, rest]), "longest quarter first, rest random")
d0 = np.abs(x0 - np.array([0., 0., 0., 0.])).sum(1)
run(np.argsort(-d0, kind="stable"), "by |x0|_1 descending")
