"""exp77: what four wavefronts per CU cost each other.  The headline kernel, one wavefront per trajectory (MYRIAD_FUSED_WAVES=1), whole launches of exactly
twelve iterations (max_iter = 12: nobody finishes early), 1 / 2 / 3 / 4 trajectories per CU: the launch time is the time of twelve iterations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ["MYRIAD_FUSED_WAVES"] = "1"; os.environ["MYRIAD_PARK_ITER"] = "0"; os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
from myriad_amd import _lib
from bench import build_workload
for B in (64, 256, 512, 768, 1024, 2048):
  x0, z0, lb, ub, T = build_workload(B, 100, 2019)
  eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, T, max_batch=B)
  o = eng.default_opts(); o.restoration = 0; o.max_iter = 12
  best = 1e9
  for rep in range(4):
    eng.kernel_time_reset()
    r = eng.solve(z0, lb, ub, opts=o)
    ms, n = eng.kernel_time(_lib.K_SOLVE)
    best = min(best, ms)
  print(f"B={B:5d} ({B / 256:.2f} wavefronts per CU): {best:.3f} ms for 12 iterations = {best / 12 * 1e3:.1f} us per iteration; iterations {int(r['iters'].min())}..{int(r['iters'].max())}", flush=True)
  eng.close()
# Result (one MI355X): 159 us per iteration at 0.25 wavefronts per CU, 163 at 1, 169 at 2, 178 at 3, 187 at 4 (371 at B = 2048: two rounds): four co-resident
# wavefronts cost each other 17 %; the first twelve iterations are expensive by themselves (1.45 sweeps per iteration, twice the line-search trials of the later ones).
