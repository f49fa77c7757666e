"""One wider system on the fused kernel's block sweep, same inputs as tools/dev/exp/exp64.py: python tools/dev/wider_one.py SYSTEM RULE B MAXITER [reps]
(the target of the counter passes in tools/profile_r05_wider.sh)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import myriad_oracle as O
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
from myriad_amd import _lib
name, rule, B, lim = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
N = 100
twin = name.endswith("_ELASTIC")
s = O.Elastic(O.SYSTEMS[name[:-8]](), 1.0) if twin else O.SYSTEMS[name]()
tr = O.hermite_simpson(s, N) if rule == "HERMITE_SIMPSON" else O.trapezoidal(s, N)
rng = np.random.default_rng(3)
z0 = np.tile(tr.guess, (B, 1)); lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))
z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
eng = _lib.Engine(name, rule, N, s.T)
o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
for rep in range(reps):
  eng.kernel_time_reset()
  r = eng.solve(z0, lb, ub, params=s.params() if twin else None, opts=o)
  ms, n = eng.kernel_time(_lib.K_SOLVE)
  print(f"{name} {rule} N={N} B={B}: solver kernels {ms:.3f} ms in {n} launches, {int(r['iters'].sum())} iterations, {int((r['status'] == 0).sum())} converged", flush=True)
