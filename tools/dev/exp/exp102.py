"""exp102: ROCKETLANDING's twin, Hermite-Simpson N = 100 (exp67's problem): per-iteration trace of the fused kernel up to the iteration where two of four
trajectories end in NaN while the lane kernel goes on (MYRIAD_HIP_LIB = a -DMYR_TRACE=4 build of the system's Hermite-Simpson object)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import myriad_oracle as O
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
from myriad_amd import _lib
name, rule, N = "ROCKETLANDING_ELASTIC", "HERMITE_SIMPSON", 100
s = O.Elastic(O.SYSTEMS[name[:-8]](), 1.0)
tr = O.hermite_simpson(s, N)
B = 4
rng = np.random.default_rng(3)
z0 = np.tile(tr.guess, (B, 1)); lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))
z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
os.environ["MYRIAD_SOLVE_MODE"] = "wave"
eng = _lib.Engine(name, rule, N, s.T)
o = eng.default_opts(); o.restoration = 0; o.max_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 9
r = eng.solve(z0, lb, ub, params=s.params(), opts=o)
print("status", r["status"], "iters", r["iters"])
