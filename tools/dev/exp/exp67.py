"""exp67: ROCKETLANDING's twin, Hermite-Simpson N = 100: the fused kernel (block sweep) against the lane kernel, iteration by iteration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import myriad_oracle as O
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
from myriad_amd import _lib
name, rule, N = "ROCKETLANDING_ELASTIC", sys.argv[1] if len(sys.argv) > 1 else "HERMITE_SIMPSON", int(sys.argv[2]) if len(sys.argv) > 2 else 100
s = O.Elastic(O.SYSTEMS[name[:-8]](), 1.0)
tr = O.hermite_simpson(s, N) if rule == "HERMITE_SIMPSON" else O.trapezoidal(s, N)
B = 4
rng = np.random.default_rng(3)
z0 = np.tile(tr.guess, (B, 1)); lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))
z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
for lim in (1, 2, 4, 6, 8, 10, 12, 16, 20, 30):
  out = {}
  for mode in ("wave", "lane"):
    os.environ["MYRIAD_SOLVE_MODE"] = mode
    eng = _lib.Engine(name, rule, N, s.T)
    o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
    out[mode] = eng.solve(z0, lb, ub, params=s.params(), opts=o)
    eng.close()
  f, l = out["wave"], out["lane"]
  fin = np.isfinite(f["z"]) & np.isfinite(l["z"])
  d = (np.abs(f["z"] - l["z"])[fin] / np.maximum(1.0, np.abs(l["z"])[fin])).max(initial=0.0)
  print(f"max_iter {lim:3d}: fused status {f['status']} iters {f['iters']} cost {f['cost'][0]:.9g} kkt {f['kkt'][0]}   lane status {l['status']} iters {l['iters']} cost {l['cost'][0]:.9g} kkt {l['kkt'][0]}   |dz| {d:.2e}", flush=True)
