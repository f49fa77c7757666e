"""exp70: ROCKETLANDING Hermite-Simpson at N = 1, 2: fused kernel (block sweep) / round 2's wavefront kernel / lane kernel, iteration by iteration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import myriad_oracle as O
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
from myriad_amd import _lib
name, rule = "ROCKETLANDING", "HERMITE_SIMPSON"
s = O.SYSTEMS[name]()
for N in (1, 2):
  tr = O.hermite_simpson(s, N)
  rng = np.random.default_rng(100 + N)
  B = 3
  z0 = np.tile(tr.guess, (B, 1)); lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
  x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))
  z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
  for lim in (1, 2, 3):
    out = {}
    for mode in ("wave", "wave1", "lane"):
      os.environ["MYRIAD_SOLVE_MODE"] = mode
      eng = _lib.Engine(name, rule, N, s.T)
      o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
      out[mode] = eng.solve(z0, lb, ub, opts=o)
      eng.close()
    def d(a, b):
      return (np.abs(out[a]["z"] - out[b]["z"]) / np.maximum(1.0, np.abs(out[b]["z"]))).max()
    print(f"N={N} max_iter={lim}: fused-lane {d('wave', 'lane'):.2e}  round2-lane {d('wave1', 'lane'):.2e}  fused-round2 {d('wave', 'wave1'):.2e}   cost fused {out['wave']['cost']} lane {out['lane']['cost']} kkt fused {out['wave']['kkt'][0]} lane {out['lane']['kkt'][0]}", flush=True)
