#!/bin/bash
# exp29: ROCKETLANDING's elastic twin, Hermite-Simpson N = 6: lane against wavefront kernel, iteration by iteration
cd $GRAFT_REPO_ROOT
python - <<'PY'
import os, numpy as np
from myriad_amd import _lib
from oracle import myriad_oracle as O
s = O.Elastic(O.SYSTEMS["ROCKETLANDING"](), 1.0)
for N in (6, 20):
  tr = O.hermite_simpson(s, N)
  for lim in range(0, 13):
    r = {}
    for mode in ("wave", "lane"):
      os.environ["MYRIAD_SOLVE_MODE"] = mode
      eng = _lib.Engine("ROCKETLANDING_ELASTIC", "HERMITE_SIMPSON", N, s.T)
      o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
      r[mode] = eng.solve(tr.guess[None], tr.bounds[None, :, 0], tr.bounds[None, :, 1], params=s.params(), opts=o); eng.close()
    w, l = r["wave"], r["lane"]
    d = np.abs(w["z"] - l["z"]) / np.maximum(1.0, np.abs(l["z"]))
    print(f"N={N} it={lim}: cost wave {w['cost'][0]:.12g} lane {l['cost'][0]:.12g}  max rel dz {d.max():.3e}  kkt wave {w['kkt'][0]} lane {l['kkt'][0]}")
PY
