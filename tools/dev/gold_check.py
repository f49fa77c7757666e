import numpy as np, glob, os, sys
sys.path.insert(0, '' + os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + '')
from myriad_amd import _lib
for path in sorted(glob.glob('' + os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))) + '/tests/golden/solve_hs_cartpole_N*.npz')):
  d = np.load(path); N = int(d["N"])
  eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, 2.0, max_batch=8)
  res = eng.solve(d["z0"], d["lb"], d["ub"])
  print(os.path.basename(path), "rel cost diff", (res["cost"] - d["cost"]) / d["cost"], "max|dz|", np.abs(res["z"] - d["z"]).max(axis=1), "iters", res["iters"], "kkt", res["kkt"])
