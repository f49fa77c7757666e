"""exp65: where do the trapezoidal elastic twins stall?  kkt residuals (feasibility, stationarity, complementarity) after k iterations of the first problem
of the elastic phase (rho = 1, the reference's guess widened by s = 0), trapezoidal against Hermite-Simpson."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import myriad_oracle as O
from myriad_amd import _lib
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
for base in sys.argv[1:] or ["VANDERPOL", "PENDULUM", "MOUNTAINCAR"]:
  for rule, N in (("TRAPEZOIDAL", 20), ("HERMITE_SIMPSON", 20)):
    for rho in (1.0, 100.0):
      s = O.Elastic(O.SYSTEMS[base](), rho)
      tr = O.hermite_simpson(s, N) if rule == "HERMITE_SIMPSON" else O.trapezoidal(s, N)
      eng = _lib.Engine(base + "_ELASTIC", rule, N, s.T)
      line = []
      for lim in (5, 10, 20, 40, 80, 160, 320, 500):
        o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
        r = eng.solve(tr.guess[None], tr.bounds[None, :, 0], tr.bounds[None, :, 1], params=s.params(), opts=o)
        line.append(f"{lim}: st {int(r['status'][0])} it {int(r['iters'][0])} f {r['cost'][0]:.6g} kkt " + " ".join(f"{v:.1e}" for v in r["kkt"][0]))
        if r["status"][0] == 0: break
      print(base, rule, "rho", rho, "\n   " + "\n   ".join(line), flush=True)
      eng.close()
