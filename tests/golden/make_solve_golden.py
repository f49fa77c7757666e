#!/usr/bin/env python3
"""Generate tests/golden/solve_hs_cartpole_N*.npz: golden optimal trajectories for the CARTPOLE Hermite-Simpson
NLP, produced by the ORACLE's SciPy SLSQP path (the reference's NLPSolverType.SLSQP branch,
/root/reference/myriad/nlp_solvers/__init__.py:50-52, on the restated callbacks) with the stopping tolerance
tightened to ftol=1e-15 -- at the reference's default ftol=1e-6 two correct solvers differ by 1e-2 in u
(SURVEY.md App. C), so default-tolerance solutions are useless as z* goldens.
x0 instances: the default x_0 plus the SURVEY 8(d) rule  clip(x_0 + 0.1 N(0,I))  from default_rng(2019).
Run from the repo root:  python tests/golden/make_solve_golden.py [N ...]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import myriad_oracle as O

OUT = os.path.dirname(os.path.abspath(__file__))
Ns = [int(a) for a in sys.argv[1:]] or [5, 10, 25]
for N in Ns:
  nb = 3 if N <= 25 else 1
  base = O.CartPole()
  x0s = np.vstack([base.x_0[None], O.random_x0(base, 8, seed=2019)[:nb - 1]]) if nb > 1 else base.x_0[None]
  zs, costs, feas, nit, lbs, ubs, z0s = [], [], [], [], [], [], []
  for b in range(nb):
    s = O.CartPole(); s.x_0 = x0s[b].copy()
    tr = O.hermite_simpson(s, N)
    cb = O.Callbacks(tr)
    t0 = time.time()
    r = O.solve(tr, "SLSQP", max_iter=2000, extra_options={"ftol": 1e-15}, cb=cb)
    z = r["xs_and_us"]
    zs.append(z); costs.append(r["cost"]); feas.append(np.abs(cb.cons(z)).max()); nit.append(r["scipy"].nit)
    lbs.append(tr.bounds[:, 0]); ubs.append(tr.bounds[:, 1]); z0s.append(tr.guess)
    print(f"N={N} b={b} cost={r['cost']:.14f} feas={feas[-1]:.2e} nit={nit[-1]} {time.time() - t0:.1f}s", flush=True)
  path = os.path.join(OUT, f"solve_hs_cartpole_N{N}.npz")
  np.savez_compressed(path, N=N, x0=x0s, z0=np.stack(z0s), lb=np.stack(lbs), ub=np.stack(ubs), z=np.stack(zs),
                      cost=np.array(costs), feas=np.array(feas), nit=np.array(nit))
  print("wrote", path)
