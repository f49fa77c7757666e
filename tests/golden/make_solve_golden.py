#!/usr/bin/env python3
"""Generate tests/golden/solve_hs_cartpole_N*.npz: golden optimal trajectories AND multipliers for the CARTPOLE Hermite-Simpson
NLP.  Two steps, both on the ORACLE's restated callbacks (torch autodiff of oracle/myriad_oracle.py):
  1. the oracle's SciPy SLSQP path (the reference's NLPSolverType.SLSQP branch, /root/reference/myriad/nlp_solvers/__init__.py:50-52)
     with the stopping tolerance tightened (ftol=1e-15 for N <= 25, 1e-10 at N = 100 where SLSQP's last digits take hours) -- at the
     reference's default ftol=1e-6 two correct solvers differ by 1e-2 in u, so default-tolerance solutions are
     useless as z* goldens;
  2. Newton polish on the KKT system with the exact Lagrangian Hessian (oracle/polish.py) to |KKT|_inf <= 1e-12; the equality
     multipliers `lam` use the sign convention of the reference's mult_g (nlp_solvers/__init__.py:82-86: L = f + lam . c), the
     bound multipliers zL, zU are what is left of grad f + J^T lam on the active set.
x0 instances: the default x_0 plus the SURVEY 8(d) rule  clip(x_0 + 0.1 N(0,I))  from default_rng(2019): 3 instances for
N <= 25, 4 for N = 100 (one SLSQP solve at N = 100 takes 13-16 minutes; the instances run in parallel processes; the polish moves
z by 2-3e-4 there and ends at |KKT| = 1.5-3e-14).
Run from the repo root:  python tests/golden/make_solve_golden.py [N ...]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OUT = os.path.dirname(os.path.abspath(__file__))


def one(args):
  N, b, x0 = args
  import torch
  torch.set_num_threads(2)
  from oracle import myriad_oracle as O
  from oracle.polish import polish
  s = O.CartPole(); s.x_0 = np.array(x0, dtype=np.float64)
  tr = O.hermite_simpson(s, N)
  cb = O.Callbacks(tr)
  t0 = time.time()
  # N <= 25: ftol = 1e-15 (seconds); N = 100: 1e-10 -- the Newton polish below does the rest, and SLSQP's last digits cost it hours there
  r = O.solve(tr, "SLSQP", max_iter=2000, extra_options={"ftol": 1e-15 if N <= 25 else 1e-10}, cb=cb)
  z_slsqp = r["xs_and_us"]
  p = polish(tr, z_slsqp, max_newton=12)
  assert p["kkt"] <= 1e-12 and p["wrong_sign"] == 0 and p["inside"], p
  z = p["z"]
  print(f"N={N} b={b} cost={p['cost']:.14f} feas={np.abs(cb.cons(z)).max():.2e} kkt={p['kkt']:.1e} |z-z_slsqp|={np.abs(z - z_slsqp).max():.1e} "
        f"nit={r['scipy'].nit} {time.time() - t0:.1f}s", flush=True)
  return dict(z=z, lam=p["lam"], zL=p["zL"], zU=p["zU"], kkt=p["kkt"], cost=p["cost"], feas=np.abs(cb.cons(z)).max(), nit=r["scipy"].nit,
              lb=tr.bounds[:, 0], ub=tr.bounds[:, 1], z0=tr.guess)


if __name__ == "__main__":
  from multiprocessing import get_context
  from oracle import myriad_oracle as O
  Ns = [int(a) for a in sys.argv[1:]] or [5, 10, 25]
  for N in Ns:
    nb = 3 if N <= 25 else 4
    base = O.CartPole()
    x0s = np.vstack([base.x_0[None], O.random_x0(base, 8, seed=2019)[:nb - 1]])
    with get_context("spawn").Pool(min(nb, 4)) as pool:
      rs = pool.map(one, [(N, b, x0s[b]) for b in range(nb)])
    path = os.path.join(OUT, f"solve_hs_cartpole_N{N}.npz")
    np.savez_compressed(path, N=N, x0=x0s, **{k: np.stack([np.asarray(r[k]) for r in rs]) for k in rs[0]})
    print("wrote", path)
