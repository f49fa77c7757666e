#!/usr/bin/env python3
"""Generate tests/golden/solve_shoot_*.npz: golden optima of the shooting transcriptions of BASELINE configs 1, 3, 4
 from the ORACLE's SciPy SLSQP path (the reference's NLPSolverType.SLSQP branch,
/root/reference/myriad/nlp_solvers/__init__.py:50-52, on the restated callbacks), stopping tolerance tightened to
ftol=1e-15, then polished by Newton on the KKT system with the exact Lagrangian Hessian (oracle/polish.py) to |KKT|_inf <= 1e-12;
`lam` are the equality multipliers in the sign convention of the reference's mult_g (nlp_solvers/__init__.py:82-86: L = f + lam . c),
zL / zU the bound multipliers left on the active set.  CANCERTREATMENT carries the default parameters and 4 points of the config-4 sweep rule.
Run from the repo root:  python tests/golden/make_shoot_golden.py
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import myriad_oracle as O
from oracle.polish import polish

OUT = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(2019)
sweep = [dict(r=rng.uniform(0.1, 0.5), delta=rng.uniform(0.2, 0.8), a=rng.uniform(1, 5), x_0=rng.uniform(0.5, 0.99)) for _ in range(4)]
CASES = [
  ("simplecase_10x100", [O.SimpleCase()], dict(intervals=10, cpi=100)),
  ("vanderpol_1x50", [O.VanDerPol()] + [None, None], dict(intervals=1, cpi=50)),
  ("cancertreatment_1x100", [O.CancerTreatment()] + [O.CancerTreatment(**kw) for kw in sweep], dict(intervals=1, cpi=100)),
]
for tag, systems, kw in CASES:
  if tag.startswith("vanderpol"):
    x0s = O.random_x0(O.VanDerPol(), 2, seed=2019)
    for i in (1, 2):
      s = O.VanDerPol(); s.x_0 = x0s[i - 1].copy(); systems[i] = s
  rows = dict(z=[], z0=[], lb=[], ub=[], cost=[], feas=[], nit=[], params=[], x0=[], lam=[], zL=[], zU=[], kkt=[])
  for s in systems:
    tr = O.shooting(s, kw["intervals"], kw["cpi"], "HEUN")
    cb = O.Callbacks(tr)
    t0 = time.time()
    r = O.solve(tr, "SLSQP", max_iter=2000, extra_options={"ftol": 1e-15}, cb=cb)
    pz = polish(tr, r["xs_and_us"])
    assert pz["kkt"] <= 1e-12 and pz["wrong_sign"] == 0 and pz["inside"], pz
    print(f"  polish: |z - z_slsqp| = {np.abs(pz['z'] - r['xs_and_us']).max():.1e}, |KKT| = {pz['kkt']:.1e}")
    z = pz["z"]; r["cost"] = pz["cost"]
    rows["lam"].append(pz["lam"]); rows["zL"].append(pz["zL"]); rows["zU"].append(pz["zU"]); rows["kkt"].append(pz["kkt"])
    rows["z"].append(z); rows["z0"].append(tr.guess); rows["lb"].append(tr.bounds[:, 0]); rows["ub"].append(tr.bounds[:, 1])
    rows["cost"].append(r["cost"]); rows["feas"].append(np.abs(cb.cons(z)).max()); rows["nit"].append(r["scipy"].nit)
    rows["params"].append(s.params()); rows["x0"].append(s.x_0)
    print(f"{tag} cost={r['cost']:.14f} feas={rows['feas'][-1]:.2e} nit={rows['nit'][-1]} {time.time() - t0:.1f}s", flush=True)
  path = os.path.join(OUT, f"solve_shoot_{tag}.npz")
  np.savez_compressed(path, T=systems[0].T, **kw, **{k: np.stack(v) for k, v in rows.items()})
  print("wrote", path, os.path.getsize(path))
