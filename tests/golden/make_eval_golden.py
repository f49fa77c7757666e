#!/usr/bin/env python3
"""Generate tests/golden/eval_*.npz: inputs z and expected f, grad f, c, J (as stage blocks) from the
oracle (torch-fp64 restatement of the reference's transcription), cross-checked here against central
finite differences.  The reference itself cannot be imported (no JAX in this image), so these
vectors are oracle outputs -- 'parity unpinned' at the JAX boundary (see oracle/myriad_oracle.py header).
Run from the repo root:  python tests/golden/make_eval_golden.py
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import myriad_oracle as O

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = [("CARTPOLE", 5, 3), ("CARTPOLE", 100, 1), ("VANDERPOL", 4, 2), ("CANCERTREATMENT", 4, 2), ("SIMPLECASE", 3, 2),
         # SURVEY.md 8(f4) systems
         ("BIOREACTOR", 3, 2), ("GLUCOSE", 3, 2), ("MOULDFUNGICIDE", 3, 2), ("SIMPLECASEWITHBOUNDS", 3, 2), ("HIVTREATMENT", 3, 2),
         ("EPIDEMICSEIRN", 3, 2), ("SEIR", 3, 2), ("BEARPOPULATIONS", 3, 2),
         ("PENDULUM", 3, 3), ("MOUNTAINCAR", 3, 3), ("ROCKETLANDING", 3, 2), ("BACTERIA", 3, 2), ("TUMOUR", 3, 2),
         ("HARVEST", 3, 2), ("TIMBERHARVEST", 3, 2)]
ONLY = set(a.upper() for a in sys.argv[1:])

for name, N, B in CASES:
  if ONLY and name not in ONLY:
    continue
  sysm = O.SYSTEMS[name]()
  tr = O.hermite_simpson(sysm, N)
  cb = O.Callbacks(tr)
  rng = np.random.default_rng(1234 + N)
  scale = 0.25 if name in ("CARTPOLE", "VANDERPOL", "CANCERTREATMENT", "SIMPLECASE") else 0.05
  z = tr.guess[None] * (1.0 + scale * rng.standard_normal((B, tr.guess.size))) + scale * rng.standard_normal((B, tr.guess.size)) \
      if scale == 0.05 else tr.guess[None] + 0.25 * rng.standard_normal((B, tr.guess.size))
  if name in ("PENDULUM", "MOUNTAINCAR"):
    z[2] = 3.0 * tr.guess + 2.5 * rng.standard_normal(tr.guess.size)   # outside the box: clip / angle_normalize branches
  if name == "TUMOUR":
    z = np.abs(z) + 1.0                                       # log(p / q): keep the states positive
  if name == "CANCERTREATMENT":
    z[:, :tr.x_rows] = np.abs(z[:, :tr.x_rows]) + 0.05      # keep x > 0 (log(1/x), cancer_treatment.py:25-29)
  f = np.array([cb.fun(zb) for zb in z])
  g = np.stack([cb.grad(zb) for zb in z])
  c = np.stack([cb.cons(zb) for zb in z])
  J = np.stack([cb.jac(zb) for zb in z])
  # finite-difference cross-check of the autodiff Jacobian / gradient (independent of torch.func)
  eps = 1e-6
  for b in range(B):
    cols = rng.choice(z.shape[1], size=min(12, z.shape[1]), replace=False)
    for j in cols:
      e = np.zeros(z.shape[1]); e[j] = eps
      fd = (cb.cons(z[b] + e) - cb.cons(z[b] - e)) / (2 * eps)
      assert np.abs(fd - J[b][:, j]).max() < 1e-6 * max(1.0, np.abs(J[b][:, j]).max()), (name, j)
      fdg = (cb.fun(z[b] + e) - cb.fun(z[b] - e)) / (2 * eps)
      assert abs(fdg - g[b][j]) < 1e-6 * max(1.0, abs(g[b][j])), (name, j)
  blk = np.stack([O.hs_blocks_from_dense(J[b], N, sysm.ns, sysm.nu) for b in range(B)])
  # the blocks + implied identity must reproduce the dense Jacobian exactly (structure check)
  for b in range(B):
    assert np.array_equal(O.hs_dense_from_blocks(blk[b], N, sysm.ns, sysm.nu), J[b])
  path = os.path.join(OUT, f"eval_hs_{name.lower()}_N{N}.npz")
  np.savez_compressed(path, z=z, f=f, gradf=g, c=c, jblk=blk, N=N, T=sysm.T, params=sysm.params())
  print("wrote", path, os.path.getsize(path))
