"""TEST INFRASTRUCTURE (like everything under oracle/): Newton polish of a solution of the oracle's SciPy path on the KKT
system of the restated NLP, so that golden fixtures hold z* AND the multipliers lam* to round-off ("goldens
must be generated with tightened tolerances and then polished (Newton on the KKT system in fp64)").

The NLP is the reference's:  min f(z)  s.t.  c(z) = 0,  lb <= z <= ub  (myriad/nlp_solvers/__init__.py:31-42 hands exactly these
callables to the solver); multipliers follow the sign convention of the `mult_g` the reference returns
(myriad/nlp_solvers/__init__.py:82-86): the Lagrangian is f + lam . c, stationarity reads grad f + J^T lam = zL - zU.
Active set: pinned variables (lb == ub) and variables SciPy left on a bound; on it the variables are frozen, elsewhere
  [ H  J_F^T ] [dz_F ]   [ grad f_F + J_F^T lam ]
  [ J_F  0   ] [dlam ] = -[ c                    ]
with the exact Hessian of the Lagrangian (torch autodiff) -- plain Newton, quadratic from a 1e-9-accurate start."""
from __future__ import annotations

import numpy as np
import torch

from oracle import myriad_oracle as O


def polish(tr, z, tol=1e-12, max_newton=8, act_tol=1e-9, verbose=False):
  """-> dict(z, lam, zL, zU, kkt, active): KKT point next to `z` with ||(grad f + J^T lam)_free, c||_inf <= tol."""
  z = np.array(z, dtype=np.float64)
  lb, ub = tr.bounds[:, 0], tr.bounds[:, 1]
  with np.errstate(invalid="ignore"):
    at_l = np.isfinite(lb) & (np.abs(z - lb) <= act_tol * np.maximum(1.0, np.abs(lb)))
    at_u = np.isfinite(ub) & (np.abs(z - ub) <= act_tol * np.maximum(1.0, np.abs(ub)))
  act = at_l | at_u
  z[at_l] = lb[at_l]; z[at_u & ~at_l] = ub[at_u & ~at_l]
  free = ~act
  cb = O.Callbacks(tr)
  m = cb.cons(z).shape[0]

  def lagr(zz, lam):
    return tr.objective(zz) + torch.dot(lam, tr.constraints(zz))
  hess = torch.func.hessian(lagr, argnums=0)
  J = cb.jac(z)
  lam = np.linalg.lstsq(J[:, free].T, -cb.grad(z)[free], rcond=None)[0]
  res = np.inf
  for it in range(max_newton):
    g = cb.grad(z); J = cb.jac(z); c = cb.cons(z)
    r = np.concatenate([(g + J.T @ lam)[free], c])
    res = np.abs(r).max()
    if verbose:
      print(f"  polish it {it}: |KKT|_inf = {res:.3e}")
    if res <= tol:
      break
    H = hess(O._t(z), O._t(lam)).numpy()
    nf = int(free.sum())
    K = np.zeros((nf + m, nf + m))
    K[:nf, :nf] = H[np.ix_(free, free)]
    K[:nf, nf:] = J[:, free].T
    K[nf:, :nf] = J[:, free]
    d = np.linalg.solve(K, -r)
    z[free] += d[:nf]; lam = lam + d[nf:]
  g = cb.grad(z); J = cb.jac(z)
  mult = g + J.T @ lam                       # = zL - zU on the active set
  zL = np.where(act & at_l, np.maximum(mult, 0.0), 0.0)
  zU = np.where(act & at_u & ~at_l, np.maximum(-mult, 0.0), 0.0)
  pinned = lb == ub
  wrong = act & ~pinned & (((at_l) & (mult < -1e-8)) | ((at_u & ~at_l) & (mult > 1e-8)))
  inside = ((z >= lb - 1e-12) & (z <= ub + 1e-12)).all()
  return {"z": z, "lam": lam, "zL": zL, "zU": zU, "kkt": float(res), "active": act, "wrong_sign": int(wrong.sum()), "inside": bool(inside),
          "cost": cb.fun(z)}
