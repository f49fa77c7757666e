"""Host mirror of /root/reference/myriad/nlp_solvers/extra_gradient.py:10-84 -- same keyword interface, same
iteration schedule (convergence test and 0.1 % step decay every 1000 iterations), same result mapping -- with the
iteration itself on the GPU: the iterate stays resident in LDS inside `myr_exgd` (csrc/colloc_products.h) for up to
1000 steps per call instead of one jitted `step` per Python iteration.

Differences by construction: no TensorBoard writer (the reference opens one at import, :7); `fun`/`constraints` are
not called by the iteration (the device owns the transcription), only for the returned 'fun'."""
from __future__ import annotations

import numpy as np


def extra_gradient(fun, x0, method, constraints, bounds, jac, options, *, optimizer=None, params=None):
  """`optimizer` carries the device-side problem descriptor (its engine); everything else is the reference's
  scipy.optimize.minimize-style argument list (extra_gradient.py:10)."""
  del method, jac
  if optimizer is None:
    raise ValueError("the extragradient solver needs the device-side problem descriptor (optimizer=...)")
  eng = optimizer.engine
  max_iter = options['maxiter'] if 'maxiter' in options else 30_000
  eta_x = options['eta_x'] if 'eta_x' in options else 1e-1    # primals   (:17)
  eta_v = options['eta_v'] if 'eta_v' in options else 1e-3    # duals     (:18)
  atol = options['atol'] if 'atol' in options else 1e-6       # convergence tolerance (:19)
  bounds = np.asarray(bounds, dtype=np.float64)
  lb, ub = bounds[:, 0], bounds[:, 1]

  x = np.asarray(x0, dtype=np.float64).copy()
  lmbda = np.ones(eng.m)                                      # :77
  x_old = x + 20                                              # :40 "so we don't terminate immediately"
  success = False
  i = 0
  while i < max_iter:
    # i is a multiple of 1000 here: the reference's checks at i % 1000 == 0 (:52-60)
    if np.allclose(x_old, x, rtol=0., atol=atol):
      success = True
      break
    eta_x *= 0.999
    eta_v *= 0.999
    chunk = min(1000, max_iter - i)
    # the reference compares the iterates before and after step i-1 at the next check: run chunk-1 steps, keep that
    # iterate as x_old, then the last step
    if chunk > 1:
      z, lam = eng.exgd(x, lmbda, lb, ub, eta_x, eta_v, chunk - 1, params=params)
      x, lmbda = z[0], lam[0]
    x_old = x
    z, lam = eng.exgd(x, lmbda, lb, ub, eta_x, eta_v, 1, params=params)
    x, lmbda = z[0], lam[0]
    i += chunk
    if np.isnan(x).any() or np.isnan(lmbda).any():            # :65-69
      print("WE GOT NANS")
      raise SystemExit
  return {'x': x, 'v': lmbda, 'fun': fun(x), 'success': success}
