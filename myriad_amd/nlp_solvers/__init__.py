"""Host mirror of /root/reference/myriad/nlp_solvers/__init__.py:18-98 with the new solver plugged in at the
reference's two extension markers (:14 "Import your new nlp solver here", :44 "Add new nlp solvers to this list")."""
from __future__ import annotations

import time
from typing import Dict

import numpy as np

from myriad_amd.config import Config, HParams, NLPSolverType


def solve(hp: HParams, cfg: Config, opt_dict: Dict) -> Dict[str, np.ndarray]:
  """Same contract as the reference: returns {'x','u','xs_and_us','cost'[, 'lambda']}.  Non-convergence is printed,
  never raised (nlp_solvers/__init__.py:64).  Unknown solver -> ValueError (:59-61)."""
  _t1 = time.time()
  opt = opt_dict.get('optimizer')
  if hp.nlpsolver in (NLPSolverType.SQP, NLPSolverType.IPOPT):
    # NEW branch.  IPOPT (cyipopt) does not exist on this platform; its slot is served by the batched SQP, which
    # owns the transcription on the device and therefore takes the problem descriptor instead of Python callables.
    if opt is None:
      raise ValueError("the SQP solver needs opt_dict['optimizer'] (the device-side problem descriptor)")
    eng = opt.engine
    o = eng.default_opts()
    o.max_iter = hp.max_iter
    res = opt.device_solve(np.asarray(opt_dict['guess'], dtype=np.float64)[None], opt_dict['bounds'][None, :, 0],
                           opt_dict['bounds'][None, :, 1], opt_dict.get('params'), o,
                           second_starts=not opt_dict.get('explicit_guess', False))         # second starts: the reference guess only
    solution = {'x': res["z"][0], 'fun': float(res["cost"][0]), 'success': bool(res["status"][0] == 0), 'v': res["lam"][0],
                'nit': int(res["iters"][0])}
  elif hp.nlpsolver in (NLPSolverType.SLSQP, NLPSolverType.TRUST):
    # the reference's SciPy branches (:50-55), fed by the GPU callbacks (small problems only: dense Jacobian)
    from scipy.optimize import minimize
    if opt is None:
      raise ValueError("opt_dict['optimizer'] is required")
    pm = opt_dict.get('params_map')     # solve_with_params: the derivatives see the same parameters as fun / constraints
    inputs = {'fun': opt_dict['objective'], 'x0': opt_dict['guess'],
              'constraints': ({'type': 'eq', 'fun': opt_dict['constraints'], 'jac': lambda z: opt.constraints_jac(z, params=pm)}),
              'bounds': opt_dict['bounds'], 'jac': lambda z: opt.objective_grad(z, params=pm), 'options': {'maxiter': hp.max_iter},
              'method': 'SLSQP' if hp.nlpsolver == NLPSolverType.SLSQP else 'trust-constr'}
    solution = minimize(**inputs)
  elif hp.nlpsolver == NLPSolverType.EXTRAGRADIENT:
    # reference branch :45-49; the iteration runs on the device (collocation transcriptions)
    from myriad_amd.defaults import learning_rates
    from myriad_amd.nlp_solvers.extra_gradient import extra_gradient
    options = {'maxiter': hp.max_iter}
    if hp.system in learning_rates:
      options = {**options, **learning_rates[hp.system]}
    solution = extra_gradient(fun=opt_dict['objective'], x0=opt_dict['guess'], method='exgd',
                              constraints={'type': 'eq', 'fun': opt_dict['constraints']}, bounds=opt_dict['bounds'],
                              jac=None, options=options, optimizer=opt, params=opt_dict.get('params'))
  else:
    print("Unknown NLP solver. Please choose among", list(NLPSolverType.__members__.keys()))
    raise ValueError
  _t2 = time.time()
  if cfg.verbose:
    print('Solver exited with success:', solution['success'])
    print(f'Completed in {_t2 - _t1} seconds.')
    print('Cost given by solver:', solution['fun'])
  lmbda = None
  if hp.nlpsolver in (NLPSolverType.IPOPT, NLPSolverType.SQP, NLPSolverType.TRUST, NLPSolverType.EXTRAGRADIENT):
    lmbda = solution['v']
    if isinstance(lmbda, list):
      lmbda = lmbda[0]
  x, u = opt_dict['unravel'](solution['x'])
  results = {'x': x, 'u': u, 'xs_and_us': solution['x'], 'cost': solution['fun']}
  if lmbda is not None:
    results['lambda'] = lmbda
  return results
