"""Second, independent pin of the oracle and of the device solver: the continuous-time
optima of SIMPLECASE and CANCERTREATMENT from Pontryagin's conditions, solved as boundary value problems by
scipy.integrate.solve_bvp (tests/golden/make_pontryagin_bvp_golden.py -> pontryagin_bvp.json; no transcription, no NLP
solver, nothing shared with the oracle or the kernels).  The Hermite-Simpson optimum must converge to them at the
rate of the scheme: O(h^4) on the smooth problem, monotonically on the one with an active control bound."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "pontryagin_bvp.json")))


def _oracle_cost(cls, N):
  from oracle import myriad_oracle as O
  s = cls()
  tr = O.hermite_simpson(s, N)
  r = O.solve(tr, "SLSQP", max_iter=2000, extra_options={"ftol": 1e-15}, cb=O.Callbacks(tr))
  K = 2 * N + 1
  return r['cost'], r['xs_and_us'][K - 1]


def test_oracle_simplecase_converges_to_the_bvp_optimum_at_fourth_order():
  from oracle import myriad_oracle as O
  g = GOLD["SIMPLECASE"]
  c10, _ = _oracle_cost(O.SimpleCase, 10)
  c20, xT = _oracle_cost(O.SimpleCase, 20)
  e10, e20 = c10 - g["cost"], c20 - g["cost"]
  assert abs(e20) < 2e-7 and abs(xT - g["x_T"]) < 1e-6
  assert 12.0 < e10 / e20 < 20.0            # h -> h/2: error / 16


def test_oracle_cancertreatment_approaches_the_bvp_optimum():
  from oracle import myriad_oracle as O
  g = GOLD["CANCERTREATMENT"]
  c20, _ = _oracle_cost(O.CancerTreatment, 20)
  c40, xT = _oracle_cost(O.CancerTreatment, 40)
  assert 0 < c40 - g["cost"] < 1e-4 and c40 - g["cost"] < 0.2 * (c20 - g["cost"])
  assert abs(xT - g["x_T"]) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name,N,tol_cost,tol_x", [("SIMPLECASE", 40, 2e-8, 1e-7), ("SIMPLECASE", 100, 2e-9, 1e-8),
                                                    ("CANCERTREATMENT", 100, 1e-5, 1e-5)])
def test_device_solve_matches_the_bvp_optimum(name, N, tol_cost, tol_x):
  from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON,
               intervals=N, nlpsolver=NLPSolverType.SQP)
  sol = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()).solve()
  g = GOLD[name]
  assert abs(sol['cost'] - g["cost"]) < tol_cost, (sol['cost'], g["cost"])
  assert abs(float(sol['x'][-1, 0]) - g["x_T"]) < tol_x
