#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (round 6, late): the reference's own solve() on instances of BASELINE config 4 -- CANCERTREATMENT, single shooting 1 x 100 (Heun), a sweep over the
system's parameters (r, a, delta) and start states: rows 0..3 of the batch tools/bench_configs.py draws (default_rng(2019): config 3's 8192 x 2 normals first, then
r ~ U(0.1, 0.5), a ~ U(1, 5), delta ~ U(0.2, 0.8), x0 ~ U(0.5, 0.99), 2048 each).  Same arrangement as make_reference_fixtures.py (JAX stand-in tests/golden/refshim, SLSQP with
complex-step derivatives of the reference's callbacks), independent processes:
    for i in 0 1 2 3; do PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_sweep.py $i & done; wait
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_sweep.py merge      ->  tests/golden/reference_solve_sweep.npz
Fixtures are data: the instance (r, a, delta, x0), the reference's end point, its cost."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402

NROWS = 4


def rows():
  """(params [2048][3], x0 [2048][1]) of tools/bench_configs.py: measure(), config 4 -- the same generator calls in the same order"""
  rng = np.random.default_rng(2019)
  rng.standard_normal((8192, 2))                     # config 3's start states
  B = 2048
  params = np.stack([rng.uniform(0.1, 0.5, B), rng.uniform(1, 5, B), rng.uniform(0.2, 0.8, B)], axis=1)
  x0 = rng.uniform(0.5, 0.99, (B, 1))
  return params, x0


def main():
  if sys.argv[1] == "merge":
    out = {}
    for i in range(NROWS):
      d = np.load(os.path.join(HERE, f"reference_solve_sweep{i}.npz")); out.update({k: d[k] for k in d.files})
    np.savez_compressed(os.path.join(HERE, "reference_solve_sweep.npz"), **out)
    print(sorted(out))
    return
  i = int(sys.argv[1])
  params, x0 = rows()
  import refshim
  refshim.install()
  sys.path.insert(0, "/root/reference")
  from myriad.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType
  from myriad.systems import SystemType
  from myriad.systems.lenhart.cancer_treatment import CancerTreatment
  from myriad.trajectory_optimizers import get_optimizer
  hp = HParams(system=SystemType.CANCERTREATMENT, nlpsolver=NLPSolverType.SLSQP, optimizer=OptimizerType.SHOOTING, integration_method=IntegrationMethod.HEUN, intervals=1,
               controls_per_interval=100, max_iter=500)
  system = CancerTreatment(r=float(params[i, 0]), a=float(params[i, 1]), delta=float(params[i, 2]), x_0=float(x0[i, 0]))
  opt = get_optimizer(hp, Config(verbose=False, plot=False, jit=False), system)
  res = opt.solve()
  key = f"sweep/{i}"
  out = {key + "/params": params[i], key + "/x0": x0[i], key + "/xs_and_us": np.real(np.asarray(res["xs_and_us"], dtype=np.float64)), key + "/cost": np.array(float(np.real(res["cost"])))}
  np.savez_compressed(os.path.join(HERE, f"reference_solve_sweep{i}.npz"), **out)
  print(key, "params", params[i], "x0", x0[i], "cost", float(out[key + "/cost"]))


if __name__ == "__main__":
  main()
