#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (round 6, late): the reference's own solve() on instances of the HEADLINE WORKLOAD -- CARTPOLE, Hermite-Simpson, N = 100, start states drawn as
bench.py draws them (x0 = clip(x_0 + 0.1 xi), default_rng(2019), rows 0..10 of the batch) -- and on README.md:83's literal (trapezoidal, N = 100, the system's own start state).
Same arrangement as make_reference_fixtures.py: /root/reference read in place through the JAX stand-in (tests/golden/refshim), SLSQP with complex-step derivatives of the
reference's callbacks, ~45 minutes per Hermite-Simpson instance on one core; the cases are independent processes:
    for i in 0 1 2 3; do PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_full_draws.py $i & done; wait
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_reference_full_draws.py merge      ->  tests/golden/reference_solve_draws.npz
Fixtures are data: start state, the reference's end point, its cost."""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402

CASES = [("HERMITE_SIMPSON", 0), ("HERMITE_SIMPSON", 1), ("HERMITE_SIMPSON", 2), ("TRAPEZOIDAL", None)] + [("HERMITE_SIMPSON", r) for r in range(3, 11)]      # (rows 3..10: a second run, merged in)


def draws():
  """rows 0..2 of bench.build_workload(B, 100, 2019): the same generator, the same clip (bench.py:41-47)"""
  x_0 = np.array([0., 0., 0., 0.]); lo = np.array([-2., -np.pi, -np.inf, -np.inf]); hi = -lo      # CartPole: x_0 and the state box (cartpole.py:48-59)
  rng = np.random.default_rng(2019)
  return np.clip(x_0[None] + 0.1 * rng.standard_normal((11, 4)), lo, hi)


def main():
  if sys.argv[1] == "merge":
    out = {}
    have = os.path.join(HERE, "reference_solve_draws.npz")
    if os.path.exists(have):      # (cases of an earlier run stay)
      d = np.load(have); out.update({k: d[k] for k in d.files})
    for i in range(len(CASES)):
      f = os.path.join(HERE, f"reference_solve_draw{i}.npz")
      if not os.path.exists(f): continue
      d = np.load(f)
      out.update({k: d[k] for k in d.files})
    np.savez_compressed(os.path.join(HERE, "reference_solve_draws.npz"), **out)
    print(sorted(out))
    return
  i = int(sys.argv[1])
  rule, row = CASES[i]
  import refshim
  refshim.install()
  sys.path.insert(0, "/root/reference")
  from myriad.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
  from myriad.systems import SystemType
  from myriad.trajectory_optimizers import get_optimizer
  hp = HParams(system=SystemType.CARTPOLE, nlpsolver=NLPSolverType.SLSQP, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=100)
  system = hp.system()
  if row is not None:
    x0 = draws()[row]
    import jax.numpy as jnp
    system.x_0 = jnp.array(x0)
  opt = get_optimizer(hp, Config(verbose=False, plot=False, jit=False), system)
  res = opt.solve()
  key = f"draw/{rule}/{'x_0' if row is None else row}"
  z = np.real(np.asarray(res["xs_and_us"], dtype=np.float64))
  out = {key + "/x0": np.asarray(np.real(np.asarray(system.x_0)), dtype=np.float64), key + "/xs_and_us": z, key + "/cost": np.array(float(np.real(res["cost"]))),
         key + "/max_abs_c": np.array(float(np.abs(np.asarray(opt.constraints(np.asarray(res["xs_and_us"])), dtype=np.float64)).max()))}
  np.savez_compressed(os.path.join(HERE, f"reference_solve_draw{i}.npz"), **out)
  print(key, "cost", float(out[key + "/cost"]), "max|c|", float(out[key + "/max_abs_c"]), "x0", out[key + "/x0"])


if __name__ == "__main__":
  main()
