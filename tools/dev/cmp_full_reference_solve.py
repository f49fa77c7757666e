import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
F = np.load("/root/repo/tests/golden/reference_solve_full.npz"); key = "solve/CARTPOLE/COLLOCATION/HERMITE_SIMPSON/100x1"
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=100, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
z_ref, c_ref = F[key + "/xs_and_us"], float(F[key + "/cost"])
eng = opt.engine; o = eng.default_opts(); o.max_iter = hp.max_iter; o.restoration = 0
z0, lb, ub = opt.batch_inputs(np.asarray(opt.system.x_0, dtype=np.float64)[None], opt.system.device_params())
r = opt.device_solve(z0, lb, ub, opt.system.device_params(), o, second_starts=False)
z = r["z"][0]
print("device from the reference's guess: status", r["status"][0], "iterations", r["iters"][0], "cost %.12f" % r["cost"][0], "reference %.12f" % c_ref, "difference %.3e" % (r["cost"][0] - c_ref),
      "max|c| %.2e" % np.abs(opt.constraints(z)).max(), "max|z - z_ref| %.3e" % np.abs(z - z_ref).max(), "states %.3e" % np.abs(z[:804] - z_ref[:804]).max())
# ... and the instances of the headline batch / README:83's literal (tests/golden/reference_solve_draws.npz)
D = np.load("/root/repo/tests/golden/reference_solve_draws.npz")
for rule in ("HERMITE_SIMPSON", "TRAPEZOIDAL"):
  keys = sorted(k.rsplit("/", 1)[0] for k in D.files if k.startswith(f"draw/{rule}/") and k.endswith("/cost"))
  hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=100, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  o = opt.engine.default_opts(); o.max_iter = hp.max_iter; o.restoration = 0
  z0, lb, ub = opt.batch_inputs(np.stack([D[k + "/x0"] for k in keys]), opt.system.device_params())
  r = opt.device_solve(z0, lb, ub, opt.system.device_params(), o, second_starts=False)
  for b, k in enumerate(keys):
    zr = D[k + "/xs_and_us"]; nx = (opt.engine.n // 5) * 4
    print(k, "status", r["status"][b], "iterations", r["iters"][b], "cost %.10f" % r["cost"][b], "reference %.10f" % float(D[k + "/cost"]), "difference %.2e" % (r["cost"][b] - float(D[k + "/cost"])),
          "feasibility %.1e" % r["kkt"][b, 0], "states %.2e" % np.abs(r["z"][b][:nx] - zr[:nx]).max(), "all %.2e" % np.abs(r["z"][b] - zr).max())
# ... and rows 0..3 of config 4's parameter sweep (tests/golden/reference_solve_sweep.npz)
from myriad_amd.config import IntegrationMethod
S = np.load("/root/repo/tests/golden/reference_solve_sweep.npz")
keys = sorted({k.rsplit("/", 1)[0] for k in S.files})
hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.SHOOTING, max_iter=500, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
r = opt.solve_batch(x0s=np.stack([S[k + "/x0"] for k in keys]), params=np.stack([S[k + "/params"] for k in keys]))
for b, k in enumerate(keys):
  print(k, "status", r["status"][b], "iterations", r["iters"][b], "cost %.10f" % r["cost"][b], "reference %.10f" % float(S[k + "/cost"]), "difference %.2e" % (r["cost"][b] - float(S[k + "/cost"])),
        "all %.2e" % np.abs(r["xs_and_us"][b] - S[k + "/xs_and_us"]).max())
