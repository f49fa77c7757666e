#!/bin/bash
# exp44: two-phase first attempt + restoration inside myr_solve: PENDULUM (needs the elastic phase), B = 2500, against whole solves
cd $GRAFT_REPO_ROOT
for k in 0 -1; do
MYRIAD_PARK_ITER=$k python - <<'PY'
import hashlib, os, time, numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.PENDULUM, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=20, nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
B = 2500
rng = np.random.default_rng(3)
x0 = np.tile(opt.system.x_0, (B, 1)) + 0.05 * rng.standard_normal((B, opt.system.x_0.size))
t0 = time.time(); o = opt.solve_batch(x0s=x0, max_iter=300); dt = time.time() - t0
bits = hashlib.sha1(b"".join(np.ascontiguousarray(o[k]).tobytes() for k in ("xs_and_us", "cost", "status", "iters"))).hexdigest()[:12]
print("PARK_ITER", os.environ["MYRIAD_PARK_ITER"], "status", np.bincount(o["status"]).tolist(), "restored", int(np.sum(o.get("restored", 0))), "attempts max", int(np.max(o.get("attempts", 1))), "iters median", np.median(o["iters"]), "bits", bits, "%.2f s" % dt)
PY
done
