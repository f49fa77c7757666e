#!/bin/bash
# exp27: zero- against pattern-initialised ld0 (backward pass, HsFused): how do the iterates differ?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp27
for t in z p; do
MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/libt_${t}_291.so MYRIAD_SECOND_STARTS=0 MYRIAD_ELASTIC=0 python - <<PY
import numpy as np
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
out = {}
for it in (0, 1, 2, 3, 5, 32):
  hp = HParams(system=SystemType.TUMOUR, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=6, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  o = opt.solve_batch(x0s=np.tile(opt.system.x_0, (1, 1)), max_iter=it)
  out["z%d" % it] = o["xs_and_us"]; out["lam%d" % it] = o.get("lam", np.zeros(1)); out["kkt%d" % it] = o["kkt"]; out["cost%d" % it] = o["cost"]
np.savez("gpurun_out/exp27/$t.npz", **out)
PY
done
python - <<'PY'
import numpy as np
a = np.load("gpurun_out/exp27/z.npz"); b = np.load("gpurun_out/exp27/p.npz")
for k in a.files:
  x, y = a[k], b[k]
  if x.shape != y.shape: print(k, "shape", x.shape, y.shape); continue
  d = np.abs(x - y); 
  print(k, "max abs diff %.3e" % d.max(initial=0.0), "rel %.3e" % (d / np.maximum(1e-300, np.abs(x))).max(initial=0.0), "nan", np.isnan(x).sum(), np.isnan(y).sum())
PY
