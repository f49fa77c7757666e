#!/bin/bash
# exp51 (round 5): the straggler solves of config 5 (75 iterations at seed 2019, 152 / 93 / 65 at seed 11 against a median of 24): per-iteration trace of each, solved alone
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp51
python - > gpurun_out/exp51/find.txt 2>&1 <<'PY'
import numpy as np, json
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
out = []
for seed in (2019, 11):
  x0 = np.clip(0.1 * np.random.default_rng(seed).standard_normal((1024, 4)), -2, 2)
  r = opt.solve_batch(x0s=x0, params=opt.system.device_params())
  idx = np.argsort(r["iters"])[::-1][:3]
  for i in idx: out.append(dict(seed=seed, i=int(i), iters=int(r["iters"][i]), cost=float(r["cost"][i]), x0=x0[i].tolist()))
  print("seed", seed, "cost percentiles", np.percentile(r["cost"], [0, 50, 99, 100]).tolist())
json.dump(out, open("gpurun_out/exp51/outliers.json", "w"))
print(out)
PY
MYRIAD_VARIANT_LIB=variants/libnodetrace.so python - > gpurun_out/exp51/trace.txt 2>&1 <<'PY'
import os, numpy as np, json
from myriad_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ["MYRIAD_VARIANT_LIB"])
from myriad_amd.config import Config, HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer
hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
opt = get_optimizer(hp, Config(verbose=False, plot=False), NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
for o in json.load(open("gpurun_out/exp51/outliers.json"))[:4]:
  print("=== seed", o["seed"], "instance", o["i"], "iters in batch", o["iters"], flush=True)
  r = opt.solve_batch(x0s=np.array([o["x0"]]), params=opt.system.device_params())
  print("=== alone: iters", r["iters"], "status", r["status"], "cost", r["cost"], flush=True)
PY
cat gpurun_out/exp51/find.txt | tail -5; grep -c . gpurun_out/exp51/trace.txt
