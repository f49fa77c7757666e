#!/usr/bin/env python3
"""What the single-GPU kernels predict for the strong split of every BASELINE config (ONE batch partitioned over G GPUs): the measured
single-GPU rate at B / G instances, times G, for G in {1, 2, 4, 8} -- the expectation the first real SCALE run has to agree with.
No xGMI byte enters: the only collective is the final gather (8-33 MB, < 0.1 ms on any link); what bounds the strong split is that a launch of B / G
instances is one solve long, so the per-GPU rate at that size is what counts.  -> profiles/r06/projected_scaling.json
  usage (GPU box): python tools/projected_scaling.py out.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from bench_configs import CFG, timed
from myriad_amd.config import HParams, IntegrationMethod, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.systems.neural_ode import NeuralODE, NodeSystem
from myriad_amd.trajectory_optimizers import get_optimizer


def sweep(name, total, gs, make):
  rows = []
  for G in gs:
    B = total // G
    opt, kw = make(B)
    res, med, mn, ms = timed(opt, reps=5, **kw)
    rows.append(dict(gpus=G, per_gpu_batch=B, converged=float((res["status"] == 0).mean()), kernel_ms=ms, wall_ms=1e3 * med,
                     per_gpu_solves_per_s_kernel=B / ms * 1e3, per_gpu_solves_per_s_wall=B / med, its_max=int(res["iters"].max())))
    opt.engine.close()
  base = rows[0]["per_gpu_solves_per_s_kernel"]
  for r in rows:
    r["projected_job_solves_per_s"] = r["gpus"] * r["per_gpu_solves_per_s_kernel"]
    r["projected_parallel_efficiency"] = r["projected_job_solves_per_s"] / (r["gpus"] * base)
  return dict(config=name, total_batch=total, rows=rows)


def main(out):
  rng = np.random.default_rng(2019)
  res = {"what": __doc__.split("->")[0].strip()}
  def c2(B):
    hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=100, nlpsolver=NLPSolverType.SQP)
    return get_optimizer(hp, CFG, hp.system()), dict(x0s=np.clip(0.1 * np.random.default_rng(2019).standard_normal((B, 4)), -2, 2))
  res["config2"] = sweep("2 CARTPOLE HS N=100, 4096 random x0", 4096, (1, 2, 4, 8), c2)
  def c3(B):
    hp = HParams(system=SystemType.VANDERPOL, optimizer=OptimizerType.SHOOTING, intervals=1, controls_per_interval=50, nlpsolver=NLPSolverType.SQP)
    return get_optimizer(hp, CFG, hp.system()), dict(x0s=np.clip(np.array([0., 1.]) + 0.1 * np.random.default_rng(2019).standard_normal((B, 2)), -4, 4))
  res["config3"] = sweep("3 VANDERPOL shooting 1x50 Heun, 65536 random x0", 65536, (1, 2, 4, 8), c3)
  def c4(B):
    hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.SHOOTING, max_iter=500, nlpsolver=NLPSolverType.SQP)
    r = np.random.default_rng(2019)
    params = np.stack([r.uniform(0.1, 0.5, B), r.uniform(1, 5, B), r.uniform(0.2, 0.8, B)], axis=1)
    return get_optimizer(hp, CFG, hp.system()), dict(x0s=r.uniform(0.5, 0.99, (B, 1)), params=params)
  res["config4"] = sweep("4 CANCERTREATMENT shooting 1x100 Heun, 8192-instance parameter sweep", 8192, (1, 2, 4), c4)
  def c5(B):
    hp = HParams(system=SystemType.CARTPOLE, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, integration_method=IntegrationMethod.RK4, intervals=100, hidden_layers=(64, 64), nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, CFG, NodeSystem(NeuralODE.load_fitted_cartpole(), hp.system()))
    return opt, dict(x0s=np.clip(0.1 * np.random.default_rng(2019).standard_normal((B, 4)), -2, 2), params=opt.system.device_params())
  res["config5"] = sweep("5 CARTPOLE + NODE (64, 64) HS N=100, 1024 random x0", 1024, (1, 2, 4, 8), c5)
  json.dump(res, open(out, "w"), indent=1)
  for k in ("config2", "config3", "config4", "config5"):
    print(k, [(r["gpus"], r["per_gpu_batch"], round(r["projected_job_solves_per_s"]), round(r["projected_parallel_efficiency"], 3)) for r in res[k]["rows"]])


if __name__ == "__main__":
  main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/projected_scaling.json")
