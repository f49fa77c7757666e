#!/usr/bin/env python3
"""One problem on REPS fresh handles (new scratch, whatever the previous kernel left in LDS) per environment set: distinct results?
   python tools/dev/fresh_stats.py SYSTEM RULE N B REPS "ENV1,ENV2,.."   (ENV = '+'-joined KEY=VALUE, '' = default)"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
s, r, N, B, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cfgs = [dict(kv.split("=", 1) for kv in c.split("+") if kv) for c in sys.argv[6].split(",")]
keys = sorted({k for c in cfgs for k in c})
max_iter = int(os.environ.get("WPROBE_MAX_ITER", "300"))
for cfg in cfgs:
  for k in keys:
    os.environ.pop(k, None)
  os.environ.update(cfg)
  seen = {}
  for _ in range(reps):
    hp = HParams(system=SystemType[s], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule["HERMITE_SIMPSON" if r == "HS" else "TRAPEZOIDAL"], intervals=N, nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    x0 = np.tile(opt.system.x_0, (B, 1)) * (1.0 + 0.01 * np.arange(B)[:, None])
    o = opt.solve_batch(x0s=x0, max_iter=max_iter)
    key = (tuple(o["status"]), tuple(o["iters"]), tuple(np.round(o["cost"], 9)), hashlib.sha1(np.ascontiguousarray(o["xs_and_us"]).tobytes()).hexdigest()[:8])
    seen[key] = seen.get(key, 0) + 1
    opt.engine.close()
  print(f"{s} {r} N={N} B={B} [{'+'.join(f'{k}={v}' for k, v in cfg.items()) or 'default'}]: {len(seen)} distinct result(s) on {reps} fresh handles")
  for k, c in sorted(seen.items(), key=lambda kv: -kv[1])[:5]:
    print(f"   x{c}: status {k[0]} iters {k[1]} cost {k[2]} z#{k[3]}")
