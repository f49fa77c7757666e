# dev tool: the elastic phase on PENDULUM / ROCKETLANDING (collocation), second starts off
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["MYRIAD_SECOND_STARTS"] = "0"
import numpy as np
from myriad_amd import _lib
from myriad_amd.config import Config, HParams, IntegrationMethod, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
for name, N, rule in (("PENDULUM", 20, "HERMITE_SIMPSON"), ("PENDULUM", 50, "HERMITE_SIMPSON"), ("PENDULUM", 100, "HERMITE_SIMPSON"), ("PENDULUM", 50, "TRAPEZOIDAL"),
                      ("ROCKETLANDING", 20, "HERMITE_SIMPSON"), ("ROCKETLANDING", 50, "HERMITE_SIMPSON"), ("ROCKETLANDING", 50, "TRAPEZOIDAL")):
  hp = HParams(system=SystemType[name], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule],
               integration_method=IntegrationMethod.HEUN, intervals=N, max_iter=3000)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  rhos = tuple(float(a) for a in sys.argv[1:]) or opt.elastic_rhos
  opt.elastic_rhos = rhos
  x0s = np.tile(opt.system.x_0, (1, 1))
  z0, lb, ub = opt.batch_inputs(x0s, opt.system.device_params())
  o = opt.engine.default_opts(); o.max_iter = 3000
  t0 = time.time()
  r1 = opt._solve_sharded(z0, lb, ub, opt.system.device_params(), o)
  print(f"{name} N={N} {rule}: first attempt status {r1['status']} iters {r1['iters']} cost {r1['cost']} kkt {r1['kkt']}", flush=True)
  r2 = opt.elastic_restoration(z0, lb, ub, opt.system.device_params(), o)
  print(f"   elastic: twin_status {r2['twin_status']} slack {r2['slack']} -> status {r2['status']} iters {r2['iters']} cost {r2['cost']} kkt {r2['kkt']}  {time.time()-t0:.1f}s", flush=True)
