import sys, os, faulthandler
sys.path.insert(0, os.getcwd())
import numpy as np
from myriad_amd.config import Config, HParams, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
for st in (SystemType.SIMPLECASE, SystemType.CANCERTREATMENT):
  for w in ("1", "2"):
    os.environ["MYRIAD_FUSED_WAVES"] = w
    hp = HParams(system=st, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.HERMITE_SIMPSON, intervals=50)
    print("solving", st.name, "W", w, flush=True)
    r = get_optimizer(hp, Config(verbose=False, plot=False), hp.system()).solve()
    print(st.name, w, r['cost'], flush=True)
