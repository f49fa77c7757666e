# exp59 (round 5): do the REGISTERS a kernel inherits decide the result of the speculative rung + called sweep build?  (exp57: the result follows what the
# previous kernel left behind; exp58: not the 528 B of private memory.)  tools/dev/regfill leaves a pattern in every VGPR / AGPR of every SIMD before each solve.
import ctypes, hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"; os.environ["MYRIAD_FUSED_WAVES"] = os.environ.get("EXP59_WAVES", "2")
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
from myriad_amd import _lib
_lib.load()
rf = ctypes.CDLL(os.path.abspath("build/libregfill.so")); rf.regfill.argtypes = [ctypes.c_uint, ctypes.c_int]
def run(pat, blk, reps=8):
  seen = {}
  for _ in range(reps):
    hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=6, nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    if pat is not None: rf.regfill(pat, blk)
    o = opt.solve_batch(x0s=np.tile(opt.system.x_0, (1, 1)), max_iter=300)
    k = (int(o["status"][0]), int(o["iters"][0]), hashlib.sha1(np.ascontiguousarray(o["xs_and_us"]).tobytes()).hexdigest()[:8])
    seen[k] = seen.get(k, 0) + 1; opt.engine.close()
  return seen
print("no register fill:", run(None, -1), flush=True)
for name, pat in (("zeros", 0), ("0x7ff40000 (NaN high word)", 0x7ff40000), ("0x3ff00000 (1.0 high word)", 0x3ff00000), ("1", 1)):
  print(f"every register = {name}:", run(pat, -1), flush=True)
for b in range(16):
  print(f"registers = 0x7ff40000, {'v' if b < 8 else 'a'}[{32 * (b % 8)}..{32 * (b % 8) + 31}] = 0:", run(0x7ff40000, b, 6), flush=True)
