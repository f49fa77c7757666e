# exp58 (round 5): WHICH private-memory dword decides the result of the speculative rung + called sweep build (exp57: the result follows what the stack inherits)
#   every dword NaN-patterned except a window of zeros; 6 fresh handles per window; the window whose zeros bring the "zero-fill" result back holds the slot
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"; os.environ["MYRIAD_FUSED_WAVES"] = "2"
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
def run(reps=6):
  seen = {}
  for _ in range(reps):
    hp = HParams(system=SystemType.CANCERTREATMENT, optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule.TRAPEZOIDAL, intervals=6, nlpsolver=NLPSolverType.SQP)
    opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
    o = opt.solve_batch(x0s=np.tile(opt.system.x_0, (1, 1)), max_iter=300)
    k = (int(o["status"][0]), int(o["iters"][0]), hashlib.sha1(np.ascontiguousarray(o["xs_and_us"]).tobytes()).hexdigest()[:8])
    seen[k] = seen.get(k, 0) + 1; opt.engine.close()
  return seen
step = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for z0 in range(0, 132, step):
  os.environ["MYRIAD_STACK_FILL_WINDOW"] = f"528:{z0}:{z0 + step}"
  print(f"zeros in dwords [{z0},{z0 + step}) (bytes {4 * z0}..{4 * (z0 + step)}):", run(), flush=True)
os.environ["MYRIAD_STACK_FILL_WINDOW"] = "528:0:0"; print("no zeros:", run(), flush=True)
os.environ["MYRIAD_STACK_FILL_WINDOW"] = "528:0:132"; print("all zeros:", run(), flush=True)
