# exp63 (round 5): the register / stack fill gate of tests/test_gpu_poison.py over MORE sizes than the suite runs (N = 3 .. 100, B = 1 / 3 / 70): every system x scheme x kernel form
# must return identical bits under MYRIAD_REG_FILL / MYRIAD_STACK_FILL = zero and nan.  A sweep for profiles/r05, not a test (20 systems x 2 schemes x 9 sizes x 4 forms x 2 fills).
import hashlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
FORMS = [{"MYRIAD_FUSED_WAVES": "1"}, {"MYRIAD_FUSED_WAVES": "2"}, {"MYRIAD_SOLVE_MODE": "wave1"}, {"MYRIAD_SOLVE_MODE": "lane"}]
def solve(system, rule, N, B, env):
  for k in ("MYRIAD_FUSED_WAVES", "MYRIAD_SOLVE_MODE", "MYRIAD_REG_FILL", "MYRIAD_STACK_FILL"): os.environ.pop(k, None)
  os.environ.update(env)
  hp = HParams(system=SystemType[system], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule[rule], intervals=N, nlpsolver=NLPSolverType.SQP)
  opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
  x0 = np.tile(opt.system.x_0, (B, 1)) * (1.0 + 0.01 * np.arange(B)[:, None])
  o = opt.solve_batch(x0s=x0, max_iter=40)
  h = hashlib.sha1(b"".join(np.ascontiguousarray(o[k]).tobytes() for k in ("xs_and_us", "lambda", "cost", "status", "iters"))).hexdigest()[:10]
  opt.engine.close()
  return h
bad = []; n = 0
for st in SystemType:
  if st.name in ("INVASIVEPLANT", "PREDATORPREY"): continue
  for rule in ("HERMITE_SIMPSON", "TRAPEZOIDAL"):
    for N, B in ((3, 1), (4, 3), (5, 3), (8, 1), (10, 3), (13, 3), (33, 3), (70, 3), (100, 70)):
      for form in FORMS:
        if form.get("MYRIAD_SOLVE_MODE") == "lane" and B > 3: continue
        a = solve(st.name, rule, N, B, dict(form, MYRIAD_REG_FILL="zero", MYRIAD_STACK_FILL="zero"))
        b = solve(st.name, rule, N, B, dict(form, MYRIAD_REG_FILL="nan", MYRIAD_STACK_FILL="nan"))
        n += 1
        if a != b: bad.append((st.name, rule, N, B, form)); print("DIFFERENT:", st.name, rule, N, B, form, flush=True)
  print(st.name, "done;", n, "cases,", len(bad), "differing", flush=True)
print(json.dumps({"cases": n, "differing": bad}))
