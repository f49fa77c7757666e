# dev tool: the shooting rows of the reference's smoke matrix (tests/test_smoke.py:14-61) on the device: status / attempts / start of every case,
# with second starts (the library default) and without -- the data behind tests/test_gpu_smoke.py::test_shooting_needs_no_elastic_phase
import os, sys, time, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd.config import Config, HParams, OptimizerType
from myriad_amd.systems import SystemType
from myriad_amd.trajectory_optimizers import get_optimizer
APPROACHES = {"single_shooting": dict(intervals=1, controls_per_interval=90), "multiple_shooting_3_controls": dict(intervals=30, controls_per_interval=3),
              "multiple_shooting_1_control": dict(intervals=90, controls_per_interval=1)}
out = {}
for st in SystemType:
  if st.name == "INVASIVEPLANT": continue
  for ap, kw in APPROACHES.items():
    row = {}
    for ss in ("default", "0"):
      if ss == "0": os.environ["MYRIAD_SECOND_STARTS"] = "0"
      else: os.environ.pop("MYRIAD_SECOND_STARTS", None)
      hp = HParams(system=st, optimizer=OptimizerType.SHOOTING, **kw)
      if st.name == "ROCKETLANDING": hp.max_iter = 150
      t0 = time.time()
      try:
        opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
        res = opt.solve_batch(x0s=np.asarray(opt.system.x_0, dtype=np.float64)[None])
        row[ss] = dict(status=int(np.asarray(res["status"]).ravel()[0]), attempts=int(np.asarray(res.get("attempts", 1)).ravel()[0]), start=int(np.asarray(res.get("start", 0)).ravel()[0]),
                       cost=float(np.asarray(res["cost"]).ravel()[0]), s=round(time.time() - t0, 2))
      except Exception as e:
        row[ss] = dict(error=type(e).__name__ + ": " + str(e)[:80])
    out[f"{st.name}/{ap}"] = row
    print(st.name, ap, json.dumps(row), flush=True)
