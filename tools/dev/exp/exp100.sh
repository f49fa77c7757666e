#!/bin/bash
# exp100: longer horizons of the headline system -- at N = 150 / 200 the solver's LDS lets three / two one-wavefront workgroups onto a CU: does the rule of exp96
# (two wavefronts per trajectory at every batch size when as many two-wavefront workgroups fit) carry over to the systems with the two-level sweep?
cd /root/repo; O=gpurun_out/exp100; mkdir -p $O
python - <<'PY' | tee $O/times.txt
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from myriad_amd import _lib
from bench import build_workload
for N in (100, 150, 200, 300):
  for B in (4096, 1024):
    x0, z0, lb, ub, T = build_workload(B, N, 2019)
    for w in ("1", "2"):
      os.environ["MYRIAD_FUSED_WAVES"] = w
      eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
      ts = []
      for _ in range(3):
        eng.kernel_time_reset(); r = eng.solve(z0, lb, ub); ms, n = eng.kernel_time(_lib.K_SOLVE); ts.append(ms)
      print(f"N={N} B={B} W={w}: {min(ts):.2f} ms in {n} launches, converged {(r['status']==0).mean():.3f}, plan {eng.solve_plan()}", flush=True)
      eng.close()
PY
