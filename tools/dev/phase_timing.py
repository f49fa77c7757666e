# dev tool: run bench workload once with the phase-timing build (prints cycles per phase for trajectories 0..3)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from myriad_amd import _lib
_lib.LIB_PATH = os.environ.get("MYRIAD_VARIANT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmyriad_hip_timing.so")
from bench import build_workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
x0, z0, lb, ub, T = build_workload(B, 100, 2019)
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 100, T, max_batch=B)
res = eng.solve(z0, lb, ub)
print("status", (res["status"] == 0).mean(), "kernel ms", eng.kernel_time(_lib.K_SOLVE))
