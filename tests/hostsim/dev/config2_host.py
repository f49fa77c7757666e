# dev tool: the headline workload (CARTPOLE HS N=100) on the host twin of the solver (lane form): iteration statistics
import ctypes as C, os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from bench import build_workload
subprocess.run(["bash", os.path.join(ROOT, "tests", "hostsim", "build.sh")], check=True)
sim = C.CDLL(os.environ.get("HOSTSIM", os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")))
dp = C.c_void_p
sim.hostsim_solve.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp]
A = lambda a: a.ctypes.data
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
x0, z, lb, ub, T = build_workload(B, int(os.environ.get("NINT", "100")), int(os.environ.get("SEED", "2019")))
z = z.copy()
if os.environ.get('ONLY'):
  i = int(os.environ['ONLY']); z, lb, ub = z[i:i+1].copy(), lb[i:i+1].copy(), ub[i:i+1].copy(); B = 1
lam = np.zeros((B, 8 * int(os.environ.get("NINT", "100")))); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
t0 = time.time()
sim.hostsim_solve(0, int(os.environ.get("NINT", "100")), T, B, A(z), A(lb), A(ub), None, 0, 1000, 1e-8, 1e-6, 1e-7, 0.1, A(lam), A(cost), A(st), A(it), A(kkt))
print("time", time.time() - t0, "converged", (st == 0).mean(), "iters mean", it.mean(), "pct", np.percentile(it, [50, 90, 99, 100]), "cost mean", cost.mean())
if os.environ.get("SAVE"):
  np.savez(os.environ["SAVE"], cost=cost, st=st, it=it)
if os.environ.get("SWEEPS"):
  print("sweeps mean", kkt[:, 2].mean(), "per iteration", kkt[:, 2].sum() / it.sum())
slow = np.argsort(-it)[:12]
print("slowest", slow.tolist(), it[slow].tolist())
