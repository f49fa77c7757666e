# dev tool: README:83 literal config (CARTPOLE trapezoidal N=100) on the host twin: convergence / iteration statistics
import ctypes as C, os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import myriad_oracle as O
subprocess.run(["bash", os.path.join(ROOT, "tests", "hostsim", "build.sh")], check=True)
sim = C.CDLL(os.environ.get("HOSTSIM", os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")))
dp = C.c_void_p
sim.hostsim_solve_trap.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
A = lambda a: a.ctypes.data
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = 100
rng = np.random.default_rng(2019)
x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
s = O.CartPole()
zs, lbs, ubs = [], [], []
for b in range(B):
  s.x_0 = x0[b]
  tr = O.trapezoidal(s, N)
  zs.append(tr.guess); lbs.append(tr.bounds[:, 0]); ubs.append(tr.bounds[:, 1])
z = np.array(zs); lb = np.array(lbs); ub = np.array(ubs)
lam = np.zeros((B, N * 4)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
t0 = time.time()
sim.hostsim_solve_trap(0, N, s.T, B, A(z), A(lb), A(ub), None, 0, 1000, A(lam), A(cost), A(st), A(it), A(kkt))
print("time", time.time() - t0, "converged", (st == 0).mean(), "iters mean", it.mean(), "pct", np.percentile(it, [50, 90, 99, 100]), "status hist", np.bincount(st))
