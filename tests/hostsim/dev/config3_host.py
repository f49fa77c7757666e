# dev tool: config 3 (VANDERPOL single shooting 1x50, Heun) on the host twin of the solver: find the stragglers
import ctypes as C, os, sys, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import myriad_oracle as O
lib = os.environ.get("HOSTSIM", os.path.join(ROOT, "tests", "hostsim", "libhostsim.so"))
subprocess.run(["bash", os.path.join(ROOT, "tests", "hostsim", "build.sh")], check=True)
sim = C.CDLL(lib)
dp = C.c_void_p
sim.hostsim_solve_shoot.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
A = lambda a: a.ctypes.data
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
only = [int(v) for v in sys.argv[2:]]
rng = np.random.default_rng(2019)
x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((8192, 2)), -4, 4)[:B]
s = O.VanDerPol()
tr = O.shooting(s, 1, 50, "HEUN")
n = tr.guess.size
z = np.tile(tr.guess, (B, 1)); lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
z[:, 0:2] = x0; lb[:, 0:2] = x0; ub[:, 0:2] = x0
if only:
  z, lb, ub = z[only].copy(), lb[only].copy(), ub[only].copy(); B = len(only)
m = 2
lam = np.zeros((B, m)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
t0 = time.time()
sim.hostsim_solve_shoot(1, 1, 50, 1, s.T, B, A(z), A(lb), A(ub), None, 0, int(os.environ.get("MAXIT", "1000")), A(lam), A(cost), A(st), A(it), A(kkt))
print("time", time.time() - t0, "converged", (st == 0).mean(), "iters pct", np.percentile(it, [50, 90, 99, 99.9, 100]))
bad = np.nonzero(st != 0)[0]
print("bad", bad[:40], "status", st[bad][:40], "iters", it[bad][:40])
slow = np.argsort(-it)[:20]
print("slowest", slow, it[slow])
if only:
  print("x0", x0[only] if len(only) else None, "cost", cost, "kkt", kkt)
if os.environ.get("SAVE"):
  np.savez(os.environ["SAVE"], cost=cost, st=st, it=it, z=z)
