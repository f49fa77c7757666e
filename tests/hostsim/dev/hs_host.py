# dev tool: one Hermite-Simpson solve on the host twin (lane form) with the reference's guess; env knobs of hostsim apply
import ctypes as C, os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import myriad_oracle as O
subprocess.run(["bash", os.path.join(ROOT, "tests", "hostsim", "build.sh")], check=True)
sim = C.CDLL(os.environ.get("HOSTSIM", os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")))
dp = C.c_void_p
sim.hostsim_solve.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp]
A = lambda a: a.ctypes.data
name, N = sys.argv[1], int(sys.argv[2])
SID = {"CARTPOLE": 0, "VANDERPOL": 1, "CANCERTREATMENT": 2, "SIMPLECASE": 3, "PENDULUM": 13, "EPIDEMICSEIRN": 10}
s = O.SYSTEMS[name]()
tr = O.hermite_simpson(s, N)
z = tr.guess.copy()[None]; lb = tr.bounds[:, 0].copy()[None]; ub = tr.bounds[:, 1].copy()[None]
m = 2 * N * s.x_0.shape[0]
lam = np.zeros((1, m)); cost = np.zeros(1); st = np.zeros(1, np.int32); it = np.zeros(1, np.int32); kkt = np.zeros((1, 3))
sim.hostsim_solve(SID[name], N, s.T, 1, A(z), A(lb), A(ub), None, 0, int(os.environ.get("MAXIT", "300")), 1e-8, 1e-6, 1e-7, 0.1, A(lam), A(cost), A(st), A(it), A(kkt))
print("status", st[0], "iters", it[0], "cost", cost[0], "kkt", kkt[0])
