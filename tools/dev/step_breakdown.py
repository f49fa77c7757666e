# dev tool: host-side wall-clock breakdown of one bench step
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from myriad_amd import _lib
from bench import build_workload
B, N = 4096, 100
x0, z0h, lbh, ubh, T = build_workload(B, N, 2019)
f64 = dict(dtype=torch.float64, device="cuda")
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, max_batch=B)
z0 = torch.tensor(z0h, **f64); lb = torch.tensor(lbh, **f64); ub = torch.tensor(ubh, **f64); z = z0.clone()
lam = torch.empty(B, eng.m, **f64); cost = torch.empty(B, **f64); kkt = torch.empty(B, 3, **f64)
status = torch.empty(B, dtype=torch.int32, device="cuda"); iters = torch.empty(B, dtype=torch.int32, device="cuda")
fv = torch.empty(B, **f64); gv = torch.empty(B, eng.ngrad, **f64); cv = torch.empty(B, eng.m, **f64); jv = torch.empty(B, eng.jblk, **f64)
opts = eng.default_opts(); opts.max_iter = 1000
for it in range(6):
  t = [time.perf_counter()]
  z.copy_(z0); torch.cuda.current_stream().synchronize(); t.append(time.perf_counter())
  eng.solve_device(B, z, lb, ub, None, 0, opts, lam, cost, status, iters, kkt); t.append(time.perf_counter())
  eng.eval_device(B, z, f=fv, gradf=gv, c=cv, jblk=jv); t.append(time.perf_counter())
  ok = (status == 0) & (cv.abs().amax(dim=1) <= 1e-8); n = int(ok.sum().item()); t.append(time.perf_counter())
  print("copy %.2f solve %.2f eval %.2f check %.2f ms | kernel %.2f" % tuple([1e3 * (t[i + 1] - t[i]) for i in range(4)] + [eng.kernel_time(_lib.K_SOLVE)[0]]))
