#!/usr/bin/env python3
"""Micro-benchmark of the matrix-free product kernels (CARTPOLE HS N=100): device-resident inputs, HIP-event timing
from the library (myr_kernel_time).  Algorithmic bytes per instance: J^T lam / grad L reads z (n) + lam (m) and writes
n doubles; J v reads z (n) + v (n) and writes m doubles; one extragradient step would be 3 such passes if it went
through HBM -- the fused kernel keeps the iterate in LDS, so its figure of merit is steps/s."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from myriad_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--intervals", type=int, default=100)
ap.add_argument("--iters", type=int, default=50)
a = ap.parse_args()
B, N = a.batch, a.intervals
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, 2.0, max_batch=B)
g = torch.Generator(device="cpu").manual_seed(0)
z = (0.3 * torch.randn(B, eng.n, dtype=torch.float64, generator=g)).cuda()
lam = torch.randn(B, eng.m, dtype=torch.float64, generator=g).cuda()
v = torch.randn(B, eng.n, dtype=torch.float64, generator=g).cuda()
on = torch.empty(B, eng.n, dtype=torch.float64, device="cuda"); om = torch.empty(B, eng.m, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for op, w, out, alg in (("vjp", lam, on, 8 * (2 * eng.n + eng.m) * B), ("jvp", v, om, 8 * (2 * eng.n + eng.m) * B)):
  for _ in range(5):
    eng.products_device(op, B, z, w, out, add_gradf=1)
  eng.kernel_time_reset()
  for _ in range(a.iters):
    eng.products_device(op, B, z, w, out, add_gradf=1)
  ms, n = eng.kernel_time(_lib.K_PROD)
  print(json.dumps({"kernel": "colloc_%s_kernel<CARTPOLE,HS>" % op, "B": B, "N": N, "ms": ms, "launches": n, "alg_bytes": alg,
                    "GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000}))
# fused extragradient: steps per second per instance batch
import ctypes as C
lb = torch.full((B, eng.n), -1e30, dtype=torch.float64, device="cuda"); ub = -lb
zz = z.clone(); ll = torch.ones(B, eng.m, dtype=torch.float64, device="cuda")
steps = 200
eng.kernel_time_reset()
_lib._chk(eng.lib.myr_exgd(eng._h, B, _lib._addr(zz), _lib._addr(ll), _lib._addr(lb), _lib._addr(ub), None, 0, 1e-3, 1e-5, steps, _lib.MEM_DEVICE), "myr_exgd")
ms, n = eng.kernel_time(_lib.K_PROD)
print(json.dumps({"kernel": "colloc_exgd_kernel<CARTPOLE,HS>", "B": B, "N": N, "steps": steps, "ms": ms,
                  "instance_steps_per_s": B * steps / ms * 1e3, "us_per_step": ms * 1e3 / steps}))
eng.close()
