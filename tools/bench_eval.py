#!/usr/bin/env python3
"""Micro-benchmark of the hs_eval kernel (CARTPOLE HS N=100): device-resident inputs, HIP-event timing
from the library (myr_kernel_time), algorithmic bytes per SURVEY.md 8(d)."""
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
for wpt, nt in (("1","1"), ("4","0"), ("4","1"), ("8","1")):
  os.environ["MYRIAD_EVAL_WPT"] = wpt; os.environ["MYRIAD_EVAL_NT"] = nt
  eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, 2.0, max_batch=B)
  g = torch.Generator(device="cpu").manual_seed(0)
  z = torch.randn(B, eng.n, dtype=torch.float64, generator=g).cuda()
  f = torch.empty(B, dtype=torch.float64, device="cuda"); gr = torch.empty(B, eng.ngrad, dtype=torch.float64, device="cuda")
  c = torch.empty(B, eng.m, dtype=torch.float64, device="cuda"); j = torch.empty(B, eng.jblk, dtype=torch.float64, device="cuda")
  torch.cuda.synchronize()
  for _ in range(5):
    eng.eval_device(B, z, f=f, gradf=gr, c=c, jblk=j)
  eng.kernel_time_reset()
  for _ in range(a.iters):
    eng.eval_device(B, z, f=f, gradf=gr, c=c, jblk=j)
  ms, n = eng.kernel_time(_lib.K_EVAL)
  K = 2 * N + 1
  alg = 8 * (K * 5 + 16 + 2 * N * 4 + N * 100 + K * 1 + 1) * B
  print(json.dumps({"wpt": int(wpt), "nt": int(nt), "B": B, "N": N, "ms": ms, "launches": n, "alg_bytes": alg,
                    "GBps": alg / ms / 1e6, "frac_of_8TBps": alg / ms / 1e6 / 8000}))
  eng.close()
