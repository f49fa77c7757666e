#!/usr/bin/env python3
"""bench.py -- headline benchmark: converged trajectory-optimisation solves per second (batched),
CARTPOLE Hermite-Simpson collocation, 100 intervals, B = 4096 random x0 per GPU (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch:
   z0 -> [myr_solve: batched SQP on device] -> z*, lambda*, cost, status
      -> [myr_eval : hs_eval kernel on z*]   -> c(z*), J blocks, grad f   (independent convergence verification)
      -> (N > 1) RCCL all_gather of z*, cost, status to every rank (the path's only collective).
Inputs are resident in HBM before the timed region.  Multi-GPU: independent instances are sharded across
ranks (weak scaling: 4096 per GPU), launched by torch.distributed.run, one process per GPU.

Prints ONE JSON line on rank 0 (see the task contract), including
  roofline     -- the hs_eval kernel (the defect+Jacobian kernel of SURVEY.md 8(d)): algorithmic bytes per launch /
                  HIP-event duration measured on the library's own stream inside the timed region;
  cpu_baseline -- the oracle's SciPy SLSQP path (the reference's NLPSolverType.SLSQP branch on restated callbacks)
                  timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_EVAL = lambda N, ns, nu: 8 * ((2 * N + 1) * (ns + nu) + 16 + 2 * N * ns + N * (5 * ns * ns + 5 * ns * nu) + (2 * N + 1) * nu + 1)
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def build_workload(B, N, seed):
  """x0_b = clip(x_0 + 0.1 xi_b), guess = linspace(x0, x_T) / u = 0 (hermite_simpson.py:37-43),
  bounds per hermite_simpson.py:55-81 with row 0 pinned to x0_b.  Pure numpy (no oracle on the product path)."""
  from myriad_amd.systems import CartPole
  s = CartPole()
  rng = np.random.default_rng(seed)
  x0 = np.clip(s.x_0[None] + 0.1 * rng.standard_normal((B, 4)), s.bounds[:4, 0], s.bounds[:4, 1])
  K = 2 * N + 1
  lin = np.linspace(0.0, 1.0, K)[None, :, None]
  xs = x0[:, None, :] * (1 - lin) + s.x_T[None, None, :] * lin
  z0 = np.concatenate([xs.reshape(B, -1), np.zeros((B, K))], axis=1)
  lb = np.empty((B, K * 5)); ub = np.empty((B, K * 5))
  lb[:, :K * 4] = np.tile(s.bounds[:4, 0], K); ub[:, :K * 4] = np.tile(s.bounds[:4, 1], K)
  lb[:, K * 4:] = s.bounds[4, 0]; ub[:, K * 4:] = s.bounds[4, 1]
  lb[:, :4] = x0; ub[:, :4] = x0
  lb[:, (K - 1) * 4:K * 4] = s.x_T; ub[:, (K - 1) * 4:K * 4] = s.x_T
  return x0, z0, lb, ub, s.T


def cpu_baseline(N, budget_s):
  """Oracle (restated reference transcription + SciPy SLSQP = the reference's NLPSolverType.SLSQP path) on host cores.
  A full N=100 solve takes minutes, so the sample is time-bounded: SLSQP runs on instance 0 until `budget_s` of wall
  time is used (the objective callback aborts it), and the measured iteration rate is scaled by the 110 iterations
  the same solve needs to converge at the reference's default tolerance."""
  from oracle import myriad_oracle as O
  import torch
  cores = os.cpu_count() or 1
  s = O.CartPole()
  tr = O.hermite_simpson(s, N)
  cb = O.Callbacks(tr)
  cb.jac(tr.guess); cb.grad(tr.guess)        # warm up autodiff

  class _Timeout(Exception):
    pass

  t0 = time.time()
  njac = [0]
  jac0 = cb.jac

  def timed_jac(z):                          # one Jacobian evaluation per SLSQP major iteration
    if time.time() - t0 > budget_s:
      raise _Timeout()
    njac[0] += 1
    return jac0(z)

  cb.jac = timed_jac
  try:
    O.solve(tr, "SLSQP", max_iter=1000, cb=cb)
  except _Timeout:
    pass
  dt = time.time() - t0
  nit = max(1, njac[0] - 1)
  its_per_s = nit / dt
  full_its = 110
  return {"value": its_per_s / full_its, "unit": "solves/s", "cores": int(torch.get_num_threads()), "host_cores": cores,
          "kind": "port",
          "sample": f"oracle SciPy-SLSQP path, CARTPOLE HS N={N} instance 0 (default x0): {nit} SLSQP iterations in {dt:.1f} s "
                    f"({its_per_s:.3f} it/s); a converged solve needs {full_its} iterations at the reference's default "
                    f"tolerance (BASELINE.md: 546 s measured), so solves/s = it/s / {full_its}"}


def cpu_same_algorithm(z0, lb, ub, N, T, nsample):
  """Informative extra: the SAME SQP core compiled for the host (tests/hostsim, test infrastructure) on all host cores."""
  import ctypes as C
  path = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
  if not os.path.exists(path):
    return None
  lib = C.CDLL(path)
  dp = C.c_void_p
  lib.hostsim_solve.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, C.c_double,
                                C.c_double, C.c_double, C.c_double, dp, dp, dp, dp, dp]
  B = min(nsample, z0.shape[0])
  z = np.ascontiguousarray(z0[:B]).copy(); l = np.ascontiguousarray(lb[:B]); u = np.ascontiguousarray(ub[:B])
  lam = np.zeros((B, 2 * N * 4)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
  t0 = time.time()
  lib.hostsim_solve(0, N, T, B, z.ctypes.data, l.ctypes.data, u.ctypes.data, None, 0, 1000, 1e-8, 1e-6, 1e-7, 0.1,
                    lam.ctypes.data, cost.ctypes.data, st.ctypes.data, it.ctypes.data, None)
  dt = time.time() - t0
  return {"value": float((st == 0).sum() / dt), "unit": "solves/s", "cores": os.cpu_count(), "kind": "host build of the same SQP core (OpenMP)",
          "sample": f"first {B} instances of the workload, {dt:.2f} s, {int((st == 0).sum())} converged"}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--batch", type=int, default=4096, help="instances per GPU")
  ap.add_argument("--intervals", type=int, default=100)
  ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the cpu_baseline sample (0 = skip)")
  a = ap.parse_args()

  import torch
  import torch.distributed as dist
  rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
  local %= torch.cuda.device_count()      # a launcher that narrows the visible devices per rank leaves one device, index 0
  torch.cuda.set_device(local)
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
  from myriad_amd import _lib

  B, N = a.batch, a.intervals
  x0, z0h, lbh, ubh, T = build_workload(B, N, seed=2019 + rank)
  eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, device=local, max_batch=B)
  dev = torch.device("cuda", local)
  f64 = dict(dtype=torch.float64, device=dev)
  z0 = torch.from_numpy(z0h).to(dev); lb = torch.from_numpy(lbh).to(dev); ub = torch.from_numpy(ubh).to(dev)
  z = torch.empty_like(z0)
  lam = torch.empty(B, eng.m, **f64); cost = torch.empty(B, **f64); kkt = torch.empty(B, 3, **f64)
  status = torch.empty(B, dtype=torch.int32, device=dev); iters = torch.empty(B, dtype=torch.int32, device=dev)
  fv = torch.empty(B, **f64); gv = torch.empty(B, eng.ngrad, **f64); cv = torch.empty(B, eng.m, **f64)
  jv = torch.empty(B, eng.jblk, **f64)
  from myriad_amd.batched import gather_solutions
  opts = eng.default_opts()
  opts.max_iter = 1000                       # hp.max_iter default (config.py:70)

  def step():
    z.copy_(z0)
    torch.cuda.current_stream().synchronize()            # library runs on its own stream
    eng.solve_device(B, z, lb, ub, None, 0, opts, lam, cost, status, iters, kkt)
    eng.eval_device(B, z, f=fv, gradf=gv, c=cv, jblk=jv)  # verification pass (also the roofline kernel)
    feas = cv.abs().amax(dim=1)
    ok = (status == 0) & (feas <= 1e-8)
    if world > 1:   # the path's only collective: final gather of the solutions over RCCL/xGMI
      gathered = gather_solutions({"z": z, "cost": cost, "status": status}, [B] * world)
      assert gathered["z"].shape[0] == B * world
    return ok

  def fence():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(a.warmup):
    int(step().sum().item())        # the same work as a timed step, including the count read-back
  eng.kernel_time_reset()
  fence()
  t0 = time.perf_counter()
  nconv = 0
  trace = []
  for _ in range(a.steps):
    ts = time.perf_counter()
    ok = step()
    nconv += int(ok.sum().item())
    trace.append(time.perf_counter() - ts)
  if os.environ.get("MYRIAD_BENCH_TRACE") and rank == 0:
    print("per-step ms:", " ".join("%.1f" % (1e3 * t) for t in trace), file=sys.stderr)
  fence()
  dt = time.perf_counter() - t0
  tt = torch.tensor([dt, float(nconv)], **f64)
  if world > 1:
    tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    dt = float(tmax[0]); nconv_all = float(tsum[1])
  else:
    nconv_all = float(nconv)
  ev_ms, ev_n = eng.kernel_time(_lib.K_EVAL)
  sv_ms, sv_n = eng.kernel_time(_lib.K_SOLVE)

  if rank == 0:
    itc = iters.cpu().numpy()
    alg = ALG_BYTES_PER_EVAL(N, 4, 1) * B
    traffic = None      # HBM bytes per launch from the PMC counters (rocprofv3, separate passes) -- committed summary
    tpath = os.path.join(ROOT, "profiles", "r01", "hs_eval_traffic.json")
    if os.path.exists(tpath) and B == 4096 and N == 100:
      traffic = json.load(open(tpath)).get("traffic_bytes_per_launch")
    out = {
      "metric": "converged trajopt solves/sec (batched), CARTPOLE collocation N=100",
      "value": nconv_all / dt, "unit": "solves/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
      "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f64", "data": "synthetic",
      "config": {"workload": f"CARTPOLE, COLLOCATION (Hermite-Simpson), intervals={N}, batch={B} random x0 per GPU "
                             f"(x0 = clip(x_0 + 0.1 N(0,I)), default_rng(2019+rank)), max_iter=1000, "
                             f"converged = status 0 and max|c| <= 1e-8 re-checked by the eval kernel",
                 "global_batch": B * world, "parallelism": f"instances sharded over {world} GPU(s); RCCL all_gather of z*, cost, status"},
      "converged_fraction": nconv_all / (a.steps * B * world),
      "iterations": {"median": float(np.median(itc)), "p99": float(np.percentile(itc, 99)), "max": int(itc.max())},
      "roofline": {"kernel": "hs_eval_kernel<CARTPOLE> (HS defect + Jacobian blocks + grad f)", "bound": "hbm",
                   "achieved": alg / (ev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                   "frac": alg / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                   "alg_bytes_per_launch": alg, "avg_ms": ev_ms, "launches": ev_n},
      "solver_kernel": {"kernel": ("hs_solve_wave_kernel<CARTPOLE> (one trajectory per wavefront, whole SQP in one launch)"
                                   if os.environ.get("MYRIAD_SOLVE_MODE", "wave") != "lane" else
                                   "hs_solve_kernel<CARTPOLE> (one trajectory per lane, whole SQP in one launch)"),
                        "avg_ms": sv_ms, "launches": sv_n, "bound": "latency/occupancy (see DESIGN.md)"},
    }
    if world == 1 and a.cpu_budget > 0:
      try:
        out["cpu_baseline"] = cpu_baseline(N, a.cpu_budget)
      except Exception as e:   # the oracle is test infrastructure: never let it break the measured line
        out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
      twin = cpu_same_algorithm(z0h, lbh, ubh, N, T, 512)
      if twin:
        out["cpu_same_algorithm"] = twin
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
