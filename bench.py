#!/usr/bin/env python3
"""bench.py -- headline benchmark: converged trajectory-optimisation solves per second (batched),
CARTPOLE Hermite-Simpson collocation, 100 intervals, B = 4096 random x0 per GPU (BASELINE.json configs[1]).

One "step" = one pass of the hot path over one batch:
   z0 -> [myr_solve: batched SQP on device] -> z*, lambda*, cost, status
      -> [myr_eval : hs_eval kernel on z*]   -> c(z*), J blocks, grad f   (independent convergence verification)
      -> (N > 1) RCCL gather of z*, cost, status to rank 0 (the path's only collective)
      -> download of z*, cost, status into pinned host buffers on rank 0 (side stream; SURVEY.md 8(d): "gathered on host rank 0").
Inputs are resident in HBM before the timed region.  Multi-GPU: independent instances are sharded across ranks, one
process per GPU.  `--scaling weak` (default): --batch instances PER GPU; `--scaling strong`: --batch instances in
total, split by myriad_amd.batched.shard_range (4096 -> 512 per GPU at 8 GPUs).
Launch: the driver starts N ranks with torch.distributed.run; `python bench.py --gpus N` WITHOUT a launcher starts them
itself (re-exec under torch.distributed.run on 127.0.0.1) -- --gpus is never silently ignored.

Prints ONE JSON line on rank 0 (see the task contract), including
  roofline     -- the hs_eval kernel (the defect+Jacobian kernel of SURVEY.md 8(d)): algorithmic bytes per launch /
                  HIP-event duration measured on the library's own stream inside the timed region;
  cpu_baseline -- the oracle's SciPy SLSQP path (the reference's NLPSolverType.SLSQP branch on restated callbacks)
                  timed on this box's host cores: one FULL converged solve at N=25 (measured) and a time-bounded sample
                  of the N=100 solve (iteration rate measured, solve rate extrapolated -- labelled so);
  other_configs -- (N = 1, after the timed region, not part of `value`) the other BASELINE configs on one GPU and the headline
                  workload through the host-buffer API solve_batch (numpy in, numpy out: the PCIe-inclusive rate),
                  tools/bench_configs.py; --no-other-configs skips them.
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
  Two measurements, both reported:
    full_solve -- ONE complete converged solve of the same problem at N=25 (about 4 s): measured solves/s, nothing
                  extrapolated, but a smaller transcription than the metric's;
    value      -- the metric's own size (N=100): a full solve takes minutes (546 s for the reference), so
                  SLSQP runs on instance 0 until `budget_s` of wall time is used, the measured iteration rate is divided
                  by the 110 iterations that solve needs at the reference's default tolerance => EXTRAPOLATED.
  SciPy's SLSQP (Fortran) is serial; the callbacks are torch autodiff, which uses `torch_threads` for its kernels."""
  from oracle import myriad_oracle as O
  import torch
  s = O.CartPole()

  def timed_solve(n, budget):
    tr = O.hermite_simpson(s, n)
    cb = O.Callbacks(tr)
    cb.jac(tr.guess); cb.grad(tr.guess)        # warm up autodiff

    class _Timeout(Exception):
      pass

    t0 = time.time()
    njac = [0]
    jac0 = cb.jac

    def timed_jac(z):                          # one Jacobian evaluation per SLSQP major iteration
      if budget is not None and time.time() - t0 > budget:
        raise _Timeout()
      njac[0] += 1
      return jac0(z)

    cb.jac = timed_jac
    done, cost = True, None
    try:
      r = O.solve(tr, "SLSQP", max_iter=1000, cb=cb)
      cost = float(r["cost"])
    except _Timeout:
      done = False
    return time.time() - t0, max(1, njac[0] - 1), done, cost

  dt25, it25, ok25, cost25 = timed_solve(25, None)
  dt, nit, done, _ = timed_solve(N, budget_s)
  its_per_s = nit / dt
  full_its, measured = 110, None
  for rnd in ("r06", "r05", "r04", "r03"):          # a whole solve measured on a GPU-box host by `bench.py --cpu-full` (committed)
    fp = os.path.join(ROOT, "profiles", rnd, "cpu_baseline_full.json")
    if N == 100 and os.path.exists(fp):
      try:
        m = json.load(open(fp))["slsqp_full_solve"]
        full_its = int(m["iterations"])
        measured = {"file": os.path.relpath(fp, ROOT), "seconds": m["seconds"], "iterations": m["iterations"], "value": m["value"],
                    "host_cores": json.load(open(fp)).get("host_cores"),
                    "value_host": {"round": rnd, "hostname": json.load(open(fp)).get("hostname"), "measured_utc": json.load(open(fp)).get("measured_utc"),
                                   "note": "where and when the committed whole solve was measured: a stale file shows here"},
                    "trust_constr": json.load(open(fp)).get("trust_constr_subsample", {}).get("value")}
        break
      except Exception:
        pass
  # value: a WHOLE measured solve -- this run's own when the sample ran to convergence, else the committed whole solve of the same
  # problem on a host of this pool (`measured_full_solve.file`); the bounded sample of this run is the cross-check beside it
  # (`sample_its_per_s`, `sample_value_extrapolated`: the first iterations of SLSQP are its slowest, so the extrapolation reads low)
  if done:
    value, src = 1.0 / dt, "this run"
  elif measured:
    value, src = float(measured["value"]), measured["file"]
  else:
    value, src = its_per_s / full_its, "extrapolated from this run's sample"
  return {"value": value, "unit": "solves/s", "cores": 1, "torch_threads": int(torch.get_num_threads()),
          "host_cores": os.cpu_count() or 1, "kind": "port", "extrapolated": (not done and not measured), "value_from": src,
          "sample_its_per_s": its_per_s, "sample_value_extrapolated": its_per_s / full_its, "measured_full_solve": measured,
          "full_solve": {"workload": "CARTPOLE HS N=25, default x0, SLSQP to its default tolerance", "seconds": dt25,
                         "iterations": it25, "converged": bool(ok25), "cost": cost25, "value": 1.0 / dt25, "unit": "solves/s"},
          "sample": f"oracle SciPy-SLSQP path (serial Fortran SLSQP + torch-autodiff callbacks), CARTPOLE HS N={N} instance 0 "
                    f"(default x0): {nit} SLSQP iterations in {dt:.1f} s ({its_per_s:.3f} it/s)" +
                    ("; ran to convergence" if done else
                     f"; a converged solve needs {full_its} iterations at the reference's default tolerance (" +
                     (f"measured: {measured['file']}, {measured['seconds']:.0f} s = {measured['value']:.5f} solves/s on a {measured['host_cores']}-core GPU-box host: that is `value`; "
                      f"this run's sample extrapolates to {its_per_s / full_its:.5f}" if measured else "BASELINE.md: 546 s measured for the reference; value = this run's it/s / 110") +
                     "); whole measured solves: `measured_full_solve` (N=100) and `full_solve` (N=25)")}


def cpu_full(N, out_path, trust_budget_s=600.0):
  """`python bench.py --cpu-full FILE`: the MEASURED legs of the CPU baseline, on the host cores of the box it
  runs on (no GPU involved): (1) ONE whole SLSQP solve of the metric's problem (CARTPOLE HS N intervals, default x0) to SciPy's default
  tolerance -- the reference's NLPSolverType.SLSQP branch (/root/reference/myriad/nlp_solvers/__init__.py:50-52) on the oracle's restated
  callbacks; about nine minutes at N=100; (2) the trust-constr branch (:53-55) on up to 8 instances of the workload's start-state
  rule at N=25 with the reference's maxiter=1000 inside `trust_budget_s` of wall time (at N=100 one trust-constr solve ran into
  maxiter after 24 minutes, SURVEY.md App. C -- a subsample at that size does not fit any bench budget; the N=100 rate is sampled
  for 60 s instead).  The default run cites the file this writes."""
  from oracle import myriad_oracle as O
  import torch
  import datetime, socket
  res = {"host_cores": os.cpu_count() or 1, "torch_threads": int(torch.get_num_threads()), "kind": "port",
         "hostname": socket.gethostname(), "measured_utc": datetime.datetime.utcnow().strftime("%Y-%m-%dT%H:%M:%SZ"),
         "what": "oracle SciPy path (serial SLSQP / trust-constr + torch-autodiff callbacks) = the reference's SciPy branches minus JAX"}
  s = O.CartPole()
  tr = O.hermite_simpson(s, N)
  cb = O.Callbacks(tr); cb.jac(tr.guess); cb.grad(tr.guess)
  t0 = time.time()
  r = O.solve(tr, "SLSQP", max_iter=1000, cb=cb)
  dt = time.time() - t0
  res["slsqp_full_solve"] = {"workload": f"CARTPOLE HS N={N}, default x0, SLSQP to its default tolerance (ftol=1e-6), maxiter=1000",
                             "seconds": dt, "iterations": int(r["scipy"].nit), "converged": bool(r["scipy"].success), "cost": float(r["cost"]),
                             "max_abs_c": float(np.abs(cb.cons(r["xs_and_us"])).max()), "value": 1.0 / dt, "unit": "solves/s", "cores": 1}
  print("[cpu-full] SLSQP", json.dumps(res["slsqp_full_solve"]), file=sys.stderr, flush=True)
  x0s = np.vstack([s.x_0[None], O.random_x0(s, 7, seed=2019)])
  rows, t_all = [], time.time()
  for b in range(8):
    if time.time() - t_all > trust_budget_s:
      break
    sb = O.CartPole(); sb.x_0 = x0s[b].copy()
    trb = O.hermite_simpson(sb, 25)
    t0 = time.time()
    rb = O.solve(trb, "TRUST", max_iter=1000)
    rows.append({"instance": b, "seconds": time.time() - t0, "iterations": int(rb["scipy"].nit), "converged": bool(rb["scipy"].success),
                 "cost": float(rb["cost"]), "max_abs_c": float(np.abs(O.Callbacks(trb).cons(rb["xs_and_us"])).max())})
    print("[cpu-full] trust-constr", json.dumps(rows[-1]), file=sys.stderr, flush=True)
  tt = sum(r_["seconds"] for r_ in rows)
  res["trust_constr_subsample"] = {"workload": "CARTPOLE HS N=25, x0 = default + clip(x_0 + 0.1 N(0,I)) from default_rng(2019), trust-constr, maxiter=1000",
                                   "instances": rows, "value": (len(rows) / tt) if rows else None, "unit": "solves/s (finished or at maxiter)", "cores": 1}

  class _Stop(Exception):
    pass
  t0 = time.time(); nit = [0]
  def cbk(xk, state=None):
    nit[0] += 1
    if time.time() - t0 > 60.0:
      raise _Stop()
  try:
    from scipy.optimize import minimize
    minimize(fun=cb.fun, x0=tr.guess, method="trust-constr", jac=cb.grad, constraints=({"type": "eq", "fun": cb.cons, "jac": cb.jac}),
             bounds=tr.bounds, options={"maxiter": 1000}, callback=cbk)
  except _Stop:
    pass
  res["trust_constr_rate_at_N"] = {"workload": f"CARTPOLE HS N={N}, default x0, trust-constr sampled for 60 s", "iterations": nit[0],
                                   "seconds": time.time() - t0, "its_per_s": nit[0] / (time.time() - t0)}
  with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
  print(json.dumps(res))


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

  def run(nb):
    z = np.ascontiguousarray(z0[:nb]).copy(); l = np.ascontiguousarray(lb[:nb]); u = np.ascontiguousarray(ub[:nb])
    lam = np.zeros((nb, 2 * N * 4)); cost = np.zeros(nb); st = np.zeros(nb, np.int32); it = np.zeros(nb, np.int32)
    t0 = time.time()
    lib.hostsim_solve(0, N, T, nb, z.ctypes.data, l.ctypes.data, u.ctypes.data, None, 0, 1000, 1e-8, 1e-6, 1e-7, 0.1,
                      lam.ctypes.data, cost.ctypes.data, st.ctypes.data, it.ctypes.data, None)
    return time.time() - t0, int((st == 0).sum())

  run(min(B, 2 * (os.cpu_count() or 1)))       # warm-up call: OpenMP thread start-up is not part of the sample
  dt, nconv = run(B)
  return {"value": float(nconv / dt), "unit": "solves/s", "cores": os.cpu_count(), "kind": "host build of the same SQP core (OpenMP)",
          "sample": f"all {B} instances of the workload after a warm-up call, {dt:.2f} s, {nconv} converged"}


def _free_port():
  import socket
  with socket.socket() as so:
    so.bind(("127.0.0.1", 0))
    return so.getsockname()[1]


def self_launch(gpus):
  """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver would
  (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...), and pass their exit code on."""
  import subprocess
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
         "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
  print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (gpus, " ".join(cmd)), file=sys.stderr, flush=True)
  return subprocess.call(cmd, env=env)


class DeviceEngine:
  """What bench.run() needs from the engine (the product path: myriad_amd._lib.Engine over the C-ABI).  tests/ swap in a
  CPU stub with the same four methods to run the multi-rank logic (sharding, gather, reductions, JSON) under gloo."""

  def __init__(self, N, T, device, max_batch):
    from myriad_amd import _lib
    self._lib = _lib
    self.eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", N, T, device=device, max_batch=max_batch)
    self.m, self.ngrad, self.jblk = self.eng.m, self.eng.ngrad, self.eng.jblk
    self.opts = self.eng.default_opts()
    self.opts.max_iter = 1000                  # hp.max_iter default (config.py:70)
    self.library_mu_init = float(self.opts.mu_init)

  def set_mu_init(self, v):
    """Initial barrier parameter of the interior-point iteration (myr_solve_opts.mu_init, a public option).  0 = the library default."""
    self.opts.mu_init = float(v) if v and v > 0 else self.library_mu_init
    return float(self.opts.mu_init)

  def solve(self, B, z, lb, ub, lam, cost, status, iters, kkt):
    self.eng.solve_device(B, z, lb, ub, None, 0, self.opts, lam, cost, status, iters, kkt)

  def eval(self, B, z, fv, gv, cv, jv):
    self.eng.eval_device(B, z, f=fv, gradf=gv, c=cv, jblk=jv)

  def timer_reset(self):
    self.eng.kernel_time_reset()

  def timers(self):
    return self.eng.kernel_time(self._lib.K_EVAL), self.eng.kernel_time(self._lib.K_SOLVE)

  def plan(self):
    """myr_solve_plan: how the library launched the last solve (the one source of truth for the description of the line)"""
    return self.eng.solve_plan()


def run(a, rank, world, dev, make_engine):
  """The measured loop on one rank; returns the JSON dict on rank 0 (None elsewhere).  `dev` is a torch.device; the
  process group (if world > 1) is already initialised."""
  import torch
  import torch.distributed as dist
  from myriad_amd.batched import gather_solutions, shard_range
  N = a.intervals
  cuda = dev.type == "cuda"
  if a.scaling == "strong":
    total = a.batch
    lo, hi = shard_range(total, rank, world)
    counts = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    # one global workload, every rank takes its contiguous shard (so the union is identical for every N)
    x0, z0h, lbh, ubh, T = build_workload(total, N, seed=2019)
    z0h, lbh, ubh = z0h[lo:hi], lbh[lo:hi], ubh[lo:hi]
  else:
    total = a.batch * world
    counts = [a.batch] * world
    x0, z0h, lbh, ubh, T = build_workload(a.batch, N, seed=2019 + rank)
  B = counts[rank]
  eng = make_engine(N, T, dev, B)
  # --mu-init: a workload-specific initial barrier parameter (an ablation knob; the default run uses the library's options).  When set it
  # is named in the line, and the same K steps are timed at the library default as well.
  mu_req = float(getattr(a, "mu_init", 0.0) or 0.0)
  mu_used = eng.set_mu_init(mu_req) if hasattr(eng, "set_mu_init") else None
  mu_lib = getattr(eng, "library_mu_init", None)
  f64 = dict(dtype=torch.float64, device=dev)
  z0 = torch.from_numpy(np.ascontiguousarray(z0h)).to(dev); lb = torch.from_numpy(np.ascontiguousarray(lbh)).to(dev)
  ub = torch.from_numpy(np.ascontiguousarray(ubh)).to(dev)
  # two sets of result buffers: the download of step k's solutions (side stream) overlaps step k + 1's solve
  zs = [torch.empty_like(z0) for _ in range(2)]
  costs = [torch.empty(B, **f64) for _ in range(2)]
  stats = [torch.empty(B, dtype=torch.int32, device=dev) for _ in range(2)]
  lam = torch.empty(B, eng.m, **f64); kkt = torch.empty(B, 3, **f64)
  iters = torch.empty(B, dtype=torch.int32, device=dev)
  # the metric ends with "all z* / status gathered on HOST rank 0" -- pinned host buffers on rank 0, filled inside the step
  host, d2h_done, copy_stream = None, [None, None], None
  if rank == 0:
    def _pin(t):
      return t.pin_memory() if cuda else t
    host = [{"z": _pin(torch.empty(total, z0.shape[1], dtype=torch.float64)), "cost": _pin(torch.empty(total, dtype=torch.float64)),
             "status": _pin(torch.empty(total, dtype=torch.int32))} for _ in range(2)]
  if cuda:
    copy_streams = [torch.cuda.Stream(device=dev) for _ in range(int(os.environ.get('MYRIAD_BENCH_COPY_STREAMS', '1')))]
    copy_stream = copy_streams[0]
  d2h_bytes = total * (z0.shape[1] * 8 + 8 + 4)
  fv = torch.empty(B, **f64); gv = torch.empty(B, eng.ngrad, **f64); cv = torch.empty(B, eng.m, **f64)
  jv = torch.empty(B, eng.jblk, **f64)

  def sync():
    if cuda:
      torch.cuda.synchronize()

  nstep = [0]
  tracing = bool(os.environ.get("MYRIAD_BENCH_TRACE")) and rank == 0
  segs = []

  def step(download=True):
    cur = nstep[0] & 1; nstep[0] += 1
    z, cost, status = zs[cur], costs[cur], stats[cur]
    if cuda and d2h_done[cur] is not None:
      torch.cuda.current_stream().wait_event(d2h_done[cur])   # this set's previous download (two steps ago) has left the device
    seg = [time.perf_counter()] if tracing else None
    z.copy_(z0)
    if cuda:
      torch.cuda.current_stream().synchronize()            # library runs on its own stream
    if tracing: seg.append(time.perf_counter())
    eng.solve(B, z, lb, ub, lam, cost, status, iters, kkt)
    if tracing: seg.append(time.perf_counter())
    eng.eval(B, z, fv, gv, cv, jv)                           # verification pass (also the roofline kernel)
    if tracing: seg.append(time.perf_counter())
    feas = cv.abs().amax(dim=1)
    ok = (status == 0) & (feas <= 1e-8)
    res = {"z": z, "cost": cost, "status": status}
    # the count is read back BEFORE the download is queued: a small device-to-host copy issued behind the 33 MB of z* can end up
    # on the same copy engine and wait for it (0.6 ms per step once the runtime has more than one engine queue open)
    n_ok = int(ok.sum().item())
    if tracing: seg.append(time.perf_counter()); segs.append(seg)

    def ship(res):
      if world > 1:   # the path's only collective: final gather of the solutions to rank 0 over RCCL/xGMI
        res = gather_solutions(res, counts, dst=0)
        if rank == 0:
          assert res["z"].shape[0] == total
      if rank == 0 and download:      # ... and down to the host: pinned buffers
        # (ONE of these enqueues per process, at a random step, holds the host for ~6 ms -- a one-time event inside the runtime's copy
        #  path that neither the order, nor a second copy stream, nor its pool / queue settings move: profiles/r04/README.md)
        for k in sorted(res, key=lambda k_: res[k_].numel()):
          t = res[k]
          t0_ = time.perf_counter()
          host[cur][k].copy_(t, non_blocking=cuda)
          if tracing and time.perf_counter() - t0_ > 1e-3:
            print("slow enqueue of the download of %s: %.2f ms (step %d)" % (k, 1e3 * (time.perf_counter() - t0_), nstep[0]), file=sys.stderr)

    if cuda:
      # gather and download run on a SIDE stream of every rank, behind the step that produced the solutions: they overlap the next
      # step's solve (which writes the other buffer set), so neither the 231 MB that rank 0 receives at N = 8 nor its download sits
      # on the critical path of a step; the fence at the end of the timed region waits for all of it.
      ready = torch.cuda.Event(); ready.record()
      copy_stream = copy_streams[cur % len(copy_streams)]
      with torch.cuda.stream(copy_stream):
        copy_stream.wait_event(ready)
        ship(res)
        d2h_done[cur] = torch.cuda.Event(); d2h_done[cur].record(copy_stream)
    else:
      ship(res)
    if tracing: seg.append(time.perf_counter())
    return n_ok

  def fence():
    sync()
    if world > 1:
      dist.barrier()
    sync()

  # The asynchronous-copy path pays a ONE-TIME host stall of 5-10 ms inside one enqueue, 0.3-0.4 s after the process's first downloads -- at step 14..27
  # of a run, whatever the warm-up count, and EARLIER in step count the more copies ran before the steps: a matter of elapsed time under traffic, not of
  # a count (tools/dev/exp/exp73.sh).  Initialisation, like the first import: the download path is exercised for `prime_ms` before the steps -- the same
  # three pinned-buffer copies a step queues, on the stream the steps use; the longest enqueue seen is reported in the line.  (MYRIAD_BENCH_PRIME_MS=0: off.)
  prime_ms = float(os.environ.get("MYRIAD_BENCH_PRIME_MS", "700"))
  prime_info = None
  if cuda and rank == 0 and prime_ms > 0:
    tp0 = time.perf_counter(); worst = 0.0; ncp = 0
    with torch.cuda.stream(copy_streams[0]):
      while 1e3 * (time.perf_counter() - tp0) < prime_ms:
        for k, t in (("status", stats[0]), ("cost", costs[0]), ("z", zs[0])):
          te = time.perf_counter()
          host[ncp & 1][k][:B].copy_(t, non_blocking=True)
          worst = max(worst, time.perf_counter() - te)
        ncp += 1
        if ncp % 4 == 0:
          copy_streams[0].synchronize()
    sync()
    prime_info = {"ms": prime_ms, "rounds_of_three_copies": ncp, "longest_enqueue_ms": 1e3 * worst}
  for _ in range(a.warmup):
    step()        # the same work as a timed step, including the count read-back
  eng.timer_reset()
  fence()
  t0 = time.perf_counter()
  nconv = 0
  trace = []
  for _ in range(a.steps):
    ts = time.perf_counter()
    last_ok = step()
    nconv += last_ok
    trace.append(time.perf_counter() - ts)
  if tracing:
    print("per-step ms:", " ".join("%.1f" % (1e3 * t) for t in trace), file=sys.stderr)
    k = int(np.argmax(trace)); sg = segs[a.warmup + k]       # host-side segments of the slowest step: wait + copy z0, solve, eval (the rest: count, ship)
    print("slowest step %d: copy %.2f solve %.2f eval %.2f check+count %.2f ship %.2f after %.2f ms" % (k, 1e3 * (sg[1] - sg[0]), 1e3 * (sg[2] - sg[1]), 1e3 * (sg[3] - sg[2]),
          1e3 * (sg[4] - sg[3]), 1e3 * (sg[5] - sg[4]), 1e3 * (trace[k] - (sg[5] - sg[0]))), file=sys.stderr)
    med = int(np.argsort(trace)[len(trace) // 2]); sg = segs[a.warmup + med]
    print("median step %d: copy %.2f solve %.2f eval %.2f check+count %.2f ship %.2f after %.2f ms" % (med, 1e3 * (sg[1] - sg[0]), 1e3 * (sg[2] - sg[1]), 1e3 * (sg[3] - sg[2]),
          1e3 * (sg[4] - sg[3]), 1e3 * (sg[5] - sg[4]), 1e3 * (trace[med] - (sg[5] - sg[0]))), file=sys.stderr)
  fence()
  dt = time.perf_counter() - t0
  (ev_ms, ev_n), (sv_ms, sv_n) = eng.timers()      # kernel timers of the measured loop only
  itc = iters.cpu().numpy().copy()
  # cross-check of the download: the last step's solutions are on the host (status of every instance, z* finite)
  if rank == 0:
    last = host[(nstep[0] - 1) & 1]
    assert bool(torch.isfinite(last["z"]).all()) and int((last["status"] == 0).sum()) >= last_ok
  # the same K steps once more WITHOUT the download (informative: what the download costs; not `value`)
  fence()
  t1 = time.perf_counter()
  for _ in range(a.steps):
    step(download=False)
  fence()
  dt_nodl = time.perf_counter() - t1
  # ... and once more at the library's default options when the measured loop ran with a workload-specific one (informative; not `value`)
  dt_lib = nconv_lib = 0.0
  tuned = mu_used is not None and mu_lib is not None and mu_used != mu_lib
  if tuned:
    eng.set_mu_init(0.0)
    step()
    fence()
    t2 = time.perf_counter()
    for _ in range(a.steps):
      nconv_lib += step()
    fence()
    dt_lib = time.perf_counter() - t2
    itc_lib = iters.cpu().numpy().copy()
    eng.set_mu_init(mu_used)
  tt = torch.tensor([dt, float(nconv), dt_nodl, dt_lib, float(nconv_lib)], dtype=torch.float64, device=dev if (world == 1 or dist.get_backend() == "nccl") else "cpu")
  if world > 1:
    tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    dt = float(tmax[0]); nconv_all = float(tsum[1]); dt_nodl = float(tmax[2]); dt_lib = float(tmax[3]); nconv_lib = float(tsum[4])
  else:
    nconv_all = float(nconv)
  if rank != 0:
    return None
  alg = ALG_BYTES_PER_EVAL(N, 4, 1) * B
  # HBM traffic: NOT measured in this run (PMC counters need rocprofv3 passes of their own); cited from the committed
  # summaries of the same command, per launch, with the file named
  prof = {}
  fused = os.environ.get("MYRIAD_SOLVE_MODE", "wave") == "wave"
  skey = "hs_solve_fused_kernel" if fused else "hs_solve_wave_kernel"
  for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
    sp = os.path.join(ROOT, "profiles", rnd, "pmc_bench_n1.json")
    if "eval" not in prof and os.path.exists(sp) and B == 4096 and N == 100:     # the newest round's PMC passes carry the roofline kernel too
      try:
        d = json.load(open(sp))["hs_eval_kernel"]["derived_traffic_bytes"]
        prof["eval"] = (float(d["fetch_x2"]) + float(d["write"]), os.path.relpath(sp, ROOT))
      except Exception:
        pass
    if "solver" not in prof and os.path.exists(sp) and B == 4096 and N == 100:
      try:
        d = json.load(open(sp))[skey]["derived_traffic_bytes"]
        prof["solver"] = (float(d["fetch_x2"]) + float(d["write"]), os.path.relpath(sp, ROOT))
      except Exception:
        pass
    if "mfma" not in prof and os.path.exists(sp) and B == 4096 and N == 100:      # matrix instructions of the solver kernel, per dispatch (same file)
      try:
        prof["mfma"] = (float(json.load(open(sp))[skey]["per_dispatch_avg"]["SQ_INSTS_MFMA"]), os.path.relpath(sp, ROOT))
      except Exception:
        pass
    tp = os.path.join(ROOT, "profiles", rnd, "hs_eval_traffic.json")
    if "eval" not in prof and os.path.exists(tp) and B == 4096 and N == 100:
      prof["eval"] = (json.load(open(tp)).get("traffic_bytes_per_launch"), os.path.relpath(tp, ROOT))
  # how the library launched the solves of this line: asked of the library (myr_solve_plan), not restated here
  plan = eng.plan() if hasattr(eng, "plan") else {"launches_per_solve": 1, "park_iter": 0, "waves_per_trajectory": 1}
  two_phase = plan["park_iter"] > 0
  lps = plan["launches_per_solve"]
  park_k1 = plan["park_iter"]
  traffic, traffic_src = prof.get("eval", (None, None))
  sol_bytes, sol_src = prof.get("solver", (None, None))
  mfma_n, mfma_src = prof.get("mfma", (None, None))
  FP64_MFMA_PEAK_TFLOPS = 78.6          # MI355X_MICROARCH.md: dense fp64 matrix peak (= the vector peak on this part)
  mfma_tf = (lps * mfma_n * 2048.0 / (sv_ms * 1e-3) / 1e12) if (mfma_n and sv_ms) else None      # v_mfma_f64_16x16x4_f64: 16 x 16 x 4 x 2 flop
  out = {
    "metric": "converged trajopt solves/sec (batched), CARTPOLE collocation N=100",
    "value": nconv_all / dt, "unit": "solves/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
    "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
    "dtype": "f64", "data": "synthetic",
    "config": {"workload": f"CARTPOLE, COLLOCATION (Hermite-Simpson), intervals={N}, " +
                           (f"batch={a.batch} random x0 per GPU (default_rng(2019+rank))" if a.scaling == "weak" else
                            f"batch={a.batch} random x0 in total, sharded {counts} (default_rng(2019))") +
                           ", x0 = clip(x_0 + 0.1 N(0,I)), max_iter=1000" +
                           (f", mu_init={mu_used:g} (solver option chosen for this workload; library default {mu_lib:g})" if tuned else "") +
                           ", converged = status 0 and max|c| <= 1e-8 re-checked by the eval kernel",
               "global_batch": total, "per_gpu_batch": counts,
               "parallelism": f"instances sharded over {world} GPU(s), one process per GPU" +
                              (f"; {dist.get_backend()} group of {dist.get_world_size()} ranks, gather of z*, cost, status to rank 0"
                               if world > 1 else "")},
    "converged_fraction": nconv_all / (a.steps * total),
    "download": {"what": "z*, cost, status of all instances -> pinned host buffers on rank 0, inside every timed step (side stream, double-buffered: "
                         "overlaps the next step's solve); SURVEY.md 8(d): the metric ends on host rank 0",
                 "bytes_per_step": d2h_bytes, "value_without_download": nconv_all / dt_nodl, "ms_per_step_without_download": 1e3 * dt_nodl / a.steps,
                 "primed_before_the_steps": prime_info},
    "solver_options": ({"mu_init": mu_used, "library_default_mu_init": mu_lib,
                        "why": "initial barrier parameter of the interior-point iteration (public field of myr_solve_opts, IPOPT's mu_init): 0.003 "
                               "saves about two iterations per solve on this workload, same tolerances, same converged fraction; it is NOT the "
                               "library default because the shooting configurations need more iterations with it (profiles/r04/README.md)",
                        "value_at_library_defaults": (nconv_lib / dt_lib) if dt_lib else None,
                        "ms_per_step_at_library_defaults": (1e3 * dt_lib / a.steps) if dt_lib else None,
                        "converged_fraction_at_library_defaults": (nconv_lib / (a.steps * total)) if dt_lib else None,
                        "iterations_at_library_defaults": ({"median": float(np.median(itc_lib)), "p99": float(np.percentile(itc_lib, 99)),
                                                            "max": int(itc_lib.max())} if dt_lib else None)}
                       if tuned else {"mu_init": mu_used, "note": "library defaults"}),
    "iterations": {"median": float(np.median(itc)), "p99": float(np.percentile(itc, 99)), "max": int(itc.max())},
    "roofline": {"kernel": "hs_eval_kernel<CARTPOLE> (HS defect + Jacobian blocks + grad f)", "bound": "hbm",
                 "achieved": alg / (ev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": alg / (ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_from_profile": traffic_src,
                 "alg_bytes_per_launch": alg, "avg_ms": ev_ms, "launches": ev_n},
    "solver_kernel": {"kernel": ("hs_solve_fused_kernel<CARTPOLE> (persistent, one trajectory per wavefront, iterate in LDS, fused backward / forward phases, Riccati sweep on fp64 MFMA, whole SQP on the device" +
                                  (": TWO launches per solve -- the first %d iterations for every trajectory, the unfinished ones parked and resumed longest-first (myr_solve_opts.park_iter); avg_ms is the sum of both)" % park_k1
                                   if two_phase else " in one launch)")
                                 if fused else
                                 ("hs_solve_wave_kernel<CARTPOLE> (round-2 kernel: persistent, one trajectory per wavefront, thirteen phases through global records)"
                                  if os.environ.get("MYRIAD_SOLVE_MODE") == "wave1" else
                                  "hs_solve_kernel<CARTPOLE> (one trajectory per lane, whole SQP in one launch)")),
                      "avg_ms": sv_ms, "launches": sv_n, "bound": "dependent-instruction latency of one wavefront per SIMD (the Riccati sweep is 45 % of an iteration; see DESIGN.md section 4)",
                      "alg_io_bytes_per_launch": B * 8 * (3 * (2 * N + 1) * 5 + (2 * N + 1) * 5 + 2 * N * 4),
                      "launches_per_solve": lps, "plan": plan,
                      "hbm_bytes_per_launch_from_profile": sol_bytes, "hbm_profile": sol_src,       # (per KERNEL launch: the profile averages over both phases)
                      "hbm_GBps": (lps * sol_bytes / (sv_ms * 1e-3) / 1e9) if (sol_bytes and sv_ms) else None,
                      "hbm_over_alg": (lps * sol_bytes / (B * 8 * (3 * (2 * N + 1) * 5 + (2 * N + 1) * 5 + 2 * N * 4))) if sol_bytes else None,
                      # the matrix-pipe view of the same kernel: instruction count from the committed counter pass, duration measured in this run
                      "mfma": {"bound": "mfma", "insts_per_launch_from_profile": mfma_n, "profile": mfma_src, "achieved": mfma_tf, "peak": FP64_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": (mfma_tf / FP64_MFMA_PEAK_TFLOPS) if mfma_tf else None,
                               "note": "the Riccati sweep is a chain of dependent 16x16x4 products, one wavefront per SIMD: low by construction (SURVEY.md 8(d))"}},
  }
  if world == 1 and a.cpu_budget > 0 and cuda:
    try:
      out["cpu_baseline"] = cpu_baseline(N, a.cpu_budget)
    except Exception as e:   # the oracle is test infrastructure: never let it break the measured line
      out["cpu_baseline"] = {"value": None, "unit": "solves/s", "cores": 1, "kind": "port", "sample": f"failed: {e}"}
    twin = cpu_same_algorithm(z0h, lbh, ubh, N, T, B)
    if twin:
      out["cpu_same_algorithm"] = twin
  if world == 1 and cuda and not getattr(a, "no_other_configs", False):
    # the other BASELINE configs (3, 4, 5 at their per-GPU share and on one GPU, README:83's trapezoidal literal), a few solves
    # each AFTER the timed region: informative lines under the same driver clock, not part of `value`
    try:
      sys.path.insert(0, os.path.join(ROOT, "tools"))
      import bench_configs
      out["other_configs"] = bench_configs.measure()
    except Exception as e:
      out["other_configs"] = {"failed": repr(e)}
  return out


def contract_line(out):
  """The one JSON line of the bench contract, below 1 900 characters: metric, value, config, dtype, roofline, cpu_baseline and one
  {config: solves_per_s_wall} map for BASELINE configs 3 / 4 / 5 and README:83's literal.  Everything else is in the detailed line printed before it."""
  r = lambda v, n=4: (None if v is None else (round(float(v), n) if abs(float(v)) < 1e6 else round(float(v))))
  keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
  line = {k: out.get(k) for k in keep}
  line["value"] = r(out.get("value"), 1); line["ms_per_step"] = r(out.get("ms_per_step"))
  cfg = out.get("config", {})
  line["config"] = {"workload": "CARTPOLE HS collocation N=%s, batch=%s random x0 (%s), converged = status 0 and max|c|<=1e-8" % (
                        cfg.get("workload", "").split("intervals=")[-1].split(",")[0], cfg.get("global_batch"), out.get("scaling")),
                    "global_batch": cfg.get("global_batch"), "per_gpu_batch": cfg.get("per_gpu_batch"),
                    "parallelism": "instances sharded over %s GPU(s), one process per GPU%s" % (out.get("n_gpus"), ", RCCL gather to rank 0" if (out.get("n_gpus") or 1) > 1 else "")}
  line["converged_fraction"] = r(out.get("converged_fraction"), 6)
  it = out.get("iterations") or {}
  line["iterations"] = {k: it.get(k) for k in ("median", "p99", "max")}
  rf = out.get("roofline") or {}
  line["roofline"] = {"kernel": "hs_eval_kernel<CARTPOLE>", "bound": rf.get("bound"), "achieved": r(rf.get("achieved"), 1), "peak": rf.get("peak"), "unit": rf.get("unit"),
                      "frac": r(rf.get("frac")), "traffic": rf.get("traffic"), "avg_ms": r(rf.get("avg_ms"), 5)}
  sk = out.get("solver_kernel") or {}
  pl = sk.get("plan") or {}
  line["solver_kernel"] = {"avg_ms": r(sk.get("avg_ms")), "launches_per_solve": sk.get("launches_per_solve"), "waves_per_trajectory": pl.get("waves_per_trajectory"),
                           "mfma_frac": r((sk.get("mfma") or {}).get("frac"))}
  cb = out.get("cpu_baseline")
  if cb:
    line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"), "sample": str(cb.get("sample"))[:150]}
  oc = out.get("other_configs")
  if isinstance(oc, list):
    line["other_configs_solves_per_s_wall"] = {("%s B=%s" % (str(o.get("config"))[:34], o.get("B"))): round(o.get("solves_per_s_wall", 0.0)) for o in oc}
  elif oc:
    line["other_configs_solves_per_s_wall"] = oc
  osc = out.get("other_scaling")
  if osc:
    line["other_scaling"] = {k: (r(osc[k]) if isinstance(osc.get(k), float) else osc.get(k)) for k in ("scaling", "value", "ms_per_step", "global_batch", "converged_fraction") if k in osc}
  ps = out.get("projected_scaling")
  if ps:
    line["projected_scaling_file"] = ps.get("file")
  line["detail"] = "previous stdout line"
  txt = json.dumps(line)
  if len(txt) > 1900:            # never over the limit: drop the least important parts first
    for k in ("projected_scaling_file", "other_scaling", "iterations", "solver_kernel"):
      line.pop(k, None)
      if len(json.dumps(line)) <= 1900: break
  return line


def both_scalings(a, rank, world, dev, make_engine):
  """The line of `--scaling` (weak by default: --batch instances per GPU) and, for N > 1, the OTHER split of the same workload measured in the same run
  right after it (`other_scaling`): SURVEY.md 8(e) partitions ONE batch over the GPUs (strong: 4096 -> 512 per GPU at N = 8, where a launch is one
  solve long and the single-GPU rate at that size bounds the efficiency -- profiles/r05/projected_scaling.json holds the expectation), the contract's
  default keeps the work per GPU fixed.  Whoever reads a SCALE file gets both numbers under one clock."""
  import copy
  out = run(a, rank, world, dev, make_engine)
  if world > 1 and not getattr(a, "no_other_scaling", False):
    b = copy.copy(a)
    b.scaling = "strong" if a.scaling == "weak" else "weak"
    if b.scaling == "weak":
      b.batch = max(1, a.batch // world)            # the per-GPU share of the strong line, held fixed
    b.cpu_budget = 0; b.no_other_configs = True; b.warmup = max(1, min(a.warmup, 2))
    o2 = run(b, rank, world, dev, make_engine)
    if rank == 0 and out is not None and o2 is not None:
      out["other_scaling"] = {k: o2[k] for k in ("scaling", "value", "unit", "ms_per_step", "steps", "warmup", "converged_fraction") if k in o2}
      out["other_scaling"]["global_batch"] = o2["config"]["global_batch"]; out["other_scaling"]["per_gpu_batch"] = o2["config"]["per_gpu_batch"]
      out["other_scaling"]["solver_kernel_avg_ms"] = o2["solver_kernel"]["avg_ms"]
  if rank == 0 and out is not None:
    pj = next((q for q in (os.path.join(ROOT, "profiles", r, "projected_scaling.json") for r in ("r06", "r05")) if os.path.exists(q)), None)
    if pj:
      try:
        out["projected_scaling"] = {"file": os.path.relpath(pj, ROOT), "config2": json.load(open(pj)).get("config2")}
      except Exception:
        pass
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--batch", type=int, default=4096, help="instances per GPU (weak scaling) / in total (strong scaling)")
  ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
  ap.add_argument("--intervals", type=int, default=100)
  ap.add_argument("--mu-init", type=float, default=0.0,
                  help="initial barrier parameter of the solves in the timed region (0 = the library default 0.1, which is what the headline is "
                       "measured with; a non-zero value is named in the line and the same steps are timed at the default as well)")
  ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the N=100 cpu_baseline sample (0 = skip)")
  ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: skip the second line (the other of weak / strong) measured after the first")
  ap.add_argument("--no-other-configs", action="store_true", help="skip the informative solves of BASELINE configs 3/4/5 after the timed region")
  ap.add_argument("--cpu-full", metavar="FILE", default=None,
                  help="measure the whole CPU baseline (one full N-interval SLSQP solve, ~9 min, + a trust-constr subsample) on this host, write FILE, exit")
  a = ap.parse_args()
  if a.cpu_full:
    cpu_full(a.intervals, a.cpu_full)
    return

  if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
    sys.exit(self_launch(a.gpus))

  import torch
  import torch.distributed as dist
  rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if world != a.gpus:
    raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
  have_gpu = torch.cuda.is_available()
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if have_gpu:
      local %= torch.cuda.device_count()  # a launcher that narrows the visible devices per rank leaves one device, index 0
      torch.cuda.set_device(local)
      # MYRIAD_BENCH_BACKEND=gloo: the REHEARSAL form of the N > 1 path for a box with fewer GPUs than ranks -- RCCL refuses two ranks on
      # one device ("Duplicate GPU detected"), gloo does not care: every rank runs the real engine on its (shared) device, the gather
      # goes through host memory.  Never the measured configuration: the line says so in `config.parallelism`.
      if os.environ.get("MYRIAD_BENCH_BACKEND", "nccl") == "gloo":
        dist.init_process_group("gloo", rank=rank, world_size=world)
      else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
      dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == 0:
      print(f"[bench] process group formed: backend={dist.get_backend()} ranks={dist.get_world_size()}", file=sys.stderr, flush=True)
  if not have_gpu:
    if world > 1:
      dist.barrier()
      dist.destroy_process_group()
    raise SystemExit("bench.py needs a GPU on every rank (the product path has no CPU fallback): torch.cuda.is_available() is False")
  if world == 1:
    local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
  mk = lambda N, T, dev, B: DeviceEngine(N, T, dev.index, B)
  out = both_scalings(a, rank, world, torch.device("cuda", local), mk)
  if rank == 0:
    # the detailed record first, the contract line LAST and short (a reader that keeps only the tail of stdout still gets every field of the
    # contract and the other configurations' rates; `detail` says where the rest is)
    print(json.dumps(dict(out, record="detail (the contract line is the last line of stdout)")), flush=True)
    print(json.dumps(contract_line(out)), flush=True)
  if world > 1:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
