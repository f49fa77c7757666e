#!/usr/bin/env python3
"""Kernel-form agreement probe: one problem, several kernel configurations (environment sets), bit-level comparison.
  python tools/dev/wprobe.py CASES CONFIGS [--lib PATH]
CASES:   comma list of SYSTEM:RULE:N:B   (RULE = HS | TRAP)   or the keyword `all` (every system x rule x N in 6,20,50,100 x B in 1,3)
CONFIGS: comma list of tags, each a '+'-joined set of KEY=VALUE (environment), e.g.  w1=MYRIAD_FUSED_WAVES=1  is written
         "MYRIAD_FUSED_WAVES=1"; "" is the default build.  The first config is the reference the others are compared with.
Prints per case and config: status, iterations, cost, and a 64-bit hash of z*; a line `DIFF` when a config differs from the first."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

def main():
  cases_arg, cfg_arg = sys.argv[1], sys.argv[2]
  max_iter = int(os.environ.get("WPROBE_MAX_ITER", "300"))
  os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
  from myriad_amd.config import Config, HParams, NLPSolverType, OptimizerType, QuadratureRule
  from myriad_amd.systems import SystemType
  from myriad_amd.trajectory_optimizers import get_optimizer
  if cases_arg == "all":
    cases = [(st.name, r, N, B) for st in SystemType if st.name not in ("INVASIVEPLANT",) for r in ("HS", "TRAP") for N in (6, 20, 50, 100) for B in (1, 3)]
  else:
    cases = []
    for c in cases_arg.split(","):
      s, r, N, B = c.split(":"); cases.append((s, r, int(N), int(B)))
  cfgs = [dict(kv.split("=", 1) for kv in c.split("+") if kv) for c in cfg_arg.split(",")]
  keys = sorted({k for c in cfgs for k in c})
  nbad = ncmp = 0
  for (s, r, N, B) in cases:
    outs = []
    for cfg in cfgs:
      for k in keys:
        os.environ.pop(k, None)
      os.environ.update(cfg)
      fill = os.environ.get("WPROBE_FILL")       # leave a bit pattern in every CU's LDS and in freed device memory before each solve
      if fill:
        import ctypes
        lf = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "variants", "libldsfill.so"))
        lf.lds_fill.argtypes = [ctypes.c_ulonglong]; lf.mem_fill.argtypes = [ctypes.c_ulonglong, ctypes.c_size_t]
        pat = {"nan": 0x7ff4dead0000beef, "big": 0x4415af1d78b58c40, "zero": 0, "one": 0x3ff0000000000000}[fill]
        assert lf.mem_fill(pat, 1 << 30) == 0 and lf.lds_fill(pat) == 0
      try:
        hp = HParams(system=SystemType[s], optimizer=OptimizerType.COLLOCATION, quadrature_rule=QuadratureRule["HERMITE_SIMPSON" if r == "HS" else "TRAPEZOIDAL"],
                     intervals=N, nlpsolver=NLPSolverType.SQP)
        opt = get_optimizer(hp, Config(verbose=False, plot=False), hp.system())
        x0 = np.tile(opt.system.x_0, (B, 1)) * (1.0 + 0.01 * np.arange(B)[:, None])
        o = opt.solve_batch(x0s=x0, max_iter=max_iter)
        outs.append((o["status"].copy(), o["iters"].copy(), o["cost"].copy(), hashlib.sha1(np.ascontiguousarray(o["xs_and_us"]).tobytes()).hexdigest()[:12]))
      except Exception as e:
        outs.append(None)
        print(f"{s} {r} N={N} B={B} [{cfg}]: {type(e).__name__} {str(e)[:100]}")
    ref = outs[0]
    for cfg, o in zip(cfgs, outs):
      if o is None or ref is None:
        continue
      ncmp += 1
      same_path = np.array_equal(o[0], ref[0]) and np.array_equal(o[1], ref[1]) and np.allclose(o[2], ref[2], rtol=1e-9, atol=1e-12, equal_nan=True)
      bit = o[3] == ref[3]
      tag = "ok  " if same_path else "DIFF"
      if not same_path:
        nbad += 1
      if not same_path or not bit or os.environ.get("WPROBE_VERBOSE"):
        print(f"{tag} {s} {r} N={N} B={B} [{'+'.join(f'{k}={v}' for k, v in cfg.items()) or 'default'}]: status {o[0]} iters {o[1]} cost {o[2]} z#{o[3]}{'' if bit else ' (bits differ from first config)'}")
  print(f"compared {ncmp} (case, config) pairs, {nbad} path mismatches")

if __name__ == "__main__":
  main()
