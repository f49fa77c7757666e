"""exp64: the wider systems (two controls / six states / the small elastic twins) on the fused kernel's block sweep (riccati_mfma_gen)
against round 2's wavefront kernel (MYRIAD_SOLVE_MODE=wave1): kernel time of one batch, same inputs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import myriad_oracle as O

def run(name, rule, N, B, lim):
  twin = name.endswith("_ELASTIC")
  s = O.Elastic(O.SYSTEMS[name[:-8]](), 1.0) if twin else O.SYSTEMS[name]()
  tr = O.hermite_simpson(s, N) if rule == "HERMITE_SIMPSON" else O.trapezoidal(s, N)
  rng = np.random.default_rng(3)
  z0 = np.tile(tr.guess, (B, 1)); lb = np.tile(tr.bounds[:, 0], (B, 1)); ub = np.tile(tr.bounds[:, 1], (B, 1))
  x0 = z0[:, :s.ns] * (1.0 + 0.02 * rng.standard_normal((B, s.ns)))
  z0[:, :s.ns] = x0; lb[:, :s.ns] = x0; ub[:, :s.ns] = x0
  out = {}
  for mode in ("wave", "wave1"):
    os.environ["MYRIAD_SOLVE_MODE"] = mode
    os.environ["MYRIAD_SECOND_STARTS"] = "0"; os.environ["MYRIAD_ELASTIC"] = "0"
    from myriad_amd import _lib
    eng = _lib.Engine(name, rule, N, s.T)
    o = eng.default_opts(); o.restoration = 0; o.max_iter = lim
    best = 1e9
    for rep in range(3):
      eng.kernel_time_reset()
      r = eng.solve(z0, lb, ub, params=s.params() if twin else None, opts=o)
      ms, n = eng.kernel_time(_lib.K_SOLVE)
      best = min(best, ms)
    out[mode] = (best, int(r["iters"].sum()), int((r["status"] == 0).sum()), eng.solve_plan()["form"])
    eng.close()
  f, w = out["wave"], out["wave1"]
  print(f"{name:22s} {rule:16s} N={N:3d} B={B:5d}  fused {f[0]:8.2f} ms ({f[3]}, {f[1]} iterations, {f[2]} converged)   round-2 wavefront kernel {w[0]:8.2f} ms ({w[1]} iterations, {w[2]} converged)   x{w[0] / f[0]:.2f}", flush=True)

if __name__ == "__main__":
  for name, rule, N, B, lim in (("BEARPOPULATIONS", "HERMITE_SIMPSON", 100, 4096, 300), ("BEARPOPULATIONS", "TRAPEZOIDAL", 100, 4096, 300),
                                ("ROCKETLANDING", "HERMITE_SIMPSON", 100, 4096, 30), ("ROCKETLANDING", "TRAPEZOIDAL", 100, 4096, 30),
                                ("PENDULUM_ELASTIC", "HERMITE_SIMPSON", 100, 4096, 60), ("VANDERPOL_ELASTIC", "HERMITE_SIMPSON", 100, 4096, 300),
                                ("CARTPOLE_ELASTIC", "HERMITE_SIMPSON", 100, 4096, 60), ("CARTPOLE_ELASTIC", "TRAPEZOIDAL", 100, 4096, 60),
                                ("BEARPOPULATIONS", "HERMITE_SIMPSON", 100, 256, 300)):
    run(name, rule, N, B, lim)
  # ROCKETLANDING's twin: round 2's kernel is not built for it any more (MYRIAD_SOLVE_MODE=wave1 ends on the lane kernel)
  run("ROCKETLANDING_ELASTIC", "HERMITE_SIMPSON", 100, 1024, 30)
  run("ROCKETLANDING_ELASTIC", "TRAPEZOIDAL", 100, 1024, 30)
