#!/bin/bash
# exp19: bench.py with the workload-specific mu_init, and the same option on the eight per-rank workloads of the weak-scaling run
cd /root/repo; mkdir -p gpurun_out/exp19
python bench.py > gpurun_out/exp19/bench_default.json 2> gpurun_out/exp19/bench_default.err
python bench.py --mu-init 0 --cpu-budget 0 --no-other-configs > gpurun_out/exp19/bench_mu_lib.json 2> gpurun_out/exp19/bench_mu_lib.err
python - > gpurun_out/exp19/seeds.txt 2>&1 <<'PY'
import numpy as np, torch, bench
N, B = 100, 4096
for seed in range(2019, 2027):
  x0, z0h, lbh, ubh, T = bench.build_workload(B, N, seed=seed)
  eng = bench.DeviceEngine(N, T, 0, B)
  dev = torch.device("cuda", 0); f64 = dict(dtype=torch.float64, device=dev)
  z0 = torch.from_numpy(np.ascontiguousarray(z0h)).to(dev); lb = torch.from_numpy(np.ascontiguousarray(lbh)).to(dev); ub = torch.from_numpy(np.ascontiguousarray(ubh)).to(dev)
  lam = torch.empty(B, eng.m, **f64); kkt = torch.empty(B, 3, **f64); cost = torch.empty(B, **f64)
  st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
  fv = torch.empty(B, **f64); gv = torch.empty(B, eng.ngrad, **f64); cv = torch.empty(B, eng.m, **f64); jv = torch.empty(B, eng.jblk, **f64)
  row = []
  for mu in (0.0, 0.003):
    eng.set_mu_init(mu)
    z = z0.clone(); torch.cuda.synchronize()
    eng.solve(B, z, lb, ub, lam, cost, st, it, kkt); eng.eval(B, z, fv, gv, cv, jv); torch.cuda.synchronize()
    ok = ((st == 0) & (cv.abs().amax(dim=1) <= 1e-8)).sum().item()
    i = it.cpu().numpy(); row.append((ok, float(np.median(i)), float(np.percentile(i, 99)), int(i.max()), float(cost.mean())))
  print(seed, "lib:", row[0], " mu=0.003:", row[1], flush=True)
PY
cat gpurun_out/exp19/seeds.txt
python -c "
import json
for f in ('bench_default','bench_mu_lib'):
  d=json.load(open('gpurun_out/exp19/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['converged_fraction'], d['iterations'], d['solver_kernel']['avg_ms'], d.get('solver_options'))
"
