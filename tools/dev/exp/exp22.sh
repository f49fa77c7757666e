#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp22
MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/libtr.so python - > gpurun_out/exp22/trace.txt 2>&1 <<'PY'
import numpy as np, torch, bench
N, B = 100, 4
x0, z0h, lbh, ubh, T = bench.build_workload(4096, N, seed=2019)
sel = [0, 1, 3319, 2929]
z0h, lbh, ubh = z0h[sel], lbh[sel], ubh[sel]
eng = bench.DeviceEngine(N, T, 0, B)
dev = torch.device("cuda", 0); f64 = dict(dtype=torch.float64, device=dev)
z = torch.from_numpy(np.ascontiguousarray(z0h)).to(dev); lb = torch.from_numpy(np.ascontiguousarray(lbh)).to(dev); ub = torch.from_numpy(np.ascontiguousarray(ubh)).to(dev)
lam = torch.empty(B, eng.m, **f64); kkt = torch.empty(B, 3, **f64); cost = torch.empty(B, **f64)
st = torch.empty(B, dtype=torch.int32, device=dev); it = torch.empty(B, dtype=torch.int32, device=dev)
import os; os.environ["MYRIAD_FUSED_WAVES"] = "1"
eng.solve(B, z, lb, ub, lam, cost, st, it, kkt); torch.cuda.synchronize()
print("iters", it.cpu().numpy(), st.cpu().numpy())
PY
grep -c . gpurun_out/exp22/trace.txt
