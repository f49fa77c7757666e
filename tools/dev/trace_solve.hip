// dev tool: run the device solver for ONE trajectory (default CARTPOLE x0) with per-iteration trace printf.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMYR_TRACE tools/dev/trace_solve.hip -o /tmp/trace_solve
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../../myriad_amd/csrc/hs_solver.h"
using namespace myriad;
using Sys = SysCARTPOLE;
__global__ void k(HsSolveOpts o, double* z, double* lb, double* ub, double* zL, double* zU, double* lam, double* dz, double* st, HsSolveResult* r) {
  double p[Sys::NPX]; Sys::default_params(p);
  HsWork w{{z, 1}, {lb, 1}, {ub, 1}, {zL, 1}, {zU, 1}, {lam, 1}, {dz, 1}, {st, 1}};
  HsSolver<Sys>::solve(w, o, p, *r);
}
int main(int argc, char** argv) {
  int N = argc > 1 ? atoi(argv[1]) : 10;
  int K = 2 * N + 1, n = K * 5, m = 8 * N;
  std::vector<double> z(n, 0.0), lb(n), ub(n);
  double xT[4] = {1.0, M_PI, 0, 0}, bl[5] = {-2, -2 * M_PI, -5, -10, -20};
  for (int j = 0; j < K; ++j) for (int c = 0; c < 4; ++c) { z[j * 4 + c] = xT[c] * j / (K - 1); lb[j * 4 + c] = bl[c]; ub[j * 4 + c] = -bl[c]; }
  for (int j = 0; j < K; ++j) { lb[K * 4 + j] = -20; ub[K * 4 + j] = 20; }
  for (int c = 0; c < 4; ++c) { lb[c] = ub[c] = 0; lb[(K - 1) * 4 + c] = ub[(K - 1) * 4 + c] = xT[c]; }
  long nst = HsSol<Sys>::stage_doubles(N);
  double *dz_, *dlb, *dub, *dzL, *dzU, *dlam, *ddz, *dst; HsSolveResult* dr;
  hipMalloc(&dz_, n * 8); hipMalloc(&dlb, n * 8); hipMalloc(&dub, n * 8); hipMalloc(&dzL, n * 8); hipMalloc(&dzU, n * 8);
  hipMalloc(&dlam, m * 8); hipMalloc(&ddz, n * 8); hipMalloc(&dst, nst * 8); hipMalloc(&dr, sizeof(HsSolveResult));
  hipMemcpy(dz_, z.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dlb, lb.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dub, ub.data(), n * 8, hipMemcpyHostToDevice);
  HsSolveOpts o; o.N = N; o.h = 2.0 / N; o.max_iter = 60; o.tol_feas = 1e-8; o.tol_stat = 1e-6; o.tol_compl = 1e-7; o.mu_init = 0.1;
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, o, dz_, dlb, dub, dzL, dzU, dlam, ddz, dst, dr);
  hipDeviceSynchronize();
  HsSolveResult r; hipMemcpy(&r, dr, sizeof(r), hipMemcpyDeviceToHost);
  printf("status %d iters %d cost %.12f feas %.3e stat %.3e\n", r.status, r.iters, r.cost, r.feas, r.stat);
  return 0;
}
