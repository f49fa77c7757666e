// What would a parallel-in-time Riccati sweep cost on this part?  (VERDICT r3, next #2.)
// The associative element of Sarkka & Garcia-Fernandez, "Temporal parallelization of dynamic programming and linear quadratic control"
// (IEEE TAC 2023): a = (A, b, C, eta, J), combination (i earlier, j later)
//     M = (I + C_i J_j)^-1,   A_ij = A_j M A_i,   b_ij = A_j M (b_i + C_i eta_j) + b_j,   C_ij = A_j M C_i A_j' + C_j,
//     eta_ij = A_i' M' (eta_j - J_j b_i) + eta_i,   J_ij = A_i' M' J_j A_i + J_i.
// For the Hermite-Simpson stage problem of this solver the state between stages is (dx, du_knot): n = NS + NU = 5 for CARTPOLE (the knot
// control is shared by two intervals and cannot be eliminated inside one), and the sweep carries NC = 2 + NS = 6 right-hand-side columns
// (gradient, barrier parameter, NS terminal multipliers): b and eta are n x 6.  One element = 25 + 30 + 15 + 30 + 15 = 115 doubles.
// This probe runs the Hillis-Steele scan over the 64 lanes of a wavefront (6 rounds; N = 100 stages would need a seventh plus a pre-combination
// of two stages per lane) and reports shader-clock cycles per combination round, next to the compiler's register / scratch figures
// (hipcc -Rpass-analysis=kernel-resource-usage).  Template parameters: N_ = state dimension, R_ = right-hand-side columns.
//   hipcc --offload-arch=gfx950 -O3 riccati_scan_probe.hip -o riccati_scan_probe ; ./riccati_scan_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N_, int R_>
struct Elem { double A[N_ * N_], b[N_ * R_], C[N_ * N_], eta[N_ * R_], J[N_ * N_]; };     // (C, J kept full here: the symmetric halves are free to the compiler)

template <int N_, int R_>
__device__ inline void combine(Elem<N_, R_>& ai, const Elem<N_, R_>& aj) {      // ai <- ai (earlier) combined with aj (later)
  constexpr int n = N_, r = R_;
  double W[n * n];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int k = 0; k < n; ++k) s += ai.C[i * n + k] * aj.J[k * n + j]; W[i * n + j] = s; }
  // LU of W (no pivoting: I + C J with C, J positive semi-definite)
  for (int k = 0; k < n; ++k) { const double ip = 1.0 / W[k * n + k]; for (int i = k + 1; i < n; ++i) { const double f = W[i * n + k] * ip; W[i * n + k] = f; for (int j = k + 1; j < n; ++j) W[i * n + j] -= f * W[k * n + j]; } }
  auto solve = [&](double* x, int cols) {      // x <- W^-1 x, x is n x cols
    for (int c = 0; c < cols; ++c) {
      for (int i = 1; i < n; ++i) for (int k = 0; k < i; ++k) x[i * cols + c] -= W[i * n + k] * x[k * cols + c];
      for (int i = n - 1; i >= 0; --i) { for (int k = i + 1; k < n; ++k) x[i * cols + c] -= W[i * n + k] * x[k * cols + c]; x[i * cols + c] /= W[i * n + i]; }
    }
  };
  double MA[n * n], MC[n * n], Mb[n * r];
  for (int q = 0; q < n * n; ++q) { MA[q] = ai.A[q]; MC[q] = ai.C[q]; }
  for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) { double s = ai.b[i * r + c]; for (int k = 0; k < n; ++k) s += ai.C[i * n + k] * aj.eta[k * r + c]; Mb[i * r + c] = s; }
  solve(MA, n); solve(MC, n); solve(Mb, r);
  // eta, J first (they need the OLD A_i, b_i, C_i):  M' v = v - J_j M C_i v
  double v[n * r], t[n * r];
  for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) { double s = aj.eta[i * r + c]; for (int k = 0; k < n; ++k) s -= aj.J[i * n + k] * ai.b[k * r + c]; v[i * r + c] = s; }
  for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) { double s = 0; for (int k = 0; k < n; ++k) s += ai.C[i * n + k] * v[k * r + c]; t[i * r + c] = s; }
  solve(t, r);
  for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) { double s = v[i * r + c]; for (int k = 0; k < n; ++k) s -= aj.J[i * n + k] * t[k * r + c]; v[i * r + c] = s; }
  double neta[n * r], JMA[n * n], nJ[n * n];
  for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) { double s = ai.eta[i * r + c]; for (int k = 0; k < n; ++k) s += ai.A[k * n + i] * v[k * r + c]; neta[i * r + c] = s; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += aj.J[i * n + k] * MA[k * n + j]; JMA[i * n + j] = s; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = ai.J[i * n + j]; for (int k = 0; k < n; ++k) s += ai.A[k * n + i] * JMA[k * n + j]; nJ[i * n + j] = s; }
  // A, b, C
  double nA[n * n], nb[n * r], T[n * n], nC[n * n];
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += aj.A[i * n + k] * MA[k * n + j]; nA[i * n + j] = s; }
  for (int i = 0; i < n; ++i) for (int c = 0; c < r; ++c) { double s = aj.b[i * r + c]; for (int k = 0; k < n; ++k) s += aj.A[i * n + k] * Mb[k * r + c]; nb[i * r + c] = s; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += aj.A[i * n + k] * MC[k * n + j]; T[i * n + j] = s; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = aj.C[i * n + j]; for (int k = 0; k < n; ++k) s += T[i * n + k] * aj.A[j * n + k]; nC[i * n + j] = s; }
  for (int q = 0; q < n * n; ++q) { ai.A[q] = nA[q]; ai.C[q] = nC[q]; ai.J[q] = nJ[q]; }
  for (int q = 0; q < n * r; ++q) { ai.b[q] = nb[q]; ai.eta[q] = neta[q]; }
}

template <int N_, int R_>
__global__ __launch_bounds__(64, 1) void scan_probe(double* out, long long* cycles, int reps) {
  const int lane = threadIdx.x;
  Elem<N_, R_> e;
  // a well-conditioned pseudo-random element per lane: A near 0.9 I, C and J small positive definite
  unsigned s = 12345u + 977u * lane + 31u * blockIdx.x;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0 - 0.5; };
  for (int i = 0; i < N_; ++i) for (int j = 0; j < N_; ++j) { e.A[i * N_ + j] = (i == j ? 0.9 : 0.0) + 0.05 * rnd(); e.C[i * N_ + j] = (i == j ? 0.2 : 0.0); e.J[i * N_ + j] = (i == j ? 0.3 : 0.0); }
  for (int q = 0; q < N_ * R_; ++q) { e.b[q] = 0.1 * rnd(); e.eta[q] = 0.1 * rnd(); }
  long long t0 = clock64();
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll 1
    for (int d = 1; d < 64; d <<= 1) {
      Elem<N_, R_> o;
      double* pe = reinterpret_cast<double*>(&e); double* po = reinterpret_cast<double*>(&o);
      for (int q = 0; q < (int)(sizeof(e) / 8); ++q) po[q] = __shfl_up(pe[q], d, 64);
      if (lane >= d) { Elem<N_, R_> mine = o; combine<N_, R_>(mine, e); e = mine; }     // earlier elements sit in the lower lanes
    }
  }
  long long t1 = clock64();
  double acc = 0; const double* pe = reinterpret_cast<const double*>(&e);
  for (int q = 0; q < (int)(sizeof(e) / 8); ++q) acc += pe[q];
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int N_, int R_>
static void run(const char* what) {
  double* out; long long* cyc;
  hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 1024 * 8);
  const int reps = 4;
  hipLaunchKernelGGL((scan_probe<N_, R_>), dim3(1024), dim3(64), 0, 0, out, cyc, reps);      // 1024 wavefronts: one per SIMD, as the solver runs
  hipDeviceSynchronize();
  long long h[1024]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 1024; ++i) m += (double)h[i];
  m /= 1024.0 * reps * 6;
  printf("%s: n = %d, rhs columns = %d, %d doubles per element: %.0f shader-clock cycles per combination round (6 rounds per 64 stages)\n", what, N_, R_,
         (int)(sizeof(Elem<N_, R_>) / 8), m);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<5, 6>("this solver (state (dx, du), columns 1, mu, nu_1..4)");
  run<4, 6>("if the knot control could be eliminated stage-wise");
  run<4, 1>("textbook LQ tracking, no extra columns");
  return 0;
}
