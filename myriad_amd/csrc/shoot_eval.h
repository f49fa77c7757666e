// shoot_eval.h -- shooting transcription evaluation (K3/K4 in DESIGN.md): constraints, their Jacobian as interval
// blocks, objective and its full gradient.
//
// Replaces jit(constraints), jit(jacrev(constraints)), jit(objective), jit(grad(objective))
// (/root/reference/myriad/nlp_solvers/__init__.py:32-40) for /root/reference/myriad/trajectory_optimizers/shooting.py:
//   constraints :230-241 (integrate_time_independent_in_parallel over the intervals, utils.py:80-134)
//   objective   :169-210 (augmented state [x; integral of g]; the reference integrates twice, here once).
// One trajectory per lane (Euler / Heun steps).  Per interval: forward rollout (states parked in scratch), then ONE
// reverse sweep carrying the ns x ns adjoint of the end state and the adjoint of the running cost -- reverse mode
// like the reference's jacrev, but all ns+1 rows at once.
// Flop/latency bound (2*cpi dependent dynamics evaluations per interval); algorithmic bytes are a few KB/instance.
#pragma once
#include <hip/hip_runtime.h>
#include "os_solver.h"

namespace myriad {

// jblk layout per interval k: Jx = d c_k / d x_k (ns x ns, row-major), then Ju = d c_k / d u_{k*cpi .. (k+1)*cpi}
// (ns x (cpi+1)*nu, row-major); d c_k / d x_{k+1} = -I is implied.
template <class Sys, int M = 1>
__global__ __launch_bounds__(64)
void shoot_eval_kernel(int B, int I, int cpi, int method, double T, const double* __restrict__ z,
                       const double* __restrict__ params, int params_stride, double* __restrict__ fout,
                       double* __restrict__ gout, double* __restrict__ cout, double* __restrict__ jout,
                       double* __restrict__ scratch, const double* __restrict__ lamin = nullptr, int with_cost = 1) {
  // M control rows per step (1: Euler / Heun / midpoint, 2: RK4 with u[2i], u[2i+1], u[2i+2]).
  // lamin [B][I*NS] (optional): gout becomes (with_cost ? grad f : 0) + J^T lam -- the gradient of the Lagrangian in z
  // (nlp_solvers/extra_gradient.py:21-33) by the same reverse sweep, seeded with lam_k at the end of interval k
  using SC = ShootCore<Sys, M>;
  constexpr int NS = Sys::NS, NU = Sys::NU, NW = Sys::NW, NY = SC::NY;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int S = I * cpi;
  const int W = M * cpi + 1;                                   // control rows of one interval
  const long n = (long)(I + 1) * NS + (long)(M * S + 1) * NU;
  const double h = T / S;
  const double* zb = z + b * n;
  const double* ub = zb + (long)(I + 1) * NS;
  SysParams<Sys> pp;
  pp.load(params, b, params_stride);
  const double* p = pp.get();
  double* xs = scratch + b * (long)(cpi + 1) * NS;      // states of the current interval
  const long jpi = (long)NS * NS + (long)NS * W * NU;
  double* gb = gout ? gout + b * n : nullptr;
  if (gb) for (long i = 0; i < n; ++i) gb[i] = 0.0;
  double ftot = 0.0;
  for (int k = 0; k < I; ++k) {
    double x[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) x[c] = zb[(long)k * NS + c];
    for (int j = 0; j < cpi; ++j) {
      const int i = k * cpi + j;
      double xn[NS], dc;
#pragma unroll
      for (int c = 0; c < NS; ++c) xs[(long)j * NS + c] = x[c];
      SC::sval(method, h, x, ub + (long)M * i * NU, p, xn, dc, h * i, i == S - 1);
      ftot += dc;
#pragma unroll
      for (int c = 0; c < NS; ++c) x[c] = xn[c];
    }
    if (cout) {
#pragma unroll
      for (int c = 0; c < NS; ++c) cout[b * (long)I * NS + (long)k * NS + c] = x[c] - zb[(long)(k + 1) * NS + c];   // shooting.py:239-241
    }
    if (!jout && !gb) continue;
    const double* lk = lamin ? lamin + b * (long)I * NS + (long)k * NS : nullptr;
    const double cw = with_cost ? 1.0 : 0.0;
    // reverse sweep: Lam = d x_end / d x_{j+1} (ns x ns), a = d (cost of the rest of the interval [+ lam_k^T x_end]) / d x_{j+1};
    // every step adds its direct contribution to each of its M+1 control rows (the rows shared by two steps get two)
    double Lam[NS * NS], a[NS];
#pragma unroll
    for (int r = 0; r < NS; ++r)
#pragma unroll
      for (int c = 0; c < NS; ++c) Lam[r * NS + c] = (r == c) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      a[c] = lk ? lk[c] : 0.0;                                   // d (lam_k^T c_k) / d x_end = lam_k
      if (lk && gb) gb[(long)(k + 1) * NS + c] -= lk[c];         // d (lam_k^T c_k) / d x_{k+1} = -lam_k
    }
    double* jb = jout ? jout + b * (long)I * jpi + (long)k * jpi : nullptr;
    if (jb) for (long q = 0; q < (long)NS * W * NU; ++q) jb[NS * NS + q] = 0.0;
    const double zero[NS] = {0};
    for (int j = cpi - 1; j >= 0; --j) {
      const int i = k * cpi + j;
      double xi[NS], Fy[NS * NY], gy[NY], Hs[NY * NY];
#pragma unroll
      for (int c = 0; c < NS; ++c) xi[c] = xs[(long)j * NS + c];
      SC::slin(method, h, xi, ub + (long)M * i * NU, p, zero, Fy, gy, Hs, h * i, i == S - 1);
      // direct contributions to the step's control rows M i + s, s = 0 .. M (local rows M j + s of the interval)
#pragma unroll
      for (int sl = 0; sl <= M; ++sl)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          const int col = NS + sl * NU + u;
          double gsum = cw * gy[col];
#pragma unroll
          for (int t = 0; t < NS; ++t) gsum += a[t] * Fy[t * NY + col];
          if (gb) gb[(long)(I + 1) * NS + (long)(M * i + sl) * NU + u] += gsum;
          if (jb) {
#pragma unroll
            for (int r = 0; r < NS; ++r) {
              double sj = 0.0;
#pragma unroll
              for (int t = 0; t < NS; ++t) sj += Lam[r * NS + t] * Fy[t * NY + col];
              jb[NS * NS + (long)r * W * NU + (long)(M * j + sl) * NU + u] += sj;
            }
          }
        }
      // propagate the adjoints through the step
      double nL[NS * NS], na[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        double sg = cw * gy[c];
#pragma unroll
        for (int t = 0; t < NS; ++t) sg += a[t] * Fy[t * NY + c];
        na[c] = sg;
#pragma unroll
        for (int r = 0; r < NS; ++r) {
          double v = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) v += Lam[r * NS + t] * Fy[t * NY + c];
          nL[r * NS + c] = v;
        }
      }
#pragma unroll
      for (int q = 0; q < NS * NS; ++q) Lam[q] = nL[q];
#pragma unroll
      for (int c = 0; c < NS; ++c) a[c] = na[c];
    }
    // the node state
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      if (gb) gb[(long)k * NS + c] += a[c];
#pragma unroll
      for (int r = 0; r < NS; ++r) if (jb) jb[r * NS + c] = Lam[r * NS + c];
    }
  }
  if (fout) fout[b] = ftot;
}

// out[b] = J(z_b) v_b for the shooting constraints c_k = x_end(x_k, u_{k cpi .. (k+1) cpi}) - x_{k+1}: forward tangent
// propagation through the steps of each interval (one trajectory per lane)
template <class Sys, int M = 1>
__global__ __launch_bounds__(64)
void shoot_jvp_kernel(int B, int I, int cpi, int method, double T, const double* __restrict__ z, const double* __restrict__ v,
                      const double* __restrict__ params, int params_stride, double* __restrict__ out) {
  using SC = ShootCore<Sys, M>;
  constexpr int NS = Sys::NS, NU = Sys::NU, NY = SC::NY;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int S = I * cpi;
  const long n = (long)(I + 1) * NS + (long)(M * S + 1) * NU;
  const double h = T / S;
  const double* zb = z + b * n; const double* vb = v + b * n;
  const double* ub = zb + (long)(I + 1) * NS; const double* vu = vb + (long)(I + 1) * NS;
  SysParams<Sys> pp;
  pp.load(params, b, params_stride);
  const double* p = pp.get();
  const double zero[NS] = {0};
  for (int k = 0; k < I; ++k) {
    double x[NS], dx[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) { x[c] = zb[(long)k * NS + c]; dx[c] = vb[(long)k * NS + c]; }
    for (int j = 0; j < cpi; ++j) {
      const int i = k * cpi + j;
      double xn[NS], dc, Fy[NS * NY], gy[NY], Hs[NY * NY], dn[NS];
      SC::slin(method, h, x, ub + (long)M * i * NU, p, zero, Fy, gy, Hs, h * i, false);
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < NS; ++c) s += Fy[r * NY + c] * dx[c];
#pragma unroll
        for (int a = 0; a < (M + 1) * NU; ++a) s += Fy[r * NY + NS + a] * vu[(long)M * i * NU + a];
        dn[r] = s;
      }
      SC::sval(method, h, x, ub + (long)M * i * NU, p, xn, dc, h * i, false);
#pragma unroll
      for (int c = 0; c < NS; ++c) { x[c] = xn[c]; dx[c] = dn[c]; }
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) out[b * (long)I * NS + (long)k * NS + c] = dx[c] - vb[(long)(k + 1) * NS + c];
  }
}

// z <- clip(zref - eta g, lb, ub) (out may alias zref); lam += eta_v c
static __global__ __launch_bounds__(256)
void exgd_update_kernel(long total, const double* __restrict__ zref, const double* __restrict__ g, const double* __restrict__ lb,
                        const double* __restrict__ ub, double eta, double* out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const double v = zref[i] - eta * g[i];
    out[i] = v < lb[i] ? lb[i] : (v > ub[i] ? ub[i] : v);
  }
}
static __global__ __launch_bounds__(256)
void axpy_kernel(long total, double a, const double* __restrict__ x, double* y) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}

}  // namespace myriad
