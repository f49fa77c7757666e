// colloc_products.h -- matrix-free Lagrangian products of the collocation transcriptions and the fused
// extragradient step built from them.
//
// Reference semantics (nikihowe/myriad, /root/reference/myriad/):
//   lagrangian(x, lmbda) = fun(x) + lmbda @ constraint_fun(x)                 nlp_solvers/extra_gradient.py:21-23,
//                                                                              experiments/e2e_sysid.py:113-116
//   step(x, lmbda): x_bar = clip(x - eta_x dL/dx(x, lmbda)); x_new = clip(x - eta_x dL/dx(x_bar, lmbda));
//                   lmbda_new = lmbda + eta_v dL/dlmbda(x_new, lmbda)          extra_gradient.py:25-33, e2e_sysid.py:118-125
// The reference gets dL/dx from jax.grad through the dense transcription; here J^T lam and J v are applied
// block-wise from the dynamics Jacobians A = df/dx, B = df/du of each collocation point and never formed:
//
//   Hermite-Simpson (hermite_simpson.py:110-128, :153-170), point j, intervals kL = j/2 - 1 (j is its end) and
//   kR = j/2 (j is its start) for a knot, k = (j-1)/2 for a midpoint, lam = [lam_d ; lam_i]:
//     knot:      a = -h/6 (lam_d[kL] + lam_d[kR]) + h/8 (lam_i[kL] - lam_i[kR])      e = lam_d[kL] - lam_d[kR] - (lam_i[kL] + lam_i[kR]) / 2
//     midpoint:  a = -4h/6 lam_d[k]                                                   e = lam_i[k]
//     (J^T lam)_x = A^T a + e,  (J^T lam)_u = B^T a
//   Trapezoidal (trapezoidal.py:151-163; defect sign opposite to HS), point j, intervals j-1 and j:
//     a = h/2 (lam[j-1] + lam[j]),  e = lam[j] - lam[j-1]
//
// Both products are pointwise / intervalwise maps with no coupling beyond a point's two neighbouring intervals, so
// the kernels are plain grid-stride loops: one thread per (instance, point) for J^T lam, one per (instance, interval)
// for J v, reading z / lam / v in place (instance-major, the reference's ravel_pytree layout).  HBM-bound by
// construction: J^T lam moves 8 (2n + m) bytes per instance.
#pragma once
#include <hip/hip_runtime.h>
#include "systems_gen.h"

namespace myriad {

enum { PROD_HS = 0, PROD_TRAP = 1 };

template <class Sys, int SCHEME>
struct CollocProducts {
  static constexpr int NS = Sys::NS, NU = Sys::NU, NW = Sys::NW;
  __host__ __device__ static inline int points(int N) { return SCHEME == PROD_HS ? 2 * N + 1 : N + 1; }
  __host__ __device__ static inline int rows(int N) { return (SCHEME == PROD_HS ? 2 : 1) * N * NS; }
  // quadrature weight of point j in the objective (hermite_simpson.py:194-214 Simpson, trapezoidal.py:80-94)
  __device__ static inline double wq(int K, int j, double h) {
    if (SCHEME == PROD_HS) return (j & 1) ? 4.0 * h / 6.0 : ((j == 0 || j == K - 1) ? h / 6.0 : 2.0 * h / 6.0);
    return (j == 0 || j == K - 1) ? 0.5 * h : h;
  }
  // multiplier combination of point j: a (applied through A^T, B^T) and e (the identity blocks)
  __device__ static inline void combine(const double* lam, int N, int j, double h, double* a, double* e) {
    if (SCHEME == PROD_HS) {
      const double h6 = h / 6.0, h8 = h / 8.0;
      const double* ld = lam;
      const double* li = lam + (long)N * NS;
      if (j & 1) {
        const int k = (j - 1) >> 1;
#pragma unroll
        for (int q = 0; q < NS; ++q) { a[q] = -4.0 * h6 * ld[k * NS + q]; e[q] = li[k * NS + q]; }
      } else {
        const int kL = (j >> 1) - 1, kR = j >> 1;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const double dl = kL >= 0 ? ld[kL * NS + q] : 0.0, il = kL >= 0 ? li[kL * NS + q] : 0.0;
          const double dr = kR < N ? ld[kR * NS + q] : 0.0, ir = kR < N ? li[kR * NS + q] : 0.0;
          a[q] = -h6 * (dl + dr) + h8 * (il - ir);
          e[q] = dl - dr - 0.5 * (il + ir);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const double l0 = j >= 1 ? lam[(j - 1) * NS + q] : 0.0, l1 = j < N ? lam[j * NS + q] : 0.0;
        a[q] = 0.5 * h * (l0 + l1);
        e[q] = l1 - l0;
      }
    }
  }
  // (grad f +) J^T lam restricted to point j, from the point's own (x, u)
  __device__ static inline void vjp_point(const double* x, const double* u, const double* p, const double* a, const double* e,
                                          double wj, bool add_gradf, double* gx, double* gu, bool last = false, double t = 0.0) {
    double f[NS], A[NS * NS], Bm[NS * NU], g, gw[NW];
    set_time<Sys>(p, t);
    Sys::lin(x, u, p, f, A, Bm, &g, gw);
    if (SCHEME == PROD_TRAP && last) fold_terminal<Sys>(x, u, p, wj, g, gw);     // trapezoidal.py:126-127
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      double s = e[q] + (add_gradf ? wj * gw[q] : 0.0);
#pragma unroll
      for (int t = 0; t < NS; ++t) s += A[t * NS + q] * a[t];
      gx[q] = s;
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) {
      double s = add_gradf ? wj * gw[NS + c] : 0.0;
#pragma unroll
      for (int t = 0; t < NS; ++t) s += Bm[t * NU + c] * a[t];
      gu[c] = s;
    }
  }
  // directional derivative of the dynamics at a point: w = A vx + B vu
  __device__ static inline void dfdir(const double* x, const double* u, const double* p, const double* vx, const double* vu, double* w) {
    double f[NS], A[NS * NS], Bm[NS * NU], g, gw[NW];
    Sys::lin(x, u, p, f, A, Bm, &g, gw);
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < NS; ++q) s += A[r * NS + q] * vx[q];
#pragma unroll
      for (int c = 0; c < NU; ++c) s += Bm[r * NU + c] * vu[c];
      w[r] = s;
    }
  }
};

// out[b] = J(z_b)^T lam_b (+ grad f(z_b) when add_gradf: the gradient of the Lagrangian in z)
template <class Sys, int SCHEME>
__global__ __launch_bounds__(256)
void colloc_vjp_kernel(int B, int N, double h, const double* __restrict__ z, const double* __restrict__ lam,
                       const double* __restrict__ params, int params_stride, double* __restrict__ out, int add_gradf) {
  using P = CollocProducts<Sys, SCHEME>;
  constexpr int NS = P::NS, NU = P::NU;
  const int K = P::points(N), n = K * (NS + NU), m = P::rows(N);
  const long total = (long)B * K;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    const long b = g / K;
    const int j = (int)(g - b * K);
    const double* zb = z + b * n;
    SysParams<Sys> pp;
    pp.load(params, b, params_stride);
    double x[NS], u[NU], a[NS], e[NS], gx[NS], gu[NU];
#pragma unroll
    for (int q = 0; q < NS; ++q) x[q] = zb[(long)j * NS + q];
#pragma unroll
    for (int c = 0; c < NU; ++c) u[c] = zb[(long)K * NS + (long)j * NU + c];
    P::combine(lam + b * m, N, j, h, a, e);
    P::vjp_point(x, u, pp.get(), a, e, P::wq(K, j, h), add_gradf != 0, gx, gu, j == K - 1, (SCHEME == PROD_HS ? 0.5 * h : h) * j);
    double* ob = out + b * n;
#pragma unroll
    for (int q = 0; q < NS; ++q) ob[(long)j * NS + q] = gx[q];
#pragma unroll
    for (int c = 0; c < NU; ++c) ob[(long)K * NS + (long)j * NU + c] = gu[c];
  }
}

// out[b] = J(z_b) v_b, rows in the reference's constraint order
template <class Sys, int SCHEME>
__global__ __launch_bounds__(256)
void colloc_jvp_kernel(int B, int N, double h, const double* __restrict__ z, const double* __restrict__ v,
                       const double* __restrict__ params, int params_stride, double* __restrict__ out) {
  using P = CollocProducts<Sys, SCHEME>;
  constexpr int NS = P::NS, NU = P::NU;
  const int K = P::points(N), n = K * (NS + NU), m = P::rows(N);
  const long total = (long)B * N;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    const long b = g / N;
    const int k = (int)(g - b * N);
    const double* zb = z + b * n;
    const double* vb = v + b * n;
    SysParams<Sys> pp;
    pp.load(params, b, params_stride);
    const double* p = pp.get();
    constexpr int NP = SCHEME == PROD_HS ? 3 : 2;
    double vx[NP][NS], w[NP][NS];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int j = (SCHEME == PROD_HS ? 2 * k : k) + i;
      double x[NS], u[NU], vu[NU];
#pragma unroll
      for (int q = 0; q < NS; ++q) { x[q] = zb[(long)j * NS + q]; vx[i][q] = vb[(long)j * NS + q]; }
#pragma unroll
      for (int c = 0; c < NU; ++c) { u[c] = zb[(long)K * NS + (long)j * NU + c]; vu[c] = vb[(long)K * NS + (long)j * NU + c]; }
      P::dfdir(x, u, p, vx[i], vu, w[i]);
    }
    double* ob = out + b * m;
    if (SCHEME == PROD_HS) {
      const double h6 = h / 6.0, h8 = h / 8.0;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        ob[(long)k * NS + q] = (vx[2][q] - vx[0][q]) - h6 * (w[0][q] + 4.0 * w[1][q] + w[2][q]);
        ob[(long)N * NS + (long)k * NS + q] = vx[1][q] - 0.5 * (vx[0][q] + vx[2][q]) - h8 * (w[0][q] - w[2][q]);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NS; ++q) ob[(long)k * NS + q] = 0.5 * h * (w[0][q] + w[1][q]) - (vx[1][q] - vx[0][q]);
    }
  }
}

// Fused extragradient iterations, one wavefront per instance, z and lam resident in LDS for `nsteps` steps
// (extra_gradient.py:25-33).  Dynamic LDS: (2 n + m + K NS) doubles.
template <class Sys, int SCHEME>
__global__ __launch_bounds__(64)
void colloc_exgd_kernel(int B, int N, double h, double* z, double* lam, const double* __restrict__ lb,
                        const double* __restrict__ ub, const double* __restrict__ params, int params_stride,
                        double eta_x, double eta_v, int nsteps) {
  using P = CollocProducts<Sys, SCHEME>;
  constexpr int NS = P::NS, NU = P::NU;
  extern __shared__ __attribute__((aligned(16))) char smem_exgd[];
  const long b = blockIdx.x;
  if (b >= B) return;
  const int lane = threadIdx.x;
  const int K = P::points(N), n = K * (NS + NU), m = P::rows(N);
  double* sz = reinterpret_cast<double*>(smem_exgd);   // z
  double* sb = sz + n;                                  // x_bar, then x_new
  double* sl = sb + n;                                  // lam
  double* sf = sl + m;                                  // f at the points of x_new
  double* zb = z + b * n;
  double* lamb = lam + b * m;
  const double* lo = lb + b * n;
  const double* hi = ub + b * n;
  SysParams<Sys> pp;
  pp.load(params, b, params_stride);
  const double* p = pp.get();
  for (int i = lane; i < n; i += 64) sz[i] = zb[i];
  for (int i = lane; i < m; i += 64) sl[i] = lamb[i];
  __syncthreads();
  auto clipd = [](double v, double l, double u) { return v < l ? l : (v > u ? u : v); };
  for (int s = 0; s < nsteps; ++s) {
    // Two gradient passes, both stepping FROM z: at z (-> x_bar in sb) and at x_bar (-> x_new, written over z).
    // dL/dz at point j needs only the point's own (x, u) and lam, and a lane keeps the same points in both passes,
    // so the passes need no barrier between them.
    for (int pass = 0; pass < 2; ++pass) {
      const double* src = pass == 0 ? sz : sb;
      double* dst = pass == 0 ? sb : sz;
      for (int j = lane; j < K; j += 64) {
        double x[NS], u[NU], a[NS], e[NS], gx[NS], gu[NU];
#pragma unroll
        for (int q = 0; q < NS; ++q) x[q] = src[j * NS + q];
#pragma unroll
        for (int c = 0; c < NU; ++c) u[c] = src[K * NS + j * NU + c];
        P::combine(sl, N, j, h, a, e);
        P::vjp_point(x, u, p, a, e, P::wq(K, j, h), true, gx, gu, j == K - 1, (SCHEME == PROD_HS ? 0.5 * h : h) * j);
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const int i = j * NS + q;
          dst[i] = clipd(sz[i] - eta_x * gx[q], lo[i], hi[i]);
        }
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          const int i = K * NS + j * NU + c;
          dst[i] = clipd(sz[i] - eta_x * gu[c], lo[i], hi[i]);
        }
      }
    }
    // f at the new points; then lam += eta_v c(x_new) (lanes over intervals read their neighbours' x, f)
    for (int j = lane; j < K; j += 64) {
      double x[NS], u[NU], f[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) x[q] = sz[j * NS + q];
#pragma unroll
      for (int c = 0; c < NU; ++c) u[c] = sz[K * NS + j * NU + c];
      Sys::f(x, u, p, f);
#pragma unroll
      for (int q = 0; q < NS; ++q) sf[j * NS + q] = f[q];
    }
    __syncthreads();
    for (int k = lane; k < N; k += 64) {
      if (SCHEME == PROD_HS) {
        const double h6 = h / 6.0, h8 = h / 8.0;
        const int js = 2 * k, jm = js + 1, je = js + 2;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const double d = (sz[je * NS + q] - sz[js * NS + q]) - h6 * (sf[js * NS + q] + 4.0 * sf[jm * NS + q] + sf[je * NS + q]);
          const double it = sz[jm * NS + q] - 0.5 * (sz[js * NS + q] + sz[je * NS + q]) - h8 * (sf[js * NS + q] - sf[je * NS + q]);
          sl[k * NS + q] += eta_v * d;
          sl[N * NS + k * NS + q] += eta_v * it;
        }
      } else {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const double d = 0.5 * h * (sf[k * NS + q] + sf[(k + 1) * NS + q]) - (sz[(k + 1) * NS + q] - sz[k * NS + q]);
          sl[k * NS + q] += eta_v * d;
        }
      }
    }
    __syncthreads();
  }
  for (int i = lane; i < n; i += 64) zb[i] = sz[i];
  for (int i = lane; i < m; i += 64) lamb[i] = sl[i];
}

template <class Sys, int SCHEME>
inline size_t colloc_exgd_lds_bytes(int N) {
  using P = CollocProducts<Sys, SCHEME>;
  const int K = P::points(N), n = K * (Sys::NS + Sys::NU), m = P::rows(N);
  return (size_t)(2 * n + m + K * Sys::NS) * 8 + 16;
}

}  // namespace myriad
