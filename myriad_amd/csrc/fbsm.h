// fbsm.h -- batched Forward-Backward Sweep: the reference's indirect solver
// (/root/reference/myriad/trajectory_optimizers/forward_backward_sweep.py:20-116, RK4 sweeps of
// myriad/utils.py:138-197, stopping rule of trajectory_optimizers/base.py:128-141) for many instances at once --
// parameter / start-state sweeps of one IndirectFHCS system.
//
// One lane per instance: the three trajectories x, u, adj (N+1 points each) live in global scratch, batch-minor
// (point i of instance b at a[i*Bp + b]), so the 64 lanes of a wavefront stream 64 consecutive doubles per step.  A sweep is
//   x   <- RK4 forward  of x' = f(x, u)                 from x(0) = x_0                        (:95-96)
//   adj <- RK4 backward of adj' = adj_ODE(adj, x, u)    from adj(T) = adj_T                    (:97-98)
//   u   <- (optim_characterization(adj, x) + u) / 2                                            (:100-102)
// with the half-step costates averaged between the two grid points (utils.py:166-175), repeated while
//   min_v ( delta * sum_t |v| - sum_t |v - v_old| ) < 0  over v in {u, x, adj} and their components (base.py:137-141).
// The sums are accumulated while the new values overwrite the old ones, so no copies of the previous iterate exist.
// Systems with terminal state conditions (the reference's secant `sequencesolver`, :118-158) are not on this path.
#pragma once
#include <hip/hip_runtime.h>
#include "systems_gen.h"

namespace myriad {

// adjoint ODE and optimality characterisation per system (the IndirectFHCS members of the reference)
template <class Sys> struct Indirect { static constexpr bool SUPPORTED = false; };

// myriad/systems/lenhart/simple_case.py:55-62 (maximisation-convention adjoint; the characterisation is clipped with
// bounds[0], the state row -- reference quirk kept: the host passes those bounds)
template <> struct Indirect<SysSIMPLECASE> {
  static constexpr bool SUPPORTED = true;
  __device__ static inline void adj_ode(const double* adj, const double* x, const double* u, const double* p, double* out) {
    (void)u;
    out[0] = -p[0] + x[0] * adj[0];
  }
  __device__ static inline void characterize(const double* adj, const double* x, const double* p, double* u) {
    (void)x;
    u[0] = (p[2] * adj[0]) / (2.0 * p[1]);
  }
};
// myriad/systems/lenhart/cancer_treatment.py:84-91
template <> struct Indirect<SysCANCERTREATMENT> {
  static constexpr bool SUPPORTED = true;
  __device__ static inline void adj_ode(const double* adj, const double* x, const double* u, const double* p, double* out) {
    out[0] = adj[0] * (p[0] + p[2] * u[0] - p[0] * log(1.0 / x[0])) - 2.0 * p[1] * x[0];
  }
  __device__ static inline void characterize(const double* adj, const double* x, const double* p, double* u) {
    u[0] = 0.5 * adj[0] * p[2] * x[0];
  }
};

template <class Sys>
__global__ __launch_bounds__(64)
void fbsm_kernel(int B, long Bp, int N, double T, const double* __restrict__ x0, const double* __restrict__ adjT,
                 const double* __restrict__ params, int params_stride, double clip_lo, double clip_hi, double delta,
                 int max_sweeps, double* xs, double* us, double* adjs, int32_t* sweeps) {
  constexpr int NS = Sys::NS, NU = Sys::NU;
  using I = Indirect<Sys>;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  SysParams<Sys> pp;
  pp.load(params, b, params_stride);
  const double* p = pp.get();
  const double h = T / N;
  double* X = xs + b; double* U = us + b; double* A = adjs + b;
  auto at = [Bp](double* a, int i, int c, int nc) -> double& { return a[((long)i * nc + c) * Bp]; };
  // guesses (:38-47): x = (x_0, 0...), u = 0, adj = (0..., adj_T)
  for (int i = 0; i <= N; ++i) {
#pragma unroll
    for (int c = 0; c < NS; ++c) { at(X, i, c, NS) = (i == 0) ? x0[b * NS + c] : 0.0; at(A, i, c, NS) = (i == N && adjT) ? adjT[c] : 0.0; }
#pragma unroll
    for (int c = 0; c < NU; ++c) at(U, i, c, NU) = 0.0;
  }
  int n = 0;
  bool go = true;
  while (go && n < max_sweeps) {
    double sx[NS], dx[NS], sa[NS], da[NS], su[NU], du[NU];
#pragma unroll
    for (int c = 0; c < NS; ++c) { sx[c] = dx[c] = sa[c] = da[c] = 0.0; }
#pragma unroll
    for (int c = 0; c < NU; ++c) { su[c] = du[c] = 0.0; }
    // ---- forward sweep ----
    double x[NS], u0[NU], u1[NU];
#pragma unroll
    for (int c = 0; c < NS; ++c) { x[c] = at(X, 0, c, NS); sx[c] += fabs(x[c]); }
#pragma unroll
    for (int c = 0; c < NU; ++c) u0[c] = at(U, 0, c, NU);
    for (int i = 0; i < N; ++i) {
      double um[NU], k1[NS], k2[NS], k3[NS], k4[NS], xt[NS];
#pragma unroll
      for (int c = 0; c < NU; ++c) { u1[c] = at(U, i + 1, c, NU); um[c] = (u0[c] + u1[c]) / 2; }
      Sys::f(x, u0, p, k1);
#pragma unroll
      for (int c = 0; c < NS; ++c) xt[c] = x[c] + h * k1[c] / 2;
      Sys::f(xt, um, p, k2);
#pragma unroll
      for (int c = 0; c < NS; ++c) xt[c] = x[c] + h * k2[c] / 2;
      Sys::f(xt, um, p, k3);
#pragma unroll
      for (int c = 0; c < NS; ++c) xt[c] = x[c] + h * k3[c];
      Sys::f(xt, u1, p, k4);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        x[c] = x[c] + (h / 6) * (k1[c] + 2 * k2[c] + 2 * k3[c] + k4[c]);
        const double old = at(X, i + 1, c, NS);
        sx[c] += fabs(x[c]); dx[c] += fabs(x[c] - old);
        at(X, i + 1, c, NS) = x[c];
      }
#pragma unroll
      for (int c = 0; c < NU; ++c) u0[c] = u1[c];
    }
    // ---- backward sweep (step -h), then the control update point by point (it needs only adj_i, x_i, u_i) ----
    double a[NS], xi[NS], xj[NS], ui[NU], uj[NU];
#pragma unroll
    for (int c = 0; c < NS; ++c) { a[c] = at(A, N, c, NS); sa[c] += fabs(a[c]); xi[c] = at(X, N, c, NS); }
#pragma unroll
    for (int c = 0; c < NU; ++c) ui[c] = at(U, N, c, NU);
    for (int i = N; i >= 0; --i) {
      // control update at point i from the NEW adj_i, x_i and the OLD u_i
      {
        double ue[NU];
        I::characterize(a, xi, p, ue);
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          const double est = fmin(clip_hi, fmax(clip_lo, ue[c]));
          const double nu_ = 0.5 * (est + ui[c]);
          su[c] += fabs(nu_); du[c] += fabs(nu_ - ui[c]);
          at(U, i, c, NU) = nu_;
        }
      }
      if (i == 0) break;
#pragma unroll
      for (int c = 0; c < NS; ++c) xj[c] = at(X, i - 1, c, NS);
#pragma unroll
      for (int c = 0; c < NU; ++c) uj[c] = at(U, i - 1, c, NU);
      double xm[NS], um[NU], k1[NS], k2[NS], k3[NS], k4[NS], t[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) xm[c] = (xi[c] + xj[c]) / 2;
#pragma unroll
      for (int c = 0; c < NU; ++c) um[c] = (ui[c] + uj[c]) / 2;
      const double hm = -h;
      I::adj_ode(a, xi, ui, p, k1);
#pragma unroll
      for (int c = 0; c < NS; ++c) t[c] = a[c] + hm * k1[c] / 2;
      I::adj_ode(t, xm, um, p, k2);
#pragma unroll
      for (int c = 0; c < NS; ++c) t[c] = a[c] + hm * k2[c] / 2;
      I::adj_ode(t, xm, um, p, k3);
#pragma unroll
      for (int c = 0; c < NS; ++c) t[c] = a[c] + hm * k3[c];
      I::adj_ode(t, xj, uj, p, k4);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        a[c] = a[c] + (hm / 6) * (k1[c] + 2 * k2[c] + 2 * k3[c] + k4[c]);
        const double old = at(A, i - 1, c, NS);
        sa[c] += fabs(a[c]); da[c] += fabs(a[c] - old);
        at(A, i - 1, c, NS) = a[c];
        xi[c] = xj[c];
      }
#pragma unroll
      for (int c = 0; c < NU; ++c) ui[c] = uj[c];
    }
    ++n;
    double mn = INFINITY;
#pragma unroll
    for (int c = 0; c < NU; ++c) mn = fmin(mn, su[c] * delta - du[c]);
#pragma unroll
    for (int c = 0; c < NS; ++c) { mn = fmin(mn, sx[c] * delta - dx[c]); mn = fmin(mn, sa[c] * delta - da[c]); }
    go = mn < 0.0;
  }
  if (sweeps) sweeps[b] = n;
}

}  // namespace myriad
