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
#include "hs_solver.h"   // VarScale (8 doubles by value)

namespace myriad {

// adjoint ODE and (unclipped) optimality characterisation per system -- the IndirectFHCS members of the reference,
// restated from the cited lines.  `t` is the time of the point (only HARVEST / TIMBERHARVEST use it); `bang` is
// max|bounds[-1]| for the bang-bang characterisations (sign(.) * 2 bang + bang, then clipped by the caller).
template <class Sys> struct Indirect { static constexpr bool SUPPORTED = false; };

#define MYR_INDIRECT(SYS, ADJ_BODY, CHAR_BODY)                                                                         \
  template <> struct Indirect<SYS> {                                                                                   \
    static constexpr bool SUPPORTED = true;                                                                            \
    __device__ static inline void adj_ode(const double* adj, const double* x, const double* u, const double* p,        \
                                          double t, double* out) { (void)adj; (void)x; (void)u; (void)p; (void)t; ADJ_BODY }  \
    __device__ static inline void characterize(const double* adj, const double* x, const double* p, double t,           \
                                               double bang, double* u) { (void)adj; (void)x; (void)p; (void)t; (void)bang; CHAR_BODY } \
  };
__device__ inline double myr_sign(double v) { return (v > 0.0) - (v < 0.0); }      // jnp.sign

// lenhart/simple_case.py:55-62 (maximisation-convention adjoint; clipped with bounds[0], the state row -- quirk kept by the host)
MYR_INDIRECT(SysSIMPLECASE, out[0] = -p[0] + x[0] * adj[0];, u[0] = (p[2] * adj[0]) / (2.0 * p[1]);)
// lenhart/cancer_treatment.py:84-91      p = (r, a, delta)
MYR_INDIRECT(SysCANCERTREATMENT, out[0] = adj[0] * (p[0] + p[2] * u[0] - p[0] * log(1.0 / x[0])) - 2.0 * p[1] * x[0];,
             u[0] = 0.5 * adj[0] * p[2] * x[0];)
// lenhart/bacteria.py:88-95              p = (r, A, B, C); adj_T = [C] (:50)
MYR_INDIRECT(SysBACTERIA, out[0] = -adj[0] * (p[0] + p[1] * u[0] + p[2] * u[0] * u[0] * exp(-x[0]));,
             u[0] = adj[0] * p[1] * x[0] / (2.0 * (1.0 + p[2] * adj[0] * exp(-x[0])));)
// lenhart/bear_populations.py:117-142    p = (r, K, m_p, m_f, c_p, c_f); two controls
MYR_INDIRECT(SysBEARPOPULATIONS,
             const double k = p[0] / p[1]; const double k2 = p[0] / (p[1] * p[1]);
             out[0] = adj[0] * (2 * k * x[0] + k2 * p[3] * x[1] * x[1] + u[0] - p[0]) - adj[1] * (2 * k * p[2] * (1 - x[1] / p[1]) * x[0])
                      + adj[2] * (2 * k * (p[2] - 1) * x[0] - k2 * p[3] * x[1] * x[1] - 2 * k2 * p[2] * x[0] * x[1]);
             out[1] = adj[1] * (2 * k * x[1] + k2 * p[2] * x[0] * x[0] + u[1] - p[0]) - adj[0] * (2 * k * p[3] * (1 - x[0] / p[1]) * x[1])
                      + adj[2] * (2 * k * (p[3] - 1) * x[1] - 2 * k2 * p[3] * x[0] * x[1] - k2 * p[2] * x[0] * x[0]);
             out[2] = -1.0;,
             u[0] = adj[0] * x[0] / (2.0 * p[4]); u[1] = adj[1] * x[1] / (2.0 * p[5]);)
// lenhart/bioreactor.py:90-101           p = (K, G, D); bang-bang
MYR_INDIRECT(SysBIOREACTOR, out[0] = -p[0] - p[1] * u[0] * adj[0] + 2.0 * p[2] * x[0] * adj[0];,
             u[0] = myr_sign(-1.0 + p[1] * adj[0] * x[0]) * 2.0 * bang + bang;)
// lenhart/epidemic_seirn.py:97-111       p = (A, b, d, c, e, g, a); the last row's (d - d) factor is the reference's
MYR_INDIRECT(SysEPIDEMICSEIRN,
             out[0] = adj[0] * (p[2] + p[3] * x[2] + u[0]) - adj[1] * p[3] * x[2];
             out[1] = adj[1] * (p[4] + p[2]) - adj[2] * p[4];
             out[2] = -p[0] + adj[0] * p[3] * x[0] - adj[1] * p[3] * x[0] + adj[2] * (p[5] + p[6] + p[2]) + adj[3] * p[6];
             out[3] = -p[1] * adj[0] + adj[3] * (p[2] - p[2]);,
             u[0] = adj[0] * x[0] / 2.0;)
// lenhart/glucose.py:114-126             p = (a, b, c, A, l); the characterisation is NOT clipped by the reference
MYR_INDIRECT(SysGLUCOSE, out[0] = -2.0 * p[3] * (x[0] - p[4]) + adj[0] * p[0]; out[1] = adj[0] * p[1] + adj[1] * p[2];,
             u[0] = -adj[1] / 2.0;)
// lenhart/harvest.py:64-71               p = (A, k, m); explicit time
MYR_INDIRECT(SysHARVEST, out[0] = adj[0] * (p[2] + u[0]) - p[0] * (p[1] * t / (t + 1.0)) * u[0];,
             u[0] = 0.5 * x[0] * (p[0] * (p[1] * t / (t + 1.0)) - adj[0]);)
// lenhart/hiv_treatment.py:117-131       p = (s, m_1, m_2, m_3, r, T_max, k, N, A)
MYR_INDIRECT(SysHIVTREATMENT,
             out[0] = -p[8] + adj[0] * (p[1] - p[4] * (1 - (x[0] + x[1]) / p[5]) + p[4] * x[0] / p[5] + u[0] * p[6] * x[2]) - adj[1] * u[0] * p[6] * x[2];
             out[1] = adj[0] * p[4] * x[0] / p[5] + adj[1] * p[2] - adj[2] * p[7] * p[2];
             out[2] = adj[0] * (p[0] / ((1 + x[2]) * (1 + x[2])) + u[0] * p[6] * x[0]) - adj[1] * u[0] * p[6] * x[0] + adj[2] * p[3];,
             u[0] = 1.0 + 0.5 * p[6] * x[0] * x[2] * (adj[1] - adj[0]);)
// lenhart/mould_fungicide.py:72-79       p = (r, M, A)
MYR_INDIRECT(SysMOULDFUNGICIDE, out[0] = adj[0] * (p[0] + u[0]) - 2.0 * p[2] * x[0];, u[0] = 0.5 * adj[0] * x[0];)
// lenhart/simple_case_with_bounds.py:57-65   p = (A, C)
MYR_INDIRECT(SysSIMPLECASEWITHBOUNDS, out[0] = -p[0] + x[0] * adj[0];, u[0] = (p[1] * adj[0]) / 2.0;)
// lenhart/timber_harvest.py:87-103       p = (r, k); explicit time; bang-bang
MYR_INDIRECT(SysTIMBERHARVEST, out[0] = u[0] * (exp(-p[0] * t) - p[1] * adj[0]) - exp(-p[0] * t);,
             u[0] = myr_sign(x[0] * (p[1] * adj[0] - exp(-p[0] * t))) * 2.0 * bang + bang;)
// lenhart/predator_prey.py:124-137      p = (d_1, d_2, A); adj_T = [1, 0, 0] with the third component set by the secant solver
MYR_INDIRECT(SysPREDATORPREY,
             out[0] = adj[0] * (x[1] - 1 + p[0] * u[0]) - adj[1] * x[1];
             out[1] = adj[0] * x[0] + adj[1] * (1 - x[0] + p[1] * u[0]);
             out[2] = 0.0;,
             u[0] = (adj[0] * p[0] * x[0] + adj[1] * p[1] * x[1] - adj[2]) / p[2];)
#undef MYR_INDIRECT

template <class Sys>
__global__ __launch_bounds__(64)
void fbsm_kernel(int B, long Bp, int N, double T, const double* __restrict__ x0, const double* __restrict__ adjT,
                 const double* __restrict__ params, int params_stride, VarScale clip_lo, VarScale clip_hi, double bang, double delta,
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
        I::characterize(a, xi, p, h * i, bang, ue);
#pragma unroll
        for (int c = 0; c < NU; ++c) {
          const double est = fmin(clip_hi.s[c], fmax(clip_lo.s[c], ue[c]));
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
      const double hm = -h, ti = h * i;        // utils.py:166-175 with h < 0: stage times t, t + h/2, t + h
      I::adj_ode(a, xi, ui, p, ti, k1);
#pragma unroll
      for (int c = 0; c < NS; ++c) t[c] = a[c] + hm * k1[c] / 2;
      I::adj_ode(t, xm, um, p, ti + 0.5 * hm, k2);
#pragma unroll
      for (int c = 0; c < NS; ++c) t[c] = a[c] + hm * k2[c] / 2;
      I::adj_ode(t, xm, um, p, ti + 0.5 * hm, k3);
#pragma unroll
      for (int c = 0; c < NS; ++c) t[c] = a[c] + hm * k3[c];
      I::adj_ode(t, xj, uj, p, ti + hm, k4);
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

// ---- discrete-time variant (system.discrete: forward_backward_sweep.py:33-41, utils.py:184-188,193-197) -----------------
// The sweeps are direct recurrences with h = 1 and N = int(T) steps; u has N rows (one per step), x and adj N+1:
//   x_{i+1}   = dynamics(x_i, u_i)                              i = 0..N-1            (utils.py:186)
//   adj_{i-1} = adj_ODE(adj_i, x_i, u_{i-1})                    i = N..1              (utils.py:188: u[idx], v[idx-1])
//   u_i       = (clip(optim_characterization(adj_{i+1}, x_i)) + u_i) / 2              (shifted rows, invasive_plant.py:86-90)
// -- note the backward recurrence pairs x_i (not x_{i-1}) with u_{i-1}; that is the reference's indexing and is kept.
// The stopping rule is the one of the continuous sweep, with u summed over its N rows.
//
// INVASIVEPLANT (lenhart/invasive_plant.py:37-94): five independent foci, p = (B, k, eps), adj_T = 1.
struct DiscINVASIVEPLANT {
  static constexpr int NS = 5, NU = 5, NP = 3;
  __device__ static inline double growth(double x, const double* p) { return x + x * p[1] / (p[2] + x); }          // :70
  __device__ static inline void step(const double* x, const double* u, const double* p, double* out) {             // :68-72
#pragma unroll
    for (int c = 0; c < NS; ++c) out[c] = growth(x[c], p) * (1.0 - u[c]);
  }
  __device__ static inline void adj_prev(const double* adj, const double* x, const double* u, const double* p, double* out) {   // :77-81
#pragma unroll
    for (int c = 0; c < NS; ++c) { const double d = p[2] + x[c]; out[c] = adj[c] * (1.0 - u[c]) * (1.0 + p[2] * p[1] / (d * d)); }
  }
  __device__ static inline void characterize(const double* adj_next, const double* x, const double* p, double* u) {  // :83-90
#pragma unroll
    for (int c = 0; c < NU; ++c) u[c] = 0.5 * adj_next[c] / p[0] * growth(x[c], p);
  }
};

template <class D>
__global__ __launch_bounds__(64)
void fbsm_discrete_kernel(int B, long Bp, int N, const double* __restrict__ x0, const double* __restrict__ adjT,
                          const double* __restrict__ params, int params_stride, VarScale clip_lo, VarScale clip_hi, double delta,
                          int max_sweeps, double* xs, double* us, double* adjs, int32_t* sweeps) {
  constexpr int NS = D::NS, NU = D::NU, NP = D::NP;
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double p[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) p[i] = params[(long)b * params_stride + i];
  double* X = xs + b; double* U = us + b; double* A = adjs + b;
  auto at = [Bp](double* a, int i, int c, int nc) -> double& { return a[((long)i * nc + c) * Bp]; };
  for (int i = 0; i <= N; ++i) {
#pragma unroll
    for (int c = 0; c < NS; ++c) { at(X, i, c, NS) = (i == 0) ? x0[b * NS + c] : 0.0; at(A, i, c, NS) = (i == N && adjT) ? adjT[c] : 0.0; }
    if (i < N) {
#pragma unroll
      for (int c = 0; c < NU; ++c) at(U, i, c, NU) = 0.0;
    }
  }
  int n = 0;
  bool go = true;
  while (go && n < max_sweeps) {
    double sx[NS], dx[NS], sa[NS], da[NS], su[NU], du[NU];
#pragma unroll
    for (int c = 0; c < NS; ++c) { sx[c] = dx[c] = sa[c] = da[c] = 0.0; }
#pragma unroll
    for (int c = 0; c < NU; ++c) { su[c] = du[c] = 0.0; }
    // forward recurrence with the old controls
    double x[NS], u[NU], xn[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) { x[c] = at(X, 0, c, NS); sx[c] += fabs(x[c]); }
    for (int i = 0; i < N; ++i) {
#pragma unroll
      for (int c = 0; c < NU; ++c) u[c] = at(U, i, c, NU);
      D::step(x, u, p, xn);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        const double old = at(X, i + 1, c, NS);
        x[c] = xn[c];
        sx[c] += fabs(x[c]); dx[c] += fabs(x[c] - old);
        at(X, i + 1, c, NS) = x[c];
      }
    }
    // backward recurrence (new x, old u), then the control of the step below from the new adj_i and x_{i-1}
    double a[NS], ap[NS], xi[NS], xj[NS], ue[NU];
#pragma unroll
    for (int c = 0; c < NS; ++c) { a[c] = at(A, N, c, NS); sa[c] += fabs(a[c]); xi[c] = x[c]; }
    for (int i = N; i >= 1; --i) {
#pragma unroll
      for (int c = 0; c < NU; ++c) u[c] = at(U, i - 1, c, NU);
#pragma unroll
      for (int c = 0; c < NS; ++c) xj[c] = at(X, i - 1, c, NS);
      D::adj_prev(a, xi, u, p, ap);
      D::characterize(a, xj, p, ue);
#pragma unroll
      for (int c = 0; c < NU; ++c) {
        const double est = fmin(clip_hi.s[c], fmax(clip_lo.s[c], ue[c]));
        const double nu_ = 0.5 * (est + u[c]);
        su[c] += fabs(nu_); du[c] += fabs(nu_ - u[c]);
        at(U, i - 1, c, NU) = nu_;
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        const double old = at(A, i - 1, c, NS);
        a[c] = ap[c];
        sa[c] += fabs(a[c]); da[c] += fabs(a[c] - old);
        at(A, i - 1, c, NS) = a[c];
        xi[c] = xj[c];
      }
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
