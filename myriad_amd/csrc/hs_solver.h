// hs_solver.h -- per-trajectory interior-point SQP for the Hermite-Simpson transcription (K6/K7 in DESIGN.md).
//
// This is the new `NLPSolverType.SQP` path that replaces the reference's external NLP call
// (/root/reference/myriad/nlp_solvers/__init__.py:57-58, cyipopt.minimize_ipopt) for the problem built by
// /root/reference/myriad/trajectory_optimizers/collocation/hermite_simpson.py:
//     min f(z)  s.t.  c(z) = 0,  lb <= z <= ub            (objective :243-257, constraints :325-335, bounds :55-81)
//
// Algorithm (one trajectory; on the GPU one trajectory per lane, see hs_solve_kernel in myriad_hip.hip):
//   * bounds by a primal-dual log-barrier (mu -> 0, monotone); variables with lb == ub are pinned;
//   * each iteration solves ONE equality-constrained QP -- the Newton/SQP step of the barrier problem with
//     the exact Lagrangian Hessian -- by a stage-wise Riccati recursion over the N intervals:
//       - the midpoint state is eliminated through the interpolation rows and the end state through the
//         defect rows (one ns x ns LU per interval),
//       - the remaining per-stage unknowns q = (du_mid, du_end) are eliminated by a 2nu x 2nu Cholesky,
//         convexified on the fly when not positive definite (inertia correction),
//       - pinned terminal states are handled exactly by carrying ns extra right-hand sides (their
//         multipliers nu enter linearly) and one ns x ns solve; the barrier parameter also enters the
//         right-hand side linearly, so it is chosen AFTER the backward sweep from the measured KKT error;
//   * equality multipliers are the discrete adjoint (costate) of the current iterate, obtained in the same
//     backward sweep; x-row stationarity then holds by construction and optimality is measured on the
//     control rows;
//   * globalisation: l1 merit function, backtracking Armijo line search, fraction-to-the-boundary rule.
//
// Everything here is plain scalar fp64 code shared between the device build (hipcc, one lane per trajectory,
// arrays strided by the padded batch so that lanes coalesce) and the host test twin (tests/hostsim, stride 1).
#pragma once
#include <math.h>
#include <string.h>
#include "systems_gen.h"

namespace myriad {

// variable scales of the scaled problem the solver kernels work on (by-value kernel argument; closed-form systems have
// at most 8 variables per point, the elastic twin of ROCKETLANDING 14)
struct VarScale { double s[16]; };

// MYRIAD_POISON (tests/test_gpu_poison.py): the value written over word i of what a trajectory inherits.  A constant bit pattern (a
// signalling NaN, 1e20, ...) -- or, for the pattern MYR_POISON_RANDOM, finite values of order one that differ from word to word and
// from slot to slot: "plausible leftovers", the kind a previous trajectory leaves behind (a constant can hide a read-before-write
// behind a comparison that happens to come out right).
constexpr unsigned long long MYR_POISON_RANDOM = 0x52414e444f4d5f5fULL;
MYR_HD inline double poison_value(unsigned long long pattern, unsigned long long i, unsigned long long salt) {
  if (pattern != MYR_POISON_RANDOM) { union { unsigned long long u; double d; } c; c.u = pattern; return c.d; }
  unsigned long long x = (i + 1) * 0x9E3779B97F4A7C15ULL ^ (salt + 0x632BE59BD9B4E019ULL) * 0xD1B54A32D192ED03ULL;
  x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 32;
  return ((double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 6.0;      // uniform in [-3, 3)
}

struct HsSolveOpts {
  int N;
  double h;
  int max_iter;
  double tol_feas, tol_stat, tol_compl, mu_init;
  int cpi = 1;             // controls per interval (shooting)
  int method = 1;          // integration method id (shooting): 0 Euler, 1 Heun, 2 midpoint, 3 RK4
  int restarts = 0;        // shooting wavefront kernel: further attempts of a solve that ends without a KKT point (myr_solve_opts.restarts)
  double kappa_mu = 0.2, theta_mu = 1.5, kappa_eps = 10.0;   // barrier update: mu <- max(mu_min, min(kappa_mu mu, mu^theta_mu)) when E_mu <= kappa_eps mu
  double delta_warm_min = 3e-3;   // ... only while the previous delta was at least this large (early, non-convex phase)
  int delta_warm = 0;      // 1: start the inertia correction from the previous delta / 3 instead of 0 (set per problem
                           // class by the caller: pays when a sweep costs more than a linearisation)
  int lm_abs = 1;
  double kappa_sigma = 1e10;   // bound multipliers are kept within [mu/(kappa s), kappa mu/s] after each step
  double tau_min = 0.99;   // fraction-to-the-boundary parameter: tau = max(tau_min, 1 - mu)
  int dual_follow = 0;     // 1: bound multipliers follow the primal backtracking factor (ablation knob)
  double lm_init = 3e-5;   // Levenberg-Marquardt damping seeded when the line search cuts a step to <= 1/4 (0 = off)
  int nonmono = 3;         // non-monotone Armijo memory (0 = monotone)
  int recenter = 3;        // after this many consecutive accepted steps below recenter_alpha: mu <- 10 mu (0 = off)
  double recenter_alpha = 0.2;
  double reg_floor = 1e-3; // smallest pivot accepted in the per-stage Cholesky (inertia correction threshold)
  double rho_term = 1e4;   // quadratic weight on pinned terminal states inside the QP (does not change its solution)
};

// penalty relaxation of the l1 merit function (compile-time: the by-value options struct of the kernels is left alone,
// see DESIGN.md on the compiler's sensitivity to its layout): the penalty only has to dominate the CURRENT multipliers;
// steps blocked by bounds early on can push it orders of magnitude above that, after which every full step is rejected
// for a marginal increase of the constraint violation (Maratos-type crawl: config 3's stragglers).  When it has
// exceeded PEN_RELAX_RATIO x the value the descent condition asks for during PEN_RELAX consecutive iterations it is
// reset to twice that value, at most PEN_RELAX_MAX times per solve (so the monotone argument applies from then on).
#ifndef MYR_PEN_RELAX
#define MYR_PEN_RELAX 5          // 0 = off
#endif
// warm-started inertia correction (delta_warm): the first attempt of an iteration uses delta_last / DELTA_WARM_DIV; a
// failed attempt multiplies by 8, i.e. lands at 1.33 delta_last.  Measured on the headline workload (ms per 4096
// solves / median / p99 iterations): div 3: 39.1 / 21 / 34, 4: 39.7 / 21 / 34, 5: 38.0 / 21 / 26, 6: 36.1 / 20 / 25,
// 8: 38.9 / 20 / 25, 12: 38.7 / 21 / 27 -- with 3 the retry overshoots to 2.7 delta_last and the correction ratchets up.
#ifndef MYR_DW_DIV
#define MYR_DW_DIV 6.0
#endif
constexpr double DELTA_WARM_DIV = MYR_DW_DIV;
constexpr int PEN_RELAX = MYR_PEN_RELAX, PEN_RELAX_MAX = 8;
#ifndef MYR_PEN_RELAX_LAM
#define MYR_PEN_RELAX_LAM 1.1
#endif
constexpr double PEN_RELAX_RATIO = 10.0, PEN_RELAX_LAM = MYR_PEN_RELAX_LAM;

struct HsSolveResult {
  int status, iters;
  int sweeps = 0;          // factorisation sweeps incl. inertia-correction retries
  double cost, feas, stat, compl_;
};

// strided per-trajectory view: element i of this trajectory lives at p[i*s]
struct SV {
  double* p;
  long s;
  MYR_HD inline double& operator[](long i) const { return p[i * s]; }
};

template <class Sys>
struct HsSol {
  static constexpr int NS = Sys::NS, NU = Sys::NU, NW = Sys::NW;
  static constexpr int NY = NS + 3 * NU;   // stage unknowns: dx_s, du_s, du_m, du_e
  static constexpr int NQ = 2 * NU;        // eliminated per stage: du_m, du_e
  static constexpr int NC = 2 + NS;        // right-hand-side columns: base, mu-coefficient, terminal multipliers
  // per-stage storage for the forward sweep
  static constexpr int O_K = 0, O_KC = O_K + NQ * NW, O_GE = O_KC + NQ * NC, O_gE = O_GE + NS * NY,
                       O_GM = O_gE + NS, O_gM = O_GM + NS * NY, STAGE = O_gM + NS;
  static constexpr int HEAD = NU * NC;     // first-point control gains
  MYR_HD static constexpr long stage_doubles(int N) { return (long)HEAD + (long)N * STAGE; }
};

struct HsWork {
  SV z, lb, ub, zL, zU, lam, dz, st;   // st: stage storage
};

namespace detail {

// reciprocal for the bound terms (slacks, step limits): on the device v_rcp_f64 + two Newton steps (5 instructions, <= 1 ulp)
// instead of the ~35-instruction IEEE division sequence -- a point of the wavefront solver takes about 55 of them per
// iteration; the host twin divides.
MYR_HD inline double rcp_(double x) {
#ifdef __HIP_DEVICE_COMPILE__
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
#else
  return 1.0 / x;
#endif
}
MYR_HD inline double dmax(double a, double b) { return a > b ? a : b; }
MYR_HD inline double dmin(double a, double b) { return a < b ? a : b; }
// finite <=> exponent field not all ones (bit test: immune to value-based folding under fast-math style flags)
MYR_HD inline bool finite_(double v) {
  unsigned long long b;
  memcpy(&b, &v, sizeof(b));
  return ((b >> 52) & 0x7ffULL) != 0x7ffULL;
}

// In-place LU (no pivoting; the matrices here are I + O(h)) of an n x n row-major matrix.
template <int n>
MYR_HD inline void lu_factor(double* a) {
#pragma unroll
  for (int k = 0; k < n; ++k) {
    const double inv = 1.0 / a[k * n + k];
#pragma unroll
    for (int i = k + 1; i < n; ++i) {
      const double l = a[i * n + k] * inv;
      a[i * n + k] = l;
#pragma unroll
      for (int j = k + 1; j < n; ++j) a[i * n + j] -= l * a[k * n + j];
    }
  }
}
// solve (LU) x = b for `cols` right-hand sides stored row-major b[n][cols]
template <int n, int cols>
MYR_HD inline void lu_solve(const double* a, double* b) {
#pragma unroll
  for (int c = 0; c < cols; ++c) {
#pragma unroll
    for (int i = 1; i < n; ++i) {
      double s = b[i * cols + c];
#pragma unroll
      for (int j = 0; j < i; ++j) s -= a[i * n + j] * b[j * cols + c];
      b[i * cols + c] = s;
    }
#pragma unroll
    for (int i = n - 1; i >= 0; --i) {
      double s = b[i * cols + c];
#pragma unroll
      for (int j = i + 1; j < n; ++j) s -= a[i * n + j] * b[j * cols + c];
      b[i * cols + c] = s / a[i * n + i];
    }
  }
}
// solve (LU)^T x = b for one vector
template <int n>
MYR_HD inline void lu_solve_t(const double* a, double* b) {
  // U^T y = b (forward), then L^T x = y (backward)
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double s = b[i];
#pragma unroll
    for (int j = 0; j < i; ++j) s -= a[j * n + i] * b[j];
    b[i] = s / a[i * n + i];
  }
#pragma unroll
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
#pragma unroll
    for (int j = i + 1; j < n; ++j) s -= a[j * n + i] * b[j];
    b[i] = s;
  }
}

// Cholesky a = L L^T in place (lower), convexified on the fly: a pivot below `floor_` is replaced by
// max(|pivot|, floor_) -- equivalent to adding a PSD diagonal matrix to this block (inertia correction).
template <int n>
MYR_HD inline int chol_reg(double* a, double floor_) {
  int nreg = 0;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    double d = a[j * n + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= a[j * n + k] * a[j * n + k];
    if (!(d > floor_)) { d = dmax(fabs(d), floor_); ++nreg; }
    const double l = sqrt(d);
    a[j * n + j] = l;
    const double inv = 1.0 / l;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double s = a[i * n + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= a[i * n + k] * a[j * n + k];
      a[i * n + j] = s * inv;
    }
  }
  return nreg;
}
// solve L L^T x = b, `cols` right-hand sides row-major b[n][cols]
template <int n, int cols>
MYR_HD inline void chol_solve(const double* a, double* b) {
#pragma unroll
  for (int c = 0; c < cols; ++c) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double s = b[i * cols + c];
#pragma unroll
      for (int k = 0; k < i; ++k) s -= a[i * n + k] * b[k * cols + c];
      b[i * cols + c] = s / a[i * n + i];
    }
#pragma unroll
    for (int i = n - 1; i >= 0; --i) {
      double s = b[i * cols + c];
#pragma unroll
      for (int k = i + 1; k < n; ++k) s -= a[k * n + i] * b[k * cols + c];
      b[i * cols + c] = s / a[i * n + i];
    }
  }
}

}  // namespace detail

// One collocation point's linearisation.
template <class Sys>
struct HsPoint {
  double x[Sys::NS], u[Sys::NU], f[Sys::NS], A[Sys::NS * Sys::NS], B[Sys::NS * Sys::NU], g, gw[Sys::NW], D2[Sys::NNZ2];
};

template <class Sys>
struct HsSolver {
  static constexpr bool PEN_LAM_FLOOR = false;   // penalty relaxation: floor at the multipliers (see IpLoop)
  using D = HsSol<Sys>;
  static constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NY = D::NY, NQ = D::NQ, NC = D::NC;

  // index of variable (point j, component c of w=(x,u)) in the reference's z layout
  MYR_HD static inline long zi(int K, int j, int c) { return c < NS ? (long)j * NS + c : (long)K * NS + (long)j * NU + (c - NS); }
  // Simpson weight of point j in sum_k h/6 (g_s + 4 g_m + g_e)  (hermite_simpson.py:212-214)
  MYR_HD static inline double wsimp(int K, int j, double h) {
    return (j & 1) ? 4.0 * h / 6.0 : ((j == 0 || j == K - 1) ? h / 6.0 : 2.0 * h / 6.0);
  }

  // Raw per-variable data of one collocation point (w = (x,u)): loaded UNCONDITIONALLY and up front so that the
  // 5*NW loads of a point are independent and in flight together (the kernel is latency-bound; a load hidden
  // behind a data-dependent branch costs a full memory round trip each).
  struct VarBlk { double z[NW], l[NW], u[NW], zl[NW], zu[NW]; };
  MYR_HD static inline void load_vars(const HsWork& w, int K, int j, VarBlk& V) {
#pragma unroll
    for (int c = 0; c < NW; ++c) {
      const long i = zi(K, j, c);
      V.z[c] = w.z[i]; V.l[c] = w.lb[i]; V.u[c] = w.ub[i]; V.zl[c] = w.zL[i]; V.zu[c] = w.zU[i];
    }
  }
  MYR_HD static inline void lin_point(const VarBlk& V, const double* p, HsPoint<Sys>& P) {
#pragma unroll
    for (int c = 0; c < NS; ++c) P.x[c] = V.z[c];
#pragma unroll
    for (int c = 0; c < NU; ++c) P.u[c] = V.z[NS + c];
    Sys::lin_d2(P.x, P.u, p, P.f, P.A, P.B, &P.g, P.gw, P.D2);
  }

  // bound data of one variable (branch-free): barrier Hessian sigma, (-zL + zU) for the adjoint,
  // mu-coefficient g1 = -1/(z-l) + 1/(u-z); complementarity extremes; pinned flag
  struct BV { double sigma, g1, zlu; bool pinned; };
  MYR_HD static inline BV bound_terms(double zv, double l, double u, double zl, double zu, double& compl_max, double& compl_min) {
    BV r;
    const bool fr = l < u;
    const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
    const double sl = hl ? zv - l : 1.0, su = hu ? u - zv : 1.0;
    const double zlv = hl ? zl : 0.0, zuv = hu ? zu : 0.0;
    const double il = hl ? detail::rcp_(sl) : 0.0, iu = hu ? detail::rcp_(su) : 0.0;
    r.pinned = !fr;
    r.sigma = zlv * il + zuv * iu;
    r.g1 = iu - il;
    r.zlu = zuv - zlv;
    const double cl = sl * zlv, cu = su * zuv;
    compl_max = detail::dmax(compl_max, detail::dmax(hl ? cl : compl_max, hu ? cu : compl_max));
    compl_min = detail::dmin(compl_min, detail::dmin(hl ? cl : compl_min, hu ? cu : compl_min));
    return r;
  }

  // ------------------------------------------------------------------------------------------------
  // Backward sweep at the current iterate: constraints, adjoint multipliers, KKT error, Riccati factorisation.
  // ------------------------------------------------------------------------------------------------
  struct SweepOut {
    double f, c1, cinf, stat, compl_max, compl_min, lam_inf, sum_mult;
    int n_mult;
    double Tnu[NS * NC];   // d(dual)/d(nu_i) = Tnu[i][0] + mu*Tnu[i][1] + sum_j Tnu[i][2+j] nu_j
    bool term_pinned[NS];
    int nreg;
    bool abort_on_reg;
  };

  // `delta` is added to the diagonal of every Hessian block (global inertia correction, W + delta I).
  MYR_HD static void backward(const HsWork& w, const HsSolveOpts& o, const double* p, const double* nuT, double delta, SweepOut& so) {
    using namespace detail;
    const int N = o.N, K = 2 * N + 1;
    const double h = o.h, h6 = h / 6.0, h8 = h / 8.0;
    so.f = 0; so.c1 = 0; so.cinf = 0; so.stat = 0; so.compl_max = 0; so.compl_min = INFINITY; so.lam_inf = 0; so.sum_mult = 0; so.n_mult = 0; so.nreg = 0;
#pragma unroll
    for (int i = 0; i < NS * NC; ++i) so.Tnu[i] = 0.0;

    // value function of everything after the current stage: 1/2 s'^T P s' + s'^T (pc . theta), theta=(1,mu,nu)
    double P[NW * NW], pc[NW * NC];
#pragma unroll
    for (int i = 0; i < NW * NW; ++i) P[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NW * NC; ++i) pc[i] = 0.0;

    HsPoint<Sys> Pe, Pm, Ps;
    VarBlk Ve, Vm, Vs;
    load_vars(w, K, K - 1, Ve);
    set_time<Sys>(p, 0.5 * h * (K - 1));
    lin_point(Ve, p, Pe);
    // adjoint carries from the later stage: costate on x_e rows, control-row partial residual, Hessian multiplier part
    double pi_c[NS], ru_c[NU], mu_c[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      so.term_pinned[c] = !(Ve.l[c] < Ve.u[c]);
      pi_c[c] = so.term_pinned[c] ? nuT[c] : 0.0;
      mu_c[c] = 0.0;
      if (so.term_pinned[c]) {
        pc[c * NC + 2 + c] = 1.0;            // nu_c * dx_N[c]
        P[c * NW + c] = o.rho_term;          // + rho/2 dx_N[c]^2: vanishes at the solution (dx_N = 0 there), but keeps
      }                                      //   the nu-parametrised inner problem convex (augmented-Lagrangian form)
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) ru_c[c] = 0.0;

    for (int k = N - 1; k >= 0; --k) {
      const int jm = 2 * k + 1, js = 2 * k;
      load_vars(w, K, jm, Vm);
      load_vars(w, K, js, Vs);
      set_time<Sys>(p, 0.5 * h * jm);
      lin_point(Vm, p, Pm);
      set_time<Sys>(p, 0.5 * h * js);
      lin_point(Vs, p, Ps);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MYR_LANE_ONE_SCHED_REGION)
      // The body of this loop is ONE basic block of tens of thousands of instructions once everything below is unrolled; for six states
      // and two controls (ROCKETLANDING) the compiler's pre-RA machine scheduler, given that block whole, produces a kernel in which the
      // loop-carried Pe.g reads as 0 in every iteration (the objective loses its knot terms: 1.802207 for 2.637376; round 3's "refused
      // lane instantiation").  Splitting the scheduling region here -- or anywhere in the first two thirds of the body -- gives the right
      // code, as do -O1, -fno-unroll-loops, -mllvm -enable-misched=0, -misched-fusion=false or -join-liveintervals=false, and an asm
      // marker on the value: tools/dev/repro/lane_rocket_misched/.  No instruction is added; the scheduler just sees two regions.
      __builtin_amdgcn_sched_barrier(0);
#endif
      const double we = wsimp(K, 2 * k + 2, h), wm = wsimp(K, jm, h);
      so.f += we * Pe.g + wm * Pm.g;

      // ---- constraints of interval k (hermite_simpson.py:124-128, :167-170) ----
      double dk[NS], ik[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        dk[c] = (Pe.x[c] - Ps.x[c]) - h6 * (Ps.f[c] + 4.0 * Pm.f[c] + Pe.f[c]);
        ik[c] = Pm.x[c] - 0.5 * (Ps.x[c] + Pe.x[c]) - h8 * (Ps.f[c] - Pe.f[c]);
        so.c1 += fabs(dk[c]) + fabs(ik[c]);
        so.cinf = dmax(so.cinf, dmax(fabs(dk[c]), fabs(ik[c])));
      }

      // ---- bound terms of points e and m ----
      double sig_e[NW], g1_e[NW], zlu_e[NW], sig_m[NW], g1_m[NW], zlu_m[NW];
      bool pin_e[NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        BV b = bound_terms(Ve.z[c], Ve.l[c], Ve.u[c], Ve.zl[c], Ve.zu[c], so.compl_max, so.compl_min);
        sig_e[c] = b.sigma; g1_e[c] = b.g1; zlu_e[c] = b.zlu; pin_e[c] = b.pinned;
        BV bm = bound_terms(Vm.z[c], Vm.l[c], Vm.u[c], Vm.zl[c], Vm.zu[c], so.compl_max, so.compl_min);
        sig_m[c] = bm.sigma; g1_m[c] = bm.g1; zlu_m[c] = bm.zlu;
      }

      // ---- Cm = 4h/6 A_m, Ne = I/2 - h/8 A_e, Nsm = I/2 + h/8 A_s, E = I - h/6 A_e - Cm Ne ----
      double Cm[NS * NS], Ne[NS * NS], Nsm[NS * NS], E[NS * NS];
#pragma unroll
      for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          const double id = (r == c) ? 1.0 : 0.0;
          Cm[r * NS + c] = 4.0 * h6 * Pm.A[r * NS + c];
          Ne[r * NS + c] = 0.5 * id - h8 * Pe.A[r * NS + c];
          Nsm[r * NS + c] = 0.5 * id + h8 * Ps.A[r * NS + c];
        }
#pragma unroll
      for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          double s = ((r == c) ? 1.0 : 0.0) - h6 * Pe.A[r * NS + c];
#pragma unroll
          for (int t = 0; t < NS; ++t) s -= Cm[r * NS + t] * Ne[t * NS + c];
          E[r * NS + c] = s;
        }
      lu_factor<NS>(E);

      // ---- adjoint multipliers of interval k ----
      // x_m rows:  r_m - Cm^T lam_d + lam_i = 0 ;  x_e rows:  pi_e + (I - h/6 A_e)^T lam_d - Ne^T lam_i = 0
      double rm[NS], pie[NS], lamd[NS], lami[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        rm[c] = wm * Pm.gw[c] + zlu_m[c];
        pie[c] = pi_c[c] + ((k == N - 1 && so.term_pinned[c]) ? 0.0 : (we * Pe.gw[c] + zlu_e[c]));
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        double s = pie[c];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Ne[t * NS + c] * rm[t];
        lamd[c] = -s;
      }
      lu_solve_t<NS>(E, lamd);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        double s = -rm[c];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Cm[t * NS + c] * lamd[t];
        lami[c] = s;
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        w.lam[(long)k * NS + c] = lamd[c];
        w.lam[(long)N * NS + (long)k * NS + c] = lami[c];
        so.lam_inf = dmax(so.lam_inf, dmax(fabs(lamd[c]), fabs(lami[c])));
        so.sum_mult += fabs(lamd[c]) + fabs(lami[c]);
      }
      so.n_mult += 2 * NS;

      // ---- control-row stationarity residuals of points m and e ----
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double rum = wm * Pm.gw[NS + a] + zlu_m[NS + a];
        double rue = we * Pe.gw[NS + a] + zlu_e[NS + a] + ru_c[a];
#pragma unroll
        for (int t = 0; t < NS; ++t) {
          rum -= 4.0 * h6 * Pm.B[t * NU + a] * lamd[t];
          rue += Pe.B[t * NU + a] * (-h6 * lamd[t] + h8 * lami[t]);
        }
        so.stat = dmax(so.stat, dmax(fabs(rum), fabs(rue)));
      }

      // ---- Hessians of the Lagrangian at points e and m ----
      double mue[NS], mum[NS], We[NW * NW], Wm[NW * NW];
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        mue[c] = mu_c[c] - h6 * lamd[c] + h8 * lami[c];
        mum[c] = -4.0 * h6 * lamd[c];
      }
      Sys::hessian(Pe.x, Pe.u, p, Pe.D2, mue, we, We);
      Sys::hessian(Pm.x, Pm.u, p, Pm.D2, mum, wm, Wm);
      // value function after adding point e's own terms: P' = P + H_e, pc' = pc + gbar_e
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        const bool zr = (k == N - 1) && r < NS && so.term_pinned[r];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
          const bool zc = (k == N - 1) && c < NS && so.term_pinned[c];
          if (!zr && !zc) P[r * NW + c] += We[r * NW + c] + ((r == c) ? sig_e[r] + delta : 0.0);
        }
        if (!zr) { pc[r * NC + 0] += we * Pe.gw[r]; pc[r * NC + 1] += g1_e[r]; }
        (void)pin_e;
      }
#pragma unroll
      for (int c = 0; c < NW; ++c) Wm[c * NW + c] += sig_m[c] + delta;   // H_m = W_m + Sigma_m (+ delta I)

      // ---- eliminate x_e through the defect rows: E [Ge | ge] = [R | r] ----
      double Ge[NS * (NY + 1)];   // last column = ge
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int c = 0; c < NS; ++c) {   // dx_s
          double s = ((r == c) ? 1.0 : 0.0) + h6 * Ps.A[r * NS + c];
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Cm[r * NS + t] * Nsm[t * NS + c];
          Ge[r * (NY + 1) + c] = s;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double cb_s = 0.0, cb_e = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) { cb_s += Cm[r * NS + t] * Ps.B[t * NU + a]; cb_e += Cm[r * NS + t] * Pe.B[t * NU + a]; }
          Ge[r * (NY + 1) + NS + a] = h6 * Ps.B[r * NU + a] + h8 * cb_s;            // du_s
          Ge[r * (NY + 1) + NS + NU + a] = 4.0 * h6 * Pm.B[r * NU + a];             // du_m
          Ge[r * (NY + 1) + NS + 2 * NU + a] = h6 * Pe.B[r * NU + a] - h8 * cb_e;   // du_e
        }
        double s = -dk[r];
#pragma unroll
        for (int t = 0; t < NS; ++t) s -= Cm[r * NS + t] * ik[t];
        Ge[r * (NY + 1) + NY] = s;
      }
      lu_solve<NS, NY + 1>(E, Ge);
      // dx_m = Gm y + gm :  Gm = [Nsm, h/8 B_s, 0, -h/8 B_e] + Ne Ge ; gm = -i + Ne ge
      double Gm[NS * (NY + 1)];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int c = 0; c <= NY; ++c) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Ne[r * NS + t] * Ge[t * (NY + 1) + c];
          Gm[r * (NY + 1) + c] = s;
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) Gm[r * (NY + 1) + c] += Nsm[r * NS + c];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          Gm[r * (NY + 1) + NS + a] += h8 * Ps.B[r * NU + a];
          Gm[r * (NY + 1) + NS + 2 * NU + a] -= h8 * Pe.B[r * NU + a];
        }
        Gm[r * (NY + 1) + NY] -= ik[r];
      }

      // ---- stage quadratic in y: Q = Gm^^T H_m Gm^ + Ge^^T P' Ge^ ; qc likewise (NY x NC) ----
      // augmented maps: rows 0..NS-1 = G (dx), rows NS.. = selector of du_m (for m) / du_e (for e)
      double T1[NW * (NY + 1)];   // H_m * [Gm^ | gm^]
      double T2[NW * (NY + 1)];   // P'  * [Ge^ | ge^]
#pragma unroll
      for (int r = 0; r < NW; ++r)
#pragma unroll
        for (int c = 0; c <= NY; ++c) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) { s1 += Wm[r * NW + t] * Gm[t * (NY + 1) + c]; s2 += P[r * NW + t] * Ge[t * (NY + 1) + c]; }
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            if (c == NS + NU + a) s1 += Wm[r * NW + NS + a];
            if (c == NS + 2 * NU + a) s2 += P[r * NW + NS + a];
          }
          T1[r * (NY + 1) + c] = s1; T2[r * (NY + 1) + c] = s2;
        }
      double Q[NY * NY], qc[NY * NC];
#pragma unroll
      for (int r = 0; r < NY; ++r) {
#pragma unroll
        for (int c = 0; c < NY; ++c) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Gm[t * (NY + 1) + r] * T1[t * (NY + 1) + c] + Ge[t * (NY + 1) + r] * T2[t * (NY + 1) + c];
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            if (r == NS + NU + a) s += T1[(NS + a) * (NY + 1) + c];
            if (r == NS + 2 * NU + a) s += T2[(NS + a) * (NY + 1) + c];
          }
          Q[r * NY + c] = s;
        }
        // right-hand-side columns: G^T (H g^ [col 0] + gbar cols)
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            double vm = (cc == 0) ? (T1[t * (NY + 1) + NY] + wm * Pm.gw[t]) : ((cc == 1) ? g1_m[t] : 0.0);
            double ve = pc[t * NC + cc] + ((cc == 0) ? T2[t * (NY + 1) + NY] : 0.0);
            s += Gm[t * (NY + 1) + r] * vm + Ge[t * (NY + 1) + r] * ve;
          }
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            if (r == NS + NU + a)
              s += (cc == 0) ? (T1[(NS + a) * (NY + 1) + NY] + wm * Pm.gw[NS + a]) : ((cc == 1) ? g1_m[NS + a] : 0.0);
            if (r == NS + 2 * NU + a)
              s += pc[(NS + a) * NC + cc] + ((cc == 0) ? T2[(NS + a) * (NY + 1) + NY] : 0.0);
          }
          qc[r * NC + cc] = s;
        }
      }
      // dual bookkeeping: d/dnu_i of the constant  ge^^T (pc' theta)
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Ge[t * (NY + 1) + NY] * pc[t * NC + 2 + i];
        so.Tnu[i * NC + 0] += s;
      }

      // ---- eliminate q = (du_m, du_e): Cholesky of Qqq (convexified if needed) ----
      double Lq[NQ * NQ], Kk[NQ * NW], kc[NQ * NC];
      double qscale = 0.0;
#pragma unroll
      for (int r = 0; r < NQ; ++r)
#pragma unroll
        for (int c = 0; c < NQ; ++c) { Lq[r * NQ + c] = Q[(NW + r) * NY + NW + c]; if (r == c) qscale = dmax(qscale, fabs(Lq[r * NQ + c])); }
      so.nreg += chol_reg<NQ>(Lq, o.reg_floor);
      (void)qscale;
      if (so.nreg > 0 && so.abort_on_reg) return;
#pragma unroll
      for (int r = 0; r < NQ; ++r) {
#pragma unroll
        for (int c = 0; c < NW; ++c) Kk[r * NW + c] = Q[(NW + r) * NY + c];
#pragma unroll
        for (int c = 0; c < NC; ++c) kc[r * NC + c] = qc[(NW + r) * NC + c];
      }
      chol_solve<NQ, NW>(Lq, Kk);
      chol_solve<NQ, NC>(Lq, kc);
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          double s = 0.0;
#pragma unroll
          for (int r = 0; r < NQ; ++r) s += qc[(NW + r) * NC + 2 + i] * kc[r * NC + cc];
          so.Tnu[i * NC + cc] -= s;
        }
      // new value function (of s = (dx_s, du_s)), point s's own terms are added by the next stage
#pragma unroll
      for (int r = 0; r < NW; ++r) {
#pragma unroll
        for (int c = 0; c < NW; ++c) {
          double s = Q[r * NY + c];
#pragma unroll
          for (int t = 0; t < NQ; ++t) s -= Q[r * NY + NW + t] * Kk[t * NW + c];
          P[r * NW + c] = s;
        }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          double s = qc[r * NC + cc];
#pragma unroll
          for (int t = 0; t < NQ; ++t) s -= Q[r * NY + NW + t] * kc[t * NC + cc];
          pc[r * NC + cc] = s;
        }
      }
      // symmetrise P against round-off drift
#pragma unroll
      for (int r = 0; r < NW; ++r)
#pragma unroll
        for (int c = r + 1; c < NW; ++c) { const double v = 0.5 * (P[r * NW + c] + P[c * NW + r]); P[r * NW + c] = v; P[c * NW + r] = v; }

      // ---- store what the forward sweep needs ----
      const long base = (long)D::HEAD + (long)k * D::STAGE;
#pragma unroll
      for (int i = 0; i < NQ * NW; ++i) w.st[base + D::O_K + i] = Kk[i];
#pragma unroll
      for (int i = 0; i < NQ * NC; ++i) w.st[base + D::O_KC + i] = kc[i];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int c = 0; c < NY; ++c) { w.st[base + D::O_GE + r * NY + c] = Ge[r * (NY + 1) + c]; w.st[base + D::O_GM + r * NY + c] = Gm[r * (NY + 1) + c]; }
        w.st[base + D::O_gE + r] = Ge[r * (NY + 1) + NY];
        w.st[base + D::O_gM + r] = Gm[r * (NY + 1) + NY];
      }

      // ---- carries for stage k-1 (whose end point is this stage's start point) ----
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        double s = -lamd[c] - 0.5 * lami[c];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Ps.A[t * NS + c] * (-h6 * lamd[t] - h8 * lami[t]);
        pi_c[c] = s;
        mu_c[c] = -h6 * lamd[c] - h8 * lami[c];
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Ps.B[t * NU + a] * (-h6 * lamd[t] - h8 * lami[t]);
        ru_c[a] = s;
      }
      Pe = Ps;
      Ve = Vs;
    }

    // ---- first point (j = 0): x_0 is pinned (dx_0 = 0); add its control terms and eliminate du_0 ----
    {
      const double w0 = wsimp(K, 0, h);
      so.f += w0 * Pe.g;
      double sig0[NW], g10[NW], zlu0[NW], W0[NW * NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        BV b = bound_terms(Ve.z[c], Ve.l[c], Ve.u[c], Ve.zl[c], Ve.zu[c], so.compl_max, so.compl_min);
        sig0[c] = b.sigma; g10[c] = b.g1; zlu0[c] = b.zlu;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) so.stat = dmax(so.stat, fabs(w0 * Pe.gw[NS + a] + zlu0[NS + a] + ru_c[a]));
      Sys::hessian(Pe.x, Pe.u, p, Pe.D2, mu_c, w0, W0);
      double Puu[NU * NU], ku[NU * NC];
      double uscale = 0.0;
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int b2 = 0; b2 < NU; ++b2) {
          Puu[a * NU + b2] = P[(NS + a) * NW + NS + b2] + W0[(NS + a) * NW + NS + b2] + ((a == b2) ? sig0[NS + a] + delta : 0.0);
          if (a == b2) uscale = dmax(uscale, fabs(Puu[a * NU + b2]));
        }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)
          ku[a * NC + cc] = pc[(NS + a) * NC + cc] + ((cc == 0) ? w0 * Pe.gw[NS + a] : ((cc == 1) ? g10[NS + a] : 0.0));
      }
      double pu_nu[NU * NS];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int i = 0; i < NS; ++i) pu_nu[a * NS + i] = ku[a * NC + 2 + i];
      so.nreg += chol_reg<NU>(Puu, o.reg_floor);
      (void)uscale;
      chol_solve<NU, NC>(Puu, ku);
#pragma unroll
      for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
          double s = 0.0;
#pragma unroll
          for (int a = 0; a < NU; ++a) s += pu_nu[a * NS + i] * ku[a * NC + cc];
          so.Tnu[i * NC + cc] -= s;
        }
#pragma unroll
      for (int i = 0; i < NU * NC; ++i) w.st[i] = ku[i];
    }
  }

  // Solve for the terminal multipliers: (-Tnu[:,2:]) nu = Tnu[:,0] + mu Tnu[:,1]  (pinned components only).
  MYR_HD static inline void solve_nu(const SweepOut& so, double mu, double* nu) {
    using namespace detail;
    double M[NS * NS], rhs[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      rhs[i] = so.Tnu[i * NC + 0] + mu * so.Tnu[i * NC + 1];
#pragma unroll
      for (int j = 0; j < NS; ++j) M[i * NS + j] = -0.5 * (so.Tnu[i * NC + 2 + j] + so.Tnu[j * NC + 2 + i]);
    }
    double sc = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (!so.term_pinned[i]) {
#pragma unroll
        for (int j = 0; j < NS; ++j) { M[i * NS + j] = 0.0; M[j * NS + i] = 0.0; }
        M[i * NS + i] = 1.0; rhs[i] = 0.0;
      }
      sc = dmax(sc, fabs(M[i * NS + i]));
    }
    chol_reg<NS>(M, 1e-14 * dmax(sc, 1e-300));
    chol_solve<NS, 1>(M, rhs);
#pragma unroll
    for (int i = 0; i < NS; ++i) nu[i] = rhs[i];
  }

  // ------------------------------------------------------------------------------------------------
  // Forward sweep: step dz for theta = (1, mu, nu); also step-length limits and merit slope.
  // ------------------------------------------------------------------------------------------------
  struct FwdOut { double alpha_p, alpha_d, gphi; };

  // accumulates gphi = grad(phi_mu)^T dz and the fraction-to-the-boundary limits for z (primal) and zL, zU (dual);
  // branch-free, operands already in registers
  MYR_HD static inline void step_limits(double zv, double l, double u, double zl, double zu, double d, double mu,
                                        double wg_grad, double tau, FwdOut& fo) {
    const bool fr = l < u;
    const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
    const double sl = hl ? zv - l : 1.0, su = hu ? u - zv : 1.0;
    const double zlv = hl ? zl : 1.0, zuv = hu ? zu : 1.0;
    // five fp64 divisions instead of eight (each is a ~35-instruction sequence): one reciprocal per slack, one for the
    // step component (only the bound the step moves towards can limit it)
    const double rsl = detail::rcp_(sl), rsu = detail::rcp_(su);
    double gb = wg_grad;
    gb -= hl ? mu * rsl : 0.0;
    gb += hu ? mu * rsu : 0.0;
    const double dzl = -zlv + (mu - zlv * d) * rsl;
    const double dzu = -zuv + (mu + zuv * d) * rsu;
    const bool tol_ = hl && d < 0.0, tou_ = hu && d > 0.0;
    const double ap_ = (tol_ || tou_) ? tau * (tol_ ? sl : su) * detail::rcp_(fabs(d)) : 1.0;
    const double ad_l = (hl && dzl < 0.0) ? -tau * zlv * detail::rcp_(dzl) : 1.0;
    const double ad_u = (hu && dzu < 0.0) ? -tau * zuv * detail::rcp_(dzu) : 1.0;
    fo.alpha_p = detail::dmin(fo.alpha_p, ap_);
    fo.alpha_d = detail::dmin(fo.alpha_d, detail::dmin(ad_l, ad_u));
    fo.gphi += fr ? gb * d : 0.0;
  }

  MYR_HD static void forward(const HsWork& w, const HsSolveOpts& o, const double* p, double mu, const double* nu,
                             const bool* term_pinned, FwdOut& fo) {
    const int N = o.N, K = 2 * N + 1;
    const double h = o.h;
    const double tau = detail::dmax(o.tau_min, 1.0 - mu);
    fo.alpha_p = 1.0; fo.alpha_d = 1.0; fo.gphi = 0.0;
    double th[NC];
    th[0] = 1.0; th[1] = mu;
#pragma unroll
    for (int i = 0; i < NS; ++i) th[2 + i] = nu[i];
    double s[NW];
#pragma unroll
    for (int c = 0; c < NS; ++c) { s[c] = 0.0; w.dz[zi(K, 0, c)] = 0.0; }
    // first-point controls
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      double v = 0.0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v -= w.st[a * NC + cc] * th[cc];
      s[NS + a] = v;
    }
    VarBlk V;
    double gg, gw[NW];
    auto apply = [&](int j, const double* dx, const double* du) {
      // objective gradient of point j (cheap closed form), then step limits / merit slope for its NW variables
      load_vars(w, K, j, V);
      set_time<Sys>(p, 0.5 * h * j);
      Sys::cost_grad(V.z, V.z + NS, p, &gg, gw);
      const double wj = wsimp(K, j, h);
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        const double d = c < NS ? dx[c] : du[c - NS];
        w.dz[zi(K, j, c)] = d;
        step_limits(V.z[c], V.l[c], V.u[c], V.zl[c], V.zu[c], d, mu, wj * gw[c], tau, fo);
      }
    };
    apply(0, s, s + NS);
    for (int k = 0; k < N; ++k) {
      const long base = (long)D::HEAD + (long)k * D::STAGE;
      // stage data: independent loads first
      double Kk[NQ * NW], kc[NQ * NC], Ge[NS * NY], ge[NS], Gm[NS * NY], gm[NS];
#pragma unroll
      for (int i = 0; i < NQ * NW; ++i) Kk[i] = w.st[base + D::O_K + i];
#pragma unroll
      for (int i = 0; i < NQ * NC; ++i) kc[i] = w.st[base + D::O_KC + i];
#pragma unroll
      for (int i = 0; i < NS * NY; ++i) { Ge[i] = w.st[base + D::O_GE + i]; Gm[i] = w.st[base + D::O_GM + i]; }
#pragma unroll
      for (int i = 0; i < NS; ++i) { ge[i] = w.st[base + D::O_gE + i]; gm[i] = w.st[base + D::O_gM + i]; }
      double y[NY];
#pragma unroll
      for (int c = 0; c < NW; ++c) y[c] = s[c];
#pragma unroll
      for (int r = 0; r < NQ; ++r) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < NW; ++c) v -= Kk[r * NW + c] * s[c];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) v -= kc[r * NC + cc] * th[cc];
        y[NW + r] = v;
      }
      double dxe[NS], dxm[NS];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double ve = ge[r], vm = gm[r];
#pragma unroll
        for (int c = 0; c < NY; ++c) { ve += Ge[r * NY + c] * y[c]; vm += Gm[r * NY + c] * y[c]; }
        dxe[r] = ve; dxm[r] = vm;
      }
      if (k == N - 1) {
#pragma unroll
        for (int r = 0; r < NS; ++r) dxe[r] = term_pinned[r] ? 0.0 : dxe[r];
      }
      apply(2 * k + 1, dxm, y + NW);
      apply(2 * k + 2, dxe, y + NW + NU);
#pragma unroll
      for (int c = 0; c < NS; ++c) s[c] = dxe[c];
#pragma unroll
      for (int a = 0; a < NU; ++a) s[NS + a] = y[NW + NU + a];
    }
  }

  // merit pieces at z + alpha dz: objective, barrier, ||c||_1.  Returns false on a non-finite value.
  MYR_HD static bool trial(const HsWork& w, const HsSolveOpts& o, const double* p, double alpha, double mu,
                           double& f, double& bar, double& c1) {
    const int N = o.N, K = 2 * N + 1;
    const double h = o.h, h6 = h / 6.0, h8 = h / 8.0;
    f = 0; bar = 0; c1 = 0;
    double xs[NS], us[NU], fs[NS], xm[NS], um[NU], fm[NS], xe[NS], ue[NU], fe[NS];
    int bad = 0;   // number of bound violations (kept as an integer count, not a bool chain)
    auto get = [&](int j, double* x, double* u, double* ff) {
      double zv[NW], dv[NW], lv[NW], uv[NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        const long i = zi(K, j, c);
        zv[c] = w.z[i]; dv[c] = w.dz[i]; lv[c] = w.lb[i]; uv[c] = w.ub[i];
      }
      double slk = 1.0; int sexp = 0;
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        const double v = zv[c] + alpha * dv[c];
        const bool fr = lv[c] < uv[c];
        const bool hl = fr && (lv[c] > -INFINITY), hu = fr && (uv[c] < INFINITY);
        const double sl = hl ? v - lv[c] : 1.0, su = hu ? uv[c] - v : 1.0;
        bad += (sl > 0.0 ? 0 : 1) + (su > 0.0 ? 0 : 1);
        { int e_; slk *= frexp((sl > 0.0 ? sl : 1.0) * (su > 0.0 ? su : 1.0), &e_); sexp += e_; }
        if (c < NS) x[c] = v; else u[c - NS] = v;
      }
      // one log per point instead of 2 NW: slack pairs multiplied as mantissas, binary exponents summed (no under/overflow)
      bar -= log(slk) + sexp * 0.6931471805599453;
      Sys::f(x, u, p, ff);
      set_time<Sys>(p, 0.5 * h * j);
      f += wsimp(K, j, h) * Sys::g(x, u, p);
    };
    get(0, xs, us, fs);
    for (int k = 0; k < N; ++k) {
      get(2 * k + 1, xm, um, fm);
      get(2 * k + 2, xe, ue, fe);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        c1 += fabs((xe[c] - xs[c]) - h6 * (fs[c] + 4.0 * fm[c] + fe[c]));
        c1 += fabs(xm[c] - 0.5 * (xs[c] + xe[c]) - h8 * (fs[c] - fe[c]));
        xs[c] = xe[c]; fs[c] = fe[c];
      }
    }
    bar *= mu;
    if (bad != 0) return false;
    if (!detail::finite_(f)) return false;
    if (!detail::finite_(c1)) return false;
    return detail::finite_(bar);
  }

  // accept the step: z += a_p dz, zL += a_d dzL, zU += a_d dzU (with the usual safeguard on the bound multipliers)
  MYR_HD static void update(const HsWork& w, int n, double ap, double ad, double mu, double ksig = 1e10) {
    const double iks = 1.0 / ksig;
    for (int i = 0; i < n; ++i) {
      const double l = w.lb[i], u = w.ub[i], zv = w.z[i], d = w.dz[i], zl = w.zL[i], zu = w.zU[i];
      const bool fr = l < u;
      const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
      const double zn = fr ? zv + ap * d : zv;
      const double sl = hl ? zv - l : 1.0, su = hu ? u - zv : 1.0;
      const double snl = hl ? zn - l : 1.0, snu = hu ? u - zn : 1.0;
      double vl = zl + ad * (-zl + (mu - zl * d) / sl);
      double vu = zu + ad * (-zu + (mu + zu * d) / su);
      const double ml = mu / snl, mu_ = mu / snu;        // one division per new slack; the safeguard band is [m / ksig, m ksig]
      vl = detail::dmax(detail::dmin(vl, ksig * ml), ml * iks);
      vu = detail::dmax(detail::dmin(vu, ksig * mu_), mu_ * iks);
      w.z[i] = zn;
      w.zL[i] = hl ? vl : 0.0;
      w.zU[i] = hu ? vu : 0.0;
    }
  }

  // starting point: pinned variables on their value, the others pushed strictly inside their bounds
  MYR_HD static void init(const HsWork& w, int n) {
    const double k1 = 1e-2, k2 = 1e-2;
    for (int i = 0; i < n; ++i) {
      const double l = w.lb[i], u = w.ub[i], v0 = w.z[i];
      const bool fr = l < u;
      const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
      const double width = (hl && hu) ? (u - l) : INFINITY;
      const double pl = detail::dmin(k1 * detail::dmax(1.0, fabs(l)), k2 * width);
      const double pu = detail::dmin(k1 * detail::dmax(1.0, fabs(u)), k2 * width);
      double v = v0;
      v = hl ? detail::dmax(v, l + pl) : v;
      v = hu ? detail::dmin(v, u - pu) : v;
      v = fr ? v : l;
      w.z[i] = v;
      w.zL[i] = hl ? 1.0 : 0.0;
      w.zU[i] = hu ? 1.0 : 0.0;
    }
  }

  MYR_HD static inline int nvars(const HsSolveOpts& o) { return (2 * o.N + 1) * NW; }
  MYR_HD static void solve(const HsWork& w, const HsSolveOpts& o, const double* p, HsSolveResult& res);
};

// ----------------------------------------------------------------------------------------------------
// The interior-point outer loop, shared by every transcription core (Hermite-Simpson here, trapezoidal and
// shooting in os_solver.h).  `Core` supplies: NS, nvars(o), init, SweepOut, backward, solve_nu, FwdOut, forward,
// trial, update.
// ----------------------------------------------------------------------------------------------------
template <class Core>
struct IpLoop {
  MYR_HD static void run(const HsWork& w, const HsSolveOpts& o, const double* p, HsSolveResult& res) {
    using namespace detail;
    constexpr int NS = Core::NS;
    using SweepOut = typename Core::SweepOut;
    using FwdOut = typename Core::FwdOut;
    const int n = Core::nvars(o);
    Core::init(w, n);
    double mu = o.mu_init, pen = 1.0;
    int pen_over = 0, pen_cuts = 0;
    double nuT[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) nuT[i] = 0.0;
    const double mu_min = dmin(o.tol_compl, o.tol_stat) * 0.1;
    res.status = 1; res.iters = o.max_iter;
    int stall = 0, small_steps = 0;
    double delta_last = 0.0, lm = 0.0;
    constexpr int NMMAX = 8;
    double hist[NMMAX]; int nhist = 0, hpos = 0; double hist_mu = -1.0, hist_pen = -1.0;
    SweepOut so;
    for (int it = 0; it <= o.max_iter; ++it) {
      // inertia correction (global, as in interior-point NLP codes): retry the factorisation with W + delta I
      // until every stage pivot is positive; the last resort keeps the stage-local convexification.
      double delta = lm;     // Levenberg-Marquardt floor adapted from the line-search history (see below)
      if (o.delta_warm && delta_last > o.delta_warm_min) delta = dmax(delta, delta_last / DELTA_WARM_DIV);   // skip the doomed delta = 0 attempt
      for (int tr_ = 0; tr_ < 12; ++tr_) {
        so.abort_on_reg = (tr_ < 11);
        Core::backward(w, o, p, nuT, delta, so);
        ++res.sweeps;
        if (so.nreg == 0) break;
        if (delta == 0.0) delta = (delta_last > 0.0) ? dmax(1e-8, delta_last / 3.0) : 1e-4;
        else delta *= (delta_last > 0.0) ? 8.0 : 100.0;
        if (delta > 1e8) { so.abort_on_reg = false; }
      }
      delta_last = (delta > lm) ? delta : 0.0;
      // KKT error with the usual multiplier scaling
      double sd = 1.0;
      {
        double sm = so.sum_mult; int nm = so.n_mult;
        for (int i = 0; i < n; ++i) {
          const double l = w.lb[i], u = w.ub[i], zl = w.zL[i], zu = w.zU[i];
          const bool fr = l < u;
          const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
          sm += (hl ? zl : 0.0) + (hu ? zu : 0.0);
          nm += (hl ? 1 : 0) + (hu ? 1 : 0);
        }
        if (nm > 0) sd = dmax(1.0, sm / nm / 100.0);
      }
      const double stat = so.stat / sd, comp = so.compl_max / sd;
      res.cost = so.f; res.feas = so.cinf; res.stat = stat; res.compl_ = comp;
      if (!(finite_(so.f) && finite_(so.cinf) && finite_(so.stat))) { res.status = 2; res.iters = it; return; }
      if (so.cinf <= o.tol_feas && stat <= o.tol_stat && comp <= o.tol_compl) { res.status = 0; res.iters = it; return; }
      if (it == o.max_iter) break;
      // barrier update (monotone, superlinear): error of the barrier problem vs kappa_eps * mu
      for (int guard = 0; guard < 8; ++guard) {
        // error of the barrier problem: complementarity |s*z - mu| from the extreme products
        const double cerr = (so.compl_min <= so.compl_max) ? dmax(fabs(so.compl_max - mu), fabs(so.compl_min - mu)) : 0.0;
        const double emu = dmax(dmax(stat, so.cinf), cerr / sd);
        if (emu <= o.kappa_eps * mu && mu > mu_min) {
          const double nm = dmax(mu_min, dmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
          mu = nm;
        } else break;
      }
      double nu[NS];
      Core::solve_nu(so, mu, nu);
      FwdOut fo;
      Core::forward(w, o, p, mu, nu, so.term_pinned, fo);
      if (!(finite_(fo.gphi) && finite_(fo.alpha_p))) { res.status = 2; res.iters = it; return; }
      // l1 merit: penalty large enough to make dz a descent direction
      if (so.c1 > 0.0) {
        const double need = fo.gphi / (0.9 * so.c1);
        if (pen < need) pen = need + 1.0;
        if (PEN_RELAX > 0) {
          // never below the multipliers where the core asks for it (exactness of the l1 penalty): single shooting, whose
          // stragglers need it (config 3: slowest solve 257 -> 130 iterations).  The collocation cores keep the plain
          // rule: with the floor the trapezoidal lane kernel lost 9 % of a CARTPOLE batch on the GPU -- in fixed lane
          // positions, and not on the host build of the same code (compiler fragility).
          const double floor_ = Core::PEN_LAM_FLOOR ? PEN_RELAX_LAM * so.lam_inf : 0.0;
          const double want = dmax(2.0 * dmax(need, 0.0) + 1.0, floor_);
          pen_over = (pen > PEN_RELAX_RATIO * want) ? pen_over + 1 : 0;
          if (pen_over >= PEN_RELAX && pen_cuts < PEN_RELAX_MAX) { pen = want; pen_over = 0; ++pen_cuts; }
        }
      }
      const double Dphi = fo.gphi - pen * so.c1;
      double f0, bar0, c10;
      Core::trial(w, o, p, 0.0, mu, f0, bar0, c10);
      const double phi0 = f0 + bar0 + pen * c10;
      // non-monotone Armijo reference (Grippo-Lampariello-Lucidi): the largest of the last NM merit values of the
      // SAME merit function (history is dropped whenever mu or the penalty changes); cures Maratos-type stalls
      if (mu != hist_mu || pen != hist_pen) { nhist = 0; hpos = 0; hist_mu = mu; hist_pen = pen; }
      double phiref = phi0;
      for (int j = 0; j < nhist; ++j) phiref = dmax(phiref, hist[j]);
      if (o.nonmono > 0) { hist[hpos % o.nonmono] = phi0; ++hpos; if (nhist < o.nonmono) ++nhist; }
      double a = fo.alpha_p;
      bool ok = false;
      for (int ls = 0; ls < 40; ++ls) {
        double ft, bt, ct;
        if (Core::trial(w, o, p, a, mu, ft, bt, ct)) {
          const double phit = ft + bt + pen * ct;
          if (phit <= phiref + 1e-8 * a * Dphi + 1e-13 * fabs(phi0)) { ok = true; break; }
        }
        a *= 0.5;
      }
      if (!ok) {
        // no acceptable step along dz: take the tiny step anyway a few times (helps past round-off), then give up
        if (++stall > 5) { res.status = 3; res.iters = it; return; }
      } else stall = 0;
      // bound multipliers follow the primal backtracking factor (keeps s*z near mu when the line search cuts the step)
      const double ad = o.dual_follow ? fo.alpha_d * (a / fo.alpha_p) : fo.alpha_d;
#ifdef MYR_TRACE
      printf("it %3d f=%.8f cinf=%.2e stat=%.2e comp=%.2e mu=%.1e a=%.3g amax=%.3g ad=%.3g nreg=%d pen=%.3g gphi=%.3g ok=%d\n", it, so.f, so.cinf, stat, comp, mu, a, fo.alpha_p, ad, so.nreg, pen, fo.gphi, (int)ok);
#endif
      Core::update(w, n, a, ad, mu, o.kappa_sigma);
#pragma unroll
      for (int i = 0; i < NS; ++i) nuT[i] += a * (nu[i] - nuT[i]);
      // re-centering: a run of tiny accepted steps means the iterate left the neighbourhood of the central path for
      // this mu (barrier parameter reduced too early); go back up one decade instead of crawling
      // step-quality feedback: a step cut hard by the line search means the quadratic model over-reaches ->
      // damp the next Newton system (W + lm I); full steps relax the damping again
      if (o.lm_init > 0.0) {
        const double ratio = o.lm_abs ? a : a / fo.alpha_p;   // step actually taken, relative to the full Newton step
        if (ratio <= 0.25) lm = dmin(1e2, dmax(o.lm_init, 4.0 * lm));
        else if (ratio >= 0.99) { lm *= 0.25; if (lm < 0.1 * o.lm_init) lm = 0.0; }
      }
      if (o.recenter > 0) {
        small_steps = (a < o.recenter_alpha) ? small_steps + 1 : 0;
        if (small_steps >= o.recenter && mu < o.mu_init) { mu = dmin(o.mu_init, 10.0 * mu); small_steps = 0; }
      }
    }
    res.status = 1; res.iters = o.max_iter;
  }
};

template <class Sys>
MYR_HD void HsSolver<Sys>::solve(const HsWork& w, const HsSolveOpts& o, const double* p, HsSolveResult& res) {
  IpLoop<HsSolver<Sys>>::run(w, o, p, res);
}

}  // namespace myriad
