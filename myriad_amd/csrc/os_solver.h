// os_solver.h -- "one-step" transcriptions: trapezoidal collocation and direct (multiple) shooting.
//
// Same interior-point SQP as hs_solver.h (IpLoop), with sweep cores for the two transcriptions whose stages
// couple ONE state-control point to the next:
//   TrapCore<Sys>   /root/reference/myriad/trajectory_optimizers/collocation/trapezoidal.py:16-209
//                   c_j = h/2 (f_j + f_{j+1}) - (x_{j+1} - x_j)   (:151-163, sign opposite to Hermite-Simpson),
//                   objective sum_j h/2 (g_j + g_{j+1})           (:80-128)
//   ShootCore<Sys>  /root/reference/myriad/trajectory_optimizers/shooting.py:16-278 with the Heun rule of
//                   /root/reference/myriad/utils.py:41-44,99-102: every integration step is a stage of the recursion
//                   ("lifted" Newton: identical step to the condensed problem because the intermediate states are
//                   consistent with the rollout), node states and all controls are the decision variables,
//                   c_k = Phi_k(x_k, u_k..) - x_{k+1} (:230-241), objective = integral of the cost along the same
//                   rollouts (:169-210).
// Stage variables: state s = (dx, du) [NW], input q = du_next [NU]; value function over s; pinned terminal states
// through NS extra right-hand-side columns; barrier parameter as a right-hand-side column (see hs_solver.h).
#pragma once
#include "hs_solver.h"

#ifndef MYR_TRAP_LAM_FLOOR
#define MYR_TRAP_LAM_FLOOR false   // penalty-relaxation floor at 1.1 |lambda|_inf for the trapezoidal core
#endif

namespace myriad {

// M = control rows a stage adds after its own: 1 for trapezoid / Euler / Heun / midpoint steps (du_next), 2 for an RK4
// step, whose controls are u[2i], u[2i+1], u[2i+2] (utils.py:91-96): the stage then eliminates q = (du_mid, du_next).
template <class Sys, int M = 1>
struct OsDims {
  static constexpr int NS = Sys::NS, NU = Sys::NU, NW = Sys::NW;
  static constexpr int NY = NW + M * NU;     // stage unknowns: dx, du, (du_mid,) du_next
  static constexpr int NQ = M * NU;
  static constexpr int QN = NW + (M - 1) * NU;   // column of du_next in y
  static constexpr int NC = 2 + NS;
  static constexpr int NY1 = NY + 1;
  // per-stage storage for the forward sweep: K (NQ x NW), kc (NQ x NC), Ge|ge (NS x NY1), stage gradient (NY)
  static constexpr int O_K = 0, O_KC = O_K + NQ * NW, O_GE = O_KC + NQ * NC, O_GS = O_GE + NS * NY1, STAGE = O_GS + NY;
  static constexpr int HEAD = NU * NC;
};

// One stage of the Riccati recursion shared by both cores.
//   in : P, pc   value function of the NEXT state INCLUDING that point's own terms
//        Ge      NS x NY1: dx_next = Ge [y; 1]
//        Hs, gs  stage Hessian (NY x NY) and stage gradient (NY) in y (may be null = zero)
//   out: P, pc   value function of this stage's state (own terms of this point NOT included), K, kc, Tnu updated
//        qdiag, qg1 (may be null): own bound terms of the controls that live only in this stage (the mid control of an RK4
//                   step): added to the diagonal of the q block and to the mu column of its right-hand side
template <class Sys, int M = 1>
MYR_HD inline int os_riccati_stage(double* P, double* pc, double* Tnu, const double* Ge, const double* Hs, const double* gs,
                                   double reg_floor, double* Kk, double* kc, const double* qdiag = nullptr, const double* qg1 = nullptr) {
  using D = OsDims<Sys, M>;
  constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NY = D::NY, NQ = D::NQ, NC = D::NC, NY1 = D::NY1, QN = D::QN;
  double T2[NW * NY1];
#pragma unroll
  for (int r = 0; r < NW; ++r)
#pragma unroll
    for (int c = 0; c <= NY; ++c) {
      double s = 0.0;
#pragma unroll
      for (int t = 0; t < NS; ++t) s += P[r * NW + t] * Ge[t * NY1 + c];
#pragma unroll
      for (int a = 0; a < NU; ++a) if (c == QN + a) s += P[r * NW + NS + a];
      T2[r * NY1 + c] = s;
    }
  double Q[NY * NY], qc[NY * NC];
#pragma unroll
  for (int r = 0; r < NY; ++r) {
#pragma unroll
    for (int c = 0; c < NY; ++c) {
      double s = Hs ? Hs[r * NY + c] : 0.0;
#pragma unroll
      for (int t = 0; t < NS; ++t) s += Ge[t * NY1 + r] * T2[t * NY1 + c];
#pragma unroll
      for (int a = 0; a < NU; ++a) if (r == QN + a) s += T2[(NS + a) * NY1 + c];
      if (qdiag && r == c && r >= NW && r < QN) s += qdiag[r - NW];
      Q[r * NY + c] = s;
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      double s = (cc == 0 && gs) ? gs[r] : 0.0;
#pragma unroll
      for (int t = 0; t < NS; ++t) s += Ge[t * NY1 + r] * (pc[t * NC + cc] + (cc == 0 ? T2[t * NY1 + NY] : 0.0));
#pragma unroll
      for (int a = 0; a < NU; ++a)
        if (r == QN + a) s += pc[(NS + a) * NC + cc] + (cc == 0 ? T2[(NS + a) * NY1 + NY] : 0.0);
      if (qg1 && cc == 1 && r >= NW && r < QN) s += qg1[r - NW];
      qc[r * NC + cc] = s;
    }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < NS; ++t) s += Ge[t * NY1 + NY] * pc[t * NC + 2 + i];
    Tnu[i * NC + 0] += s;
  }
  double Lq[NQ * NQ];
#pragma unroll
  for (int r = 0; r < NQ; ++r)
#pragma unroll
    for (int c = 0; c < NQ; ++c) Lq[r * NQ + c] = Q[(NW + r) * NY + NW + c];
  const int nreg = detail::chol_reg<NQ>(Lq, reg_floor);
#pragma unroll
  for (int r = 0; r < NQ; ++r) {
#pragma unroll
    for (int c = 0; c < NW; ++c) Kk[r * NW + c] = Q[(NW + r) * NY + c];
#pragma unroll
    for (int c = 0; c < NC; ++c) kc[r * NC + c] = qc[(NW + r) * NC + c];
  }
  detail::chol_solve<NQ, NW>(Lq, Kk);
  detail::chol_solve<NQ, NC>(Lq, kc);
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      double s = 0.0;
#pragma unroll
      for (int r = 0; r < NQ; ++r) s += qc[(NW + r) * NC + 2 + i] * kc[r * NC + cc];
      Tnu[i * NC + cc] -= s;
    }
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
#pragma unroll
  for (int r = 0; r < NW; ++r)
#pragma unroll
    for (int c = r + 1; c < NW; ++c) { const double v = 0.5 * (P[r * NW + c] + P[c * NW + r]); P[r * NW + c] = v; P[c * NW + r] = v; }
  return nreg;
}

// first-point elimination shared by both cores: dx_0 = 0, eliminate du_0
template <class Sys>
MYR_HD inline int os_first_point(const double* P, const double* pc, const double* Huu, const double* g0u, const double* g1u,
                                 double reg_floor, double* Tnu, double* ku_out) {
  using D = OsDims<Sys>;
  constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NC = D::NC;
  double Puu[NU * NU], ku[NU * NC], pun[NU * NS];
#pragma unroll
  for (int a = 0; a < NU; ++a) {
#pragma unroll
    for (int b = 0; b < NU; ++b) Puu[a * NU + b] = P[(NS + a) * NW + NS + b] + Huu[a * NU + b];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) ku[a * NC + cc] = pc[(NS + a) * NC + cc] + (cc == 0 ? g0u[a] : (cc == 1 ? g1u[a] : 0.0));
#pragma unroll
    for (int i = 0; i < NS; ++i) pun[a * NS + i] = ku[a * NC + 2 + i];
  }
  const int nreg = detail::chol_reg<NU>(Puu, reg_floor);
  detail::chol_solve<NU, NC>(Puu, ku);
#pragma unroll
  for (int i = 0; i < NS; ++i)
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < NU; ++a) s += pun[a * NS + i] * ku[a * NC + cc];
      Tnu[i * NC + cc] -= s;
    }
#pragma unroll
  for (int i = 0; i < NU * NC; ++i) ku_out[i] = ku[i];
  return nreg;
}

// ====================================================================================================
// Trapezoidal collocation
// ====================================================================================================
template <class Sys>
struct TrapCore {
  using H = HsSolver<Sys>;
  static constexpr bool PEN_LAM_FLOOR = MYR_TRAP_LAM_FLOOR;
  using D = OsDims<Sys>;
  static constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NY = D::NY, NQ = D::NQ, NC = D::NC, NY1 = D::NY1;
  using SweepOut = typename H::SweepOut;
  using FwdOut = typename H::FwdOut;
  using VarBlk = typename H::VarBlk;

  MYR_HD static inline int nvars(const HsSolveOpts& o) { return (o.N + 1) * NW; }
  MYR_HD static inline long stage_doubles(int N) { return (long)D::HEAD + (long)N * D::STAGE; }
  MYR_HD static inline long zi(int Kp, int j, int c) { return c < NS ? (long)j * NS + c : (long)Kp * NS + (long)j * NU + (c - NS); }
  // trapezoid weight of point j in sum_j h/2 (g_j + g_{j+1})  (trapezoidal.py:80-94)
  MYR_HD static inline double wtrap(int Kp, int j, double h) { return (j == 0 || j == Kp - 1) ? 0.5 * h : h; }

  MYR_HD static inline void load_vars(const HsWork& w, int Kp, int j, VarBlk& V) {
#pragma unroll
    for (int c = 0; c < NW; ++c) {
      const long i = zi(Kp, j, c);
      V.z[c] = w.z[i]; V.l[c] = w.lb[i]; V.u[c] = w.ub[i]; V.zl[c] = w.zL[i]; V.zu[c] = w.zU[i];
    }
  }
  MYR_HD static void init(const HsWork& w, int n) { H::init(w, n); }
  MYR_HD static void update(const HsWork& w, int n, double ap, double ad, double mu, double ksig) { H::update(w, n, ap, ad, mu, ksig); }
  MYR_HD static void solve_nu(const SweepOut& so, double mu, double* nu) { H::solve_nu(so, mu, nu); }

  MYR_HD static void backward(const HsWork& w, const HsSolveOpts& o, const double* p, const double* nuT, double delta, SweepOut& so) {
    using namespace detail;
    const int N = o.N, Kp = N + 1;
    const double h = o.h, hh = 0.5 * h;
    so.f = 0; so.c1 = 0; so.cinf = 0; so.stat = 0; so.compl_max = 0; so.compl_min = INFINITY; so.lam_inf = 0; so.sum_mult = 0; so.n_mult = 0; so.nreg = 0;
#pragma unroll
    for (int i = 0; i < NS * NC; ++i) so.Tnu[i] = 0.0;
    double P[NW * NW], pc[NW * NC];
#pragma unroll
    for (int i = 0; i < NW * NW; ++i) P[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NW * NC; ++i) pc[i] = 0.0;
    HsPoint<Sys> Pe, Ps;
    VarBlk Ve, Vs;
    load_vars(w, Kp, Kp - 1, Ve);
    set_time<Sys>(p, h * (Kp - 1));
    H::lin_point(Ve, p, Pe);
    fold_terminal<Sys>(Pe.x, Pe.u, p, wtrap(Kp, Kp - 1, h), Pe.g, Pe.gw);    // trapezoidal.py:126-127
    double pi_c[NS], ru_c[NU], mu_c[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      so.term_pinned[c] = !(Ve.l[c] < Ve.u[c]);
      pi_c[c] = so.term_pinned[c] ? nuT[c] : 0.0;
      mu_c[c] = 0.0;
      if (so.term_pinned[c]) { pc[c * NC + 2 + c] = 1.0; P[c * NW + c] = o.rho_term; }
    }
#pragma unroll
    for (int c = 0; c < NU; ++c) ru_c[c] = 0.0;

    for (int j = N - 1; j >= 0; --j) {
      load_vars(w, Kp, j, Vs);
      set_time<Sys>(p, h * j);
      H::lin_point(Vs, p, Ps);
      const double we = wtrap(Kp, j + 1, h);
      so.f += we * Pe.g;
      double cj[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        cj[c] = hh * (Ps.f[c] + Pe.f[c]) - (Pe.x[c] - Ps.x[c]);          // trapezoidal.py:161-163
        so.c1 += fabs(cj[c]);
        so.cinf = dmax(so.cinf, fabs(cj[c]));
      }
      double sig_e[NW], g1_e[NW], zlu_e[NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        typename H::BV b = H::bound_terms(Ve.z[c], Ve.l[c], Ve.u[c], Ve.zl[c], Ve.zu[c], so.compl_max, so.compl_min);
        sig_e[c] = b.sigma; g1_e[c] = b.g1; zlu_e[c] = b.zlu;
      }
      // E = I - h/2 A_e   (= -dc_j/dx_{j+1})
      double E[NS * NS];
#pragma unroll
      for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int c = 0; c < NS; ++c) E[r * NS + c] = ((r == c) ? 1.0 : 0.0) - hh * Pe.A[r * NS + c];
      lu_factor<NS>(E);
      // adjoint: E^T lam_j = own_e + pi_c
      double lam[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c)
        lam[c] = pi_c[c] + ((j == N - 1 && so.term_pinned[c]) ? 0.0 : (we * Pe.gw[c] + zlu_e[c]));
      lu_solve_t<NS>(E, lam);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        w.lam[(long)j * NS + c] = lam[c];
        so.lam_inf = dmax(so.lam_inf, fabs(lam[c]));
        so.sum_mult += fabs(lam[c]);
      }
      so.n_mult += NS;
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double r = we * Pe.gw[NS + a] + zlu_e[NS + a] + ru_c[a];
#pragma unroll
        for (int t = 0; t < NS; ++t) r += hh * Pe.B[t * NU + a] * lam[t];
        so.stat = dmax(so.stat, fabs(r));
      }
      double mue[NS], We[NW * NW];
#pragma unroll
      for (int c = 0; c < NS; ++c) mue[c] = mu_c[c] + hh * lam[c];
      Sys::hessian(Pe.x, Pe.u, p, Pe.D2, mue, we, We);
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        const bool zr = (j == N - 1) && r < NS && so.term_pinned[r];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
          const bool zc = (j == N - 1) && c < NS && so.term_pinned[c];
          if (!zr && !zc) P[r * NW + c] += We[r * NW + c] + ((r == c) ? sig_e[r] + delta : 0.0);
        }
        if (!zr) { pc[r * NC + 0] += we * Pe.gw[r]; pc[r * NC + 1] += g1_e[r]; }
      }
      // E dx_e = (h/2 A_s + I) dx_s + h/2 B_s du_s + h/2 B_e du_e + c_j
      double Ge[NS * NY1];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int c = 0; c < NS; ++c) Ge[r * NY1 + c] = ((r == c) ? 1.0 : 0.0) + hh * Ps.A[r * NS + c];
#pragma unroll
        for (int a = 0; a < NU; ++a) { Ge[r * NY1 + NS + a] = hh * Ps.B[r * NU + a]; Ge[r * NY1 + NW + a] = hh * Pe.B[r * NU + a]; }
        Ge[r * NY1 + NY] = cj[r];
      }
      lu_solve<NS, NY1>(E, Ge);
      double Kk[NQ * NW], kc[NQ * NC];
      so.nreg += os_riccati_stage<Sys>(P, pc, so.Tnu, Ge, nullptr, nullptr, o.reg_floor, Kk, kc);
      if (so.nreg > 0 && so.abort_on_reg) return;
      const long base = (long)D::HEAD + (long)j * D::STAGE;
#pragma unroll
      for (int i = 0; i < NQ * NW; ++i) w.st[base + D::O_K + i] = Kk[i];
#pragma unroll
      for (int i = 0; i < NQ * NC; ++i) w.st[base + D::O_KC + i] = kc[i];
#pragma unroll
      for (int i = 0; i < NS * NY1; ++i) w.st[base + D::O_GE + i] = Ge[i];
      // carries: stage j-1 sees this stage's start point as its end point
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        double s = lam[c];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += hh * Ps.A[t * NS + c] * lam[t];
        pi_c[c] = s;
        mu_c[c] = hh * lam[c];
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NS; ++t) s += hh * Ps.B[t * NU + a] * lam[t];
        ru_c[a] = s;
      }
      Pe = Ps; Ve = Vs;
    }
    // first point
    {
      const double w0 = wtrap(Kp, 0, h);
      so.f += w0 * Pe.g;
      double sig0[NW], g10[NW], zlu0[NW], W0[NW * NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        typename H::BV b = H::bound_terms(Ve.z[c], Ve.l[c], Ve.u[c], Ve.zl[c], Ve.zu[c], so.compl_max, so.compl_min);
        sig0[c] = b.sigma; g10[c] = b.g1; zlu0[c] = b.zlu;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) so.stat = dmax(so.stat, fabs(w0 * Pe.gw[NS + a] + zlu0[NS + a] + ru_c[a]));
      Sys::hessian(Pe.x, Pe.u, p, Pe.D2, mu_c, w0, W0);
      double Huu[NU * NU], g0u[NU], g1u[NU], ku[NU * NC];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int b = 0; b < NU; ++b) Huu[a * NU + b] = W0[(NS + a) * NW + NS + b] + ((a == b) ? sig0[NS + a] + delta : 0.0);
        g0u[a] = w0 * Pe.gw[NS + a]; g1u[a] = g10[NS + a];
      }
      so.nreg += os_first_point<Sys>(P, pc, Huu, g0u, g1u, o.reg_floor, so.Tnu, ku);
#pragma unroll
      for (int i = 0; i < NU * NC; ++i) w.st[i] = ku[i];
    }
  }

  MYR_HD static void forward(const HsWork& w, const HsSolveOpts& o, const double* p, double mu, const double* nu,
                             const bool* term_pinned, FwdOut& fo) {
    const int N = o.N, Kp = N + 1;
    const double h = o.h;
    const double tau = detail::dmax(o.tau_min, 1.0 - mu);
    fo.alpha_p = 1.0; fo.alpha_d = 1.0; fo.gphi = 0.0;
    double th[NC];
    th[0] = 1.0; th[1] = mu;
#pragma unroll
    for (int i = 0; i < NS; ++i) th[2 + i] = nu[i];
    double s[NW];
#pragma unroll
    for (int c = 0; c < NS; ++c) s[c] = 0.0;
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      double v = 0.0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v -= w.st[a * NC + cc] * th[cc];
      s[NS + a] = v;
    }
    VarBlk V;
    double gg, gw[NW];
    auto apply = [&](int j, const double* d) {
      load_vars(w, Kp, j, V);
      set_time<Sys>(p, h * j);
      Sys::cost_grad(V.z, V.z + NS, p, &gg, gw);
      const double wj = wtrap(Kp, j, h);
      if (j == Kp - 1) fold_terminal<Sys>(V.z, V.z + NS, p, wj, gg, gw);
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        w.dz[zi(Kp, j, c)] = d[c];
        H::step_limits(V.z[c], V.l[c], V.u[c], V.zl[c], V.zu[c], d[c], mu, wj * gw[c], tau, fo);
      }
    };
    apply(0, s);
    for (int j = 0; j < N; ++j) {
      const long base = (long)D::HEAD + (long)j * D::STAGE;
      double y[NY];
#pragma unroll
      for (int c = 0; c < NW; ++c) y[c] = s[c];
#pragma unroll
      for (int r = 0; r < NQ; ++r) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < NW; ++c) v -= w.st[base + D::O_K + r * NW + c] * s[c];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) v -= w.st[base + D::O_KC + r * NC + cc] * th[cc];
        y[NW + r] = v;
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double v = w.st[base + D::O_GE + r * NY1 + NY];
#pragma unroll
        for (int c = 0; c < NY; ++c) v += w.st[base + D::O_GE + r * NY1 + c] * y[c];
        s[r] = (j == N - 1 && term_pinned[r]) ? 0.0 : v;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) s[NS + a] = y[NW + a];
      apply(j + 1, s);
    }
  }

  MYR_HD static bool trial(const HsWork& w, const HsSolveOpts& o, const double* p, double alpha, double mu,
                           double& f, double& bar, double& c1) {
    const int N = o.N, Kp = N + 1;
    const double h = o.h, hh = 0.5 * h;
    f = 0; bar = 0; c1 = 0;
    int bad = 0;
    double xs[NS], us[NU], fs[NS], xe[NS], ue[NU], fe[NS];
    auto get = [&](int j, double* x, double* u, double* ff) {
      double zv[NW], dv[NW], lv[NW], uv[NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) { const long i = zi(Kp, j, c); zv[c] = w.z[i]; dv[c] = w.dz[i]; lv[c] = w.lb[i]; uv[c] = w.ub[i]; }
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
      set_time<Sys>(p, h * j);
      double gj = Sys::g(x, u, p);
      if (j == Kp - 1) fold_terminal<Sys>(x, u, p, wtrap(Kp, j, h), gj, nullptr);
      f += wtrap(Kp, j, h) * gj;
    };
    get(0, xs, us, fs);
    for (int j = 0; j < N; ++j) {
      get(j + 1, xe, ue, fe);
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        c1 += fabs(hh * (fs[c] + fe[c]) - (xe[c] - xs[c]));
        xs[c] = xe[c]; fs[c] = fe[c];
      }
    }
    bar *= mu;
    if (bad != 0) return false;
    if (!detail::finite_(f)) return false;
    if (!detail::finite_(c1)) return false;
    return detail::finite_(bar);
  }

  MYR_HD static void solve(const HsWork& w, const HsSolveOpts& o, const double* p, HsSolveResult& res) {
    IpLoop<TrapCore<Sys>>::run(w, o, p, res);
  }
};

// ====================================================================================================
// Direct (multiple) shooting, Euler / Heun steps
// ====================================================================================================
template <class Sys, int M = 1>
struct ShootCore {
  using H = HsSolver<Sys>;
  static constexpr bool PEN_LAM_FLOOR = true;
  using D = OsDims<Sys, M>;
  static constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NY = D::NY, NQ = D::NQ, NC = D::NC, NY1 = D::NY1, QN = D::QN;
  static constexpr int CM = M;                       // control rows per step: step i uses rows M i .. M i + M
  using SweepOut = typename H::SweepOut;
  using FwdOut = typename H::FwdOut;

  MYR_HD static inline int steps(const HsSolveOpts& o) { return o.N * o.cpi; }
  MYR_HD static inline int nvars(const HsSolveOpts& o) { return (o.N + 1) * NS + (M * steps(o) + 1) * NU; }
  MYR_HD static inline long stage_doubles(int I, int cpi) { return (long)D::HEAD + (long)I * cpi * D::STAGE + (long)(I * cpi + 1) * NS; }
  MYR_HD static inline long xi(int k, int c) { return (long)k * NS + c; }                                   // node state
  MYR_HD static inline long ui(const HsSolveOpts& o, int i, int a) { return (long)(o.N + 1) * NS + (long)i * NU + a; }   // control ROW i
  MYR_HD static inline double hstep(const HsSolveOpts& o) { return o.h / o.cpi; }                          // o.h = T / intervals

  MYR_HD static void init(const HsWork& w, int n) { H::init(w, n); }
  MYR_HD static void update(const HsWork& w, int n, double ap, double ad, double mu, double ksig) { H::update(w, n, ap, ad, mu, ksig); }
  MYR_HD static void solve_nu(const SweepOut& so, double mu, double* nu) { H::solve_nu(so, mu, nu); }

  // What a step linearisation asks of the system at its stage points: value + first derivatives + cost terms of stage j (`lin`), and the second
  // derivatives contracted with the stage weights (`hess`).  SysEval evaluates the system in place (closed-form systems, and the per-lane form of
  // the network); the wavefront kernel of the network system (shoot_solver_wave.h) passes an evaluator that reads what matrix-core passes over all
  // stage points of the horizon have left in records.
  struct SysEval {
    MYR_HD inline void lin(int, HsPoint<Sys>& P, const double* p) const { Sys::lin_d2(P.x, P.u, p, P.f, P.A, P.B, &P.g, P.gw, P.D2); }
    MYR_HD inline void hess(int, const HsPoint<Sys>& P, const double* p, const double* mu, double w, double* W) const { Sys::hessian(P.x, P.u, p, P.D2, mu, w, W); }
  };

  // one integration step of [x; integral of g]  (utils.py:41-44 Heun, :52-54 Euler), plain values
  // t0: time at the start of the step (cost functions with explicit time, shooting.py:196-203 integrates with the global
  // step times); last: the final step of the final interval, whose end state carries the terminal cost (shooting.py:206-208)
  MYR_HD static inline void step_val(int method, double h, const double* x, const double* u, const double* un, const double* p,
                                     double* xn, double& dc, double t0 = 0.0, bool last = false) {
    double f1[NS];
    Sys::f(x, u, p, f1);
    set_time<Sys>(p, t0);
    const double g1 = Sys::g(x, u, p);
    if (method == 0) {
#pragma unroll
      for (int c = 0; c < NS; ++c) xn[c] = x[c] + h * f1[c];
      dc = h * g1;
      if constexpr (Sys::HAS_TERMINAL) { if (last) dc += Sys::term(xn, un, p); }
      return;
    }
    double xt[NS], f2[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) xt[c] = x[c] + h * f1[c];
    if (method == 2) {      // the reference's "midpoint" rule (utils.py:47-50): full Euler predictor, averaged control, time t + h/2
      double um[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) um[a] = 0.5 * (u[a] + un[a]);
      Sys::f(xt, um, p, f2);
      set_time<Sys>(p, t0 + 0.5 * h);
      const double gm = Sys::g(xt, um, p);
#pragma unroll
      for (int c = 0; c < NS; ++c) xn[c] = x[c] + h * f2[c];
      dc = h * gm;
      if constexpr (Sys::HAS_TERMINAL) { if (last) dc += Sys::term(xn, un, p); }
      return;
    }
    Sys::f(xt, un, p, f2);
    set_time<Sys>(p, t0 + h);
    const double g2 = Sys::g(xt, un, p);
#pragma unroll
    for (int c = 0; c < NS; ++c) xn[c] = x[c] + 0.5 * h * (f1[c] + f2[c]);
    dc = 0.5 * h * (g1 + g2);
    if constexpr (Sys::HAS_TERMINAL) { if (last) dc += Sys::term(xn, un, p); }
  }

  // step linearisation: Fy (NS x NY) = d x_next / d (x, u, u_next), gy = d dc / dy, and, for costate pin of x_next,
  // Hs = d2 (dc + pin^T x_next) / dy2
  // A (linear) terminal cost on x_next is folded into the last step: its gradient joins the costate for the Hessian term
  // and is pulled back through Fy into gy.
  template <class EV = SysEval>
  MYR_HD static inline void step_lin(int method, double h, const double* x, const double* u, const double* un, const double* p,
                                     const double* pin_in, double* Fy, double* gy, double* Hs, double t0 = 0.0, bool last = false, const EV& ev = EV()) {
    double pin[NS], tg[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) tg[c] = 0.0;
    if constexpr (Sys::HAS_TERMINAL) { if (last) Sys::term_grad(x, un, p, tg); }
#pragma unroll
    for (int c = 0; c < NS; ++c) pin[c] = pin_in[c] + tg[c];
    auto fold = [&]() {
      if constexpr (Sys::HAS_TERMINAL) {
        if (last) {
#pragma unroll
          for (int c = 0; c < NY; ++c) {
            double s2 = 0.0;
#pragma unroll
            for (int t = 0; t < NS; ++t) s2 += tg[t] * Fy[t * NY + c];
            gy[c] += s2;
          }
#pragma unroll
          for (int a = 0; a < NU; ++a) gy[NW + a] += tg[NS + a];
        }
      }
    };
    HsPoint<Sys> P1;
#pragma unroll
    for (int c = 0; c < NS; ++c) P1.x[c] = x[c];
#pragma unroll
    for (int a = 0; a < NU; ++a) P1.u[a] = u[a];
    set_time<Sys>(p, t0);
    ev.lin(0, P1, p);
#pragma unroll
    for (int i = 0; i < NY * NY; ++i) Hs[i] = 0.0;
    if (method == 0) {
      double W1[NW * NW], mu1[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) mu1[c] = h * pin[c];
      ev.hess(0, P1, p, mu1, h, W1);
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int c = 0; c < NS; ++c) Fy[r * NY + c] = ((r == c) ? 1.0 : 0.0) + h * P1.A[r * NS + c];
#pragma unroll
        for (int a = 0; a < NU; ++a) { Fy[r * NY + NS + a] = h * P1.B[r * NU + a]; Fy[r * NY + NW + a] = 0.0; }
      }
#pragma unroll
      for (int c = 0; c < NW; ++c) gy[c] = h * P1.gw[c];
#pragma unroll
      for (int a = 0; a < NU; ++a) gy[NW + a] = 0.0;
#pragma unroll
      for (int r = 0; r < NW; ++r)
#pragma unroll
        for (int c = 0; c < NW; ++c) Hs[r * NY + c] = W1[r * NW + c];
      fold();
      return;
    }
    // Two-stage rules share one derivation: stage 2 is evaluated at w2 = (x + h f1, u2) and
    //   x_next = x + a1 h f1 + a2 h f2(w2),   dc = a1 h g1 + a2 h g2(w2)
    //   Heun     (utils.py:41-44): u2 = u_next,            a1 = a2 = 1/2, time t + h
    //   midpoint (utils.py:47-50): u2 = (u + u_next) / 2,  a1 = 0, a2 = 1, time t + h/2
    const bool mid = (method == 2);
    const double a1 = mid ? 0.0 : 0.5, a2 = mid ? 1.0 : 0.5, cu = mid ? 0.5 : 0.0, cun = mid ? 0.5 : 1.0;
    const double h1 = a1 * h, h2 = a2 * h;
    HsPoint<Sys> P2;
#pragma unroll
    for (int c = 0; c < NS; ++c) P2.x[c] = x[c] + h * P1.f[c];
#pragma unroll
    for (int a = 0; a < NU; ++a) P2.u[a] = cu * u[a] + cun * un[a];
    set_time<Sys>(p, t0 + (mid ? 0.5 * h : h));
    ev.lin(1, P2, p);
    // J2 = d w2 / dy (NW x NY): x~ rows [I + h A1, h B1, 0], u2 rows [0, cu I, cun I]
    double J2[NW * NY];
#pragma unroll
    for (int i = 0; i < NW * NY; ++i) J2[i] = 0.0;
#pragma unroll
    for (int r = 0; r < NS; ++r) {
#pragma unroll
      for (int c = 0; c < NS; ++c) J2[r * NY + c] = ((r == c) ? 1.0 : 0.0) + h * P1.A[r * NS + c];
#pragma unroll
      for (int a = 0; a < NU; ++a) J2[r * NY + NS + a] = h * P1.B[r * NU + a];
    }
#pragma unroll
    for (int a = 0; a < NU; ++a) { J2[(NS + a) * NY + NS + a] = cu; J2[(NS + a) * NY + NW + a] = cun; }
    // Fy = [I 0 0] + h1 [A1 B1 0] + h2 [A2 B2] J2
#pragma unroll
    for (int r = 0; r < NS; ++r)
#pragma unroll
      for (int c = 0; c < NY; ++c) {
        double s = (c < NS) ? (((r == c) ? 1.0 : 0.0) + h1 * P1.A[r * NS + c]) : ((c < NW) ? h1 * P1.B[r * NU + (c - NS)] : 0.0);
#pragma unroll
        for (int t = 0; t < NS; ++t) s += h2 * P2.A[r * NS + t] * J2[t * NY + c];
#pragma unroll
        for (int a = 0; a < NU; ++a) s += h2 * P2.B[r * NU + a] * J2[(NS + a) * NY + c];
        Fy[r * NY + c] = s;
      }
    // gy = h1 gw1 E1 + h2 gw2 J2
#pragma unroll
    for (int c = 0; c < NY; ++c) {
      double s = (c < NW) ? h1 * P1.gw[c] : 0.0;
#pragma unroll
      for (int t = 0; t < NW; ++t) s += h2 * P2.gw[t] * J2[t * NY + c];
      gy[c] = s;
    }
    // Hessians: stage 1 carries its own weight h1 (pin, g) plus what flows back through the predictor x~ = x + h f1
    double mu1[NS], mu2[NS], W1[NW * NW], W2[NW * NW];
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      double s = P2.gw[c];
#pragma unroll
      for (int t = 0; t < NS; ++t) s += P2.A[t * NS + c] * pin[t];
      mu1[c] = h1 * pin[c] + h * h2 * s;
      mu2[c] = h2 * pin[c];
    }
    ev.hess(0, P1, p, mu1, h1, W1);
    ev.hess(1, P2, p, mu2, h2, W2);
    double T[NW * NY];
#pragma unroll
    for (int r = 0; r < NW; ++r)
#pragma unroll
      for (int c = 0; c < NY; ++c) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NW; ++t) s += W2[r * NW + t] * J2[t * NY + c];
        T[r * NY + c] = s;
      }
#pragma unroll
    for (int r = 0; r < NY; ++r)
#pragma unroll
      for (int c = 0; c < NY; ++c) {
        double s = (r < NW && c < NW) ? W1[r * NW + c] : 0.0;
#pragma unroll
        for (int t = 0; t < NW; ++t) s += J2[t * NY + r] * T[t * NY + c];
        Hs[r * NY + c] = s;
      }
    fold();
  }

  // ---- RK4 step (utils.py:31-38): y = (x, u1, u2, u3); stage points W_j = (X_j, U_j) with U = (u1, u2, u2, u3), X_1 = x,
  // X_{j+1} = x + a_{j+1} h k_j, k_j = f(W_j); x_next = x + h sum b_j k_j, dc = h sum b_j g(W_j, t + a_j h) ------------------
  MYR_HD static inline void rk4_val(double h, const double* x, const double* u1, const double* u2, const double* u3, const double* p,
                                    double* xn, double& dc, double t0, bool last) {
    const double* U[4] = {u1, u2, u2, u3};
    const double aj[4] = {0.0, 0.5, 0.5, 1.0}, bj[4] = {1.0 / 6.0, 2.0 / 6.0, 2.0 / 6.0, 1.0 / 6.0};
    double k[NS], X[NS], acc[NS], g = 0.0;
#pragma unroll
    for (int c = 0; c < NS; ++c) { acc[c] = 0.0; k[c] = 0.0; }
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < NS; ++c) X[c] = x[c] + aj[j] * h * k[c];
      Sys::f(X, U[j], p, k);
      set_time<Sys>(p, t0 + aj[j] * h);
      g += bj[j] * Sys::g(X, U[j], p);
#pragma unroll
      for (int c = 0; c < NS; ++c) acc[c] += bj[j] * k[c];
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) xn[c] = x[c] + h * acc[c];
    dc = h * g;
    if constexpr (Sys::HAS_TERMINAL) { if (last) dc += Sys::term(xn, u3, p); }
  }

  // Fy (NS x NY) = d x_next / dy, gy = d dc / dy, Hs = d2 (dc + pin^T x_next) / dy2 -- exact, by forward Jacobians J_j = dW_j/dy
  // and a reverse pass for the stage weights: kappa_4 = h b_4 pin, xi_j = A_j^T kappa_j + h b_j dg/dx_j,
  // kappa_{j-1} = h b_{j-1} pin + h a_j xi_j;  Hs = sum_j J_j^T [sum_i kappa_j,i d2 f_i + h b_j d2 g](W_j) J_j
  template <class EV = SysEval>
  MYR_HD static inline void rk4_lin(double h, const double* x, const double* u1, const double* u2, const double* u3, const double* p,
                                    const double* pin_in, double* Fy, double* gy, double* Hs, double t0, bool last, const EV& ev = EV()) {
    static_assert(M == 2, "RK4 stages carry three control rows");
    const double* U[4] = {u1, u2, u2, u3};
    const int usel[4] = {0, 1, 1, 2};
    const double aj[4] = {0.0, 0.5, 0.5, 1.0}, bj[4] = {1.0 / 6.0, 2.0 / 6.0, 2.0 / 6.0, 1.0 / 6.0};
    double pin[NS], tg[NW];
#pragma unroll
    for (int c = 0; c < NW; ++c) tg[c] = 0.0;
    if constexpr (Sys::HAS_TERMINAL) { if (last) Sys::term_grad(x, u3, p, tg); }
#pragma unroll
    for (int c = 0; c < NS; ++c) pin[c] = pin_in[c] + tg[c];
    HsPoint<Sys> P[4];
    double J[4][NW * NY], dk[NS * NY], kprev[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) kprev[c] = 0.0;
#pragma unroll
    for (int i = 0; i < NS * NY; ++i) { dk[i] = 0.0; Fy[i] = 0.0; }
#pragma unroll
    for (int c = 0; c < NY; ++c) gy[c] = 0.0;
    for (int j = 0; j < 4; ++j) {
      // W_j and J_j
#pragma unroll
      for (int c = 0; c < NS; ++c) P[j].x[c] = x[c] + aj[j] * h * kprev[c];
#pragma unroll
      for (int a = 0; a < NU; ++a) P[j].u[a] = U[j][a];
#pragma unroll
      for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int c = 0; c < NY; ++c) J[j][r * NY + c] = ((r == c) ? 1.0 : 0.0) + aj[j] * h * dk[r * NY + c];
#pragma unroll
      for (int a = 0; a < NU; ++a)
#pragma unroll
        for (int c = 0; c < NY; ++c) J[j][(NS + a) * NY + c] = (c == NS + usel[j] * NU + a) ? 1.0 : 0.0;
      set_time<Sys>(p, t0 + aj[j] * h);
      ev.lin(j, P[j], p);
      // dk_j = [A_j B_j] J_j ; accumulate Fy, gy
#pragma unroll
      for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int c = 0; c < NY; ++c) {
          double v = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) v += P[j].A[r * NS + t] * J[j][t * NY + c];
#pragma unroll
          for (int a = 0; a < NU; ++a) v += P[j].B[r * NU + a] * J[j][(NS + a) * NY + c];
          dk[r * NY + c] = v;
          Fy[r * NY + c] += h * bj[j] * v;
        }
#pragma unroll
      for (int c = 0; c < NY; ++c) {
        double v = 0.0;
#pragma unroll
        for (int t = 0; t < NW; ++t) v += P[j].gw[t] * J[j][t * NY + c];
        gy[c] += h * bj[j] * v;
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) kprev[c] = P[j].f[c];
    }
#pragma unroll
    for (int r = 0; r < NS; ++r) Fy[r * NY + r] += 1.0;
    // reverse pass: stage weights and Hessian
#pragma unroll
    for (int i = 0; i < NY * NY; ++i) Hs[i] = 0.0;
    double kap[NS], W[NW * NW], T[NW * NY];
#pragma unroll
    for (int c = 0; c < NS; ++c) kap[c] = h * bj[3] * pin[c];
    for (int j = 3; j >= 0; --j) {
      ev.hess(j, P[j], p, kap, h * bj[j], W);
#pragma unroll
      for (int r = 0; r < NW; ++r)
#pragma unroll
        for (int c = 0; c < NY; ++c) {
          double v = 0.0;
#pragma unroll
          for (int t = 0; t < NW; ++t) v += W[r * NW + t] * J[j][t * NY + c];
          T[r * NY + c] = v;
        }
#pragma unroll
      for (int r = 0; r < NY; ++r)
#pragma unroll
        for (int c = 0; c < NY; ++c) {
          double v = 0.0;
#pragma unroll
          for (int t = 0; t < NW; ++t) v += J[j][t * NY + r] * T[t * NY + c];
          Hs[r * NY + c] += v;
        }
      if (j > 0) {
        double xiv[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          double v = h * bj[j] * P[j].gw[c];
#pragma unroll
          for (int t = 0; t < NS; ++t) v += P[j].A[t * NS + c] * kap[t];
          xiv[c] = v;
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) kap[c] = h * bj[j - 1] * pin[c] + h * aj[j] * xiv[c];
      }
    }
    if constexpr (Sys::HAS_TERMINAL) {
      if (last) {
#pragma unroll
        for (int c = 0; c < NY; ++c) {
          double v = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) v += tg[t] * Fy[t * NY + c];
          gy[c] += v;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) gy[QN + a] += tg[NS + a];
      }
    }
  }

  // step on the (M+1) control rows uc = [u_{Mi} | .. | u_{Mi+M}] of step i
  MYR_HD static inline void sval(int method, double h, const double* x, const double* uc, const double* p, double* xn, double& dc,
                                 double t0, bool last) {
    if constexpr (M == 2) rk4_val(h, x, uc, uc + NU, uc + 2 * NU, p, xn, dc, t0, last);
    else step_val(method, h, x, uc, uc + NU, p, xn, dc, t0, last);
  }
  template <class EV = SysEval>
  MYR_HD static inline void slin(int method, double h, const double* x, const double* uc, const double* p, const double* pin,
                                 double* Fy, double* gy, double* Hs, double t0, bool last, const EV& ev = EV()) {
    if constexpr (M == 2) rk4_lin(h, x, uc, uc + NU, uc + 2 * NU, p, pin, Fy, gy, Hs, t0, last, ev);
    else step_lin(method, h, x, uc, uc + NU, p, pin, Fy, gy, Hs, t0, last, ev);
  }

  // own (bound) terms of one decision variable
  struct Own { double sigma, g1, zlu; bool pinned; };
  MYR_HD static inline Own own_of(const HsWork& w, long i, SweepOut& so) {
    typename H::BV b = H::bound_terms(w.z[i], w.lb[i], w.ub[i], w.zL[i], w.zU[i], so.compl_max, so.compl_min);
    Own r; r.sigma = b.sigma; r.g1 = b.g1; r.zlu = b.zlu; r.pinned = b.pinned;
    return r;
  }

  MYR_HD static void backward(const HsWork& w, const HsSolveOpts& o, const double* p, const double* nuT, double delta, SweepOut& so) {
    using namespace detail;
    const int I = o.N, cpi = o.cpi, S = I * cpi, method = o.method;
    const double h = hstep(o);
    so.f = 0; so.c1 = 0; so.cinf = 0; so.stat = 0; so.compl_max = 0; so.compl_min = INFINITY; so.lam_inf = 0; so.sum_mult = 0; so.n_mult = 0; so.nreg = 0;
#pragma unroll
    for (int i = 0; i < NS * NC; ++i) so.Tnu[i] = 0.0;
    const long xs0 = (long)D::HEAD + (long)S * D::STAGE;   // rollout states x_i, i = 0..S (internal ones are not variables)
    // ---- rollout: states at every step, continuity defects, objective ----
    for (int k = 0; k < I; ++k) {
      double x[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) x[c] = w.z[xi(k, c)];
      for (int i = k * cpi; i < (k + 1) * cpi; ++i) {
        double uc[(M + 1) * NU], xn[NS], dc;
#pragma unroll
        for (int c = 0; c < NS; ++c) w.st[xs0 + (long)i * NS + c] = x[c];
#pragma unroll
        for (int a = 0; a < (M + 1) * NU; ++a) uc[a] = w.z[ui(o, M * i, a)];     // rows M i .. M i + M are contiguous
        sval(method, h, x, uc, p, xn, dc, h * i, i == S - 1);
        so.f += dc;
#pragma unroll
        for (int c = 0; c < NS; ++c) x[c] = xn[c];
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        const double ck = x[c] - w.z[xi(k + 1, c)];                        // shooting.py:239-241
        w.lam[(long)k * NS + c] = ck;                                      // parked here until the sweep overwrites it with lam_k
        so.c1 += fabs(ck);
        so.cinf = dmax(so.cinf, fabs(ck));
      }
    }
    // ---- terminal value function: own terms of u_S and of the last node state ----
    double P[NW * NW], pc[NW * NC];
#pragma unroll
    for (int i = 0; i < NW * NW; ++i) P[i] = 0.0;
#pragma unroll
    for (int i = 0; i < NW * NC; ++i) pc[i] = 0.0;
    double pi_c[NS], ru_c[NU];
#pragma unroll
    for (int c = 0; c < NS; ++c) {
      Own ow = own_of(w, xi(I, c), so);
      so.term_pinned[c] = ow.pinned;
      if (ow.pinned) { pi_c[c] = nuT[c]; pc[c * NC + 2 + c] = 1.0; P[c * NW + c] = o.rho_term; }
      else { pi_c[c] = ow.zlu; P[c * NW + c] = ow.sigma + delta; pc[c * NC + 1] = ow.g1; }
    }
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      Own ow = own_of(w, ui(o, M * S, a), so);
      ru_c[a] = ow.zlu;
      P[(NS + a) * NW + NS + a] = ow.sigma + delta; pc[(NS + a) * NC + 1] = ow.g1;
    }
    // ---- backward sweep over steps ----
    for (int i = S - 1; i >= 0; --i) {
      const int k = i / cpi;
      const bool node_next = ((i + 1) % cpi) == 0, node_here = (i % cpi) == 0;
      double pin[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) pin[c] = pi_c[c];
      double caff[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) caff[c] = 0.0;
      if (node_next) {
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          caff[c] = w.lam[(long)k * NS + c];           // c_k parked by the rollout pass
          w.lam[(long)k * NS + c] = pin[c];            // lam_k = costate of the node
          so.lam_inf = dmax(so.lam_inf, fabs(pin[c]));
          so.sum_mult += fabs(pin[c]);
        }
        so.n_mult += NS;
      }
      double x[NS], uc[(M + 1) * NU], Fy[NS * NY], gy[NY], Hs[NY * NY];
#pragma unroll
      for (int c = 0; c < NS; ++c) x[c] = w.st[xs0 + (long)i * NS + c];
#pragma unroll
      for (int a = 0; a < (M + 1) * NU; ++a) uc[a] = w.z[ui(o, M * i, a)];
      slin(method, h, x, uc, p, pin, Fy, gy, Hs, h * i, i == S - 1);
      // control-row stationarity of the step's last control (row M i + M)
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double r = ru_c[a] + gy[QN + a];
#pragma unroll
        for (int t = 0; t < NS; ++t) r += Fy[t * NY + QN + a] * pin[t];
        so.stat = dmax(so.stat, fabs(r));
      }
      // controls that live only in this stage (the mid control of an RK4 step): stationarity and own bound terms
      double qdiag[NQ], qg1[NQ];
#pragma unroll
      for (int q = 0; q < NQ; ++q) { qdiag[q] = 0.0; qg1[q] = 0.0; }
      if constexpr (M > 1) {
#pragma unroll
        for (int q = 0; q < (M - 1) * NU; ++q) {
          Own om = own_of(w, ui(o, M * i + 1, q), so);
          double r = gy[NW + q] + om.zlu;
#pragma unroll
          for (int t = 0; t < NS; ++t) r += Fy[t * NY + NW + q] * pin[t];
          so.stat = dmax(so.stat, fabs(r));
          qdiag[q] = om.sigma + delta; qg1[q] = om.g1;
        }
      }
      double Ge[NS * NY1];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int c = 0; c < NY; ++c) Ge[r * NY1 + c] = Fy[r * NY + c];
        Ge[r * NY1 + NY] = caff[r];
      }
      double Kk[NQ * NW], kc[NQ * NC];
      so.nreg += os_riccati_stage<Sys, M>(P, pc, so.Tnu, Ge, Hs, gy, o.reg_floor, Kk, kc, qdiag, qg1);
      if (so.nreg > 0 && so.abort_on_reg) return;
      const long base = (long)D::HEAD + (long)i * D::STAGE;
#pragma unroll
      for (int q = 0; q < NQ * NW; ++q) w.st[base + D::O_K + q] = Kk[q];
#pragma unroll
      for (int q = 0; q < NQ * NC; ++q) w.st[base + D::O_KC + q] = kc[q];
#pragma unroll
      for (int q = 0; q < NS * NY1; ++q) w.st[base + D::O_GE + q] = Ge[q];
#pragma unroll
      for (int q = 0; q < NY; ++q) w.st[base + D::O_GS + q] = gy[q];
      // carries + own terms of point i (added to the value function for stage i-1; point 0 is handled below)
      double npi[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        double s = gy[c];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Fy[t * NY + c] * pin[t];
        npi[c] = s;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        double s = gy[NS + a];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Fy[t * NY + NS + a] * pin[t];
        Own ow = own_of(w, ui(o, M * i, a), so);
        ru_c[a] = s + ow.zlu;
        if (i > 0) { P[(NS + a) * NW + NS + a] += ow.sigma + delta; pc[(NS + a) * NC + 1] += ow.g1; }
      }
      if (node_here && i > 0) {
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          Own ow = own_of(w, xi(k, c), so);
          npi[c] += ow.zlu;
          P[c * NW + c] += ow.sigma + delta; pc[c * NC + 1] += ow.g1;
        }
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) pi_c[c] = npi[c];
    }
    // ---- first point: x_0 pinned, eliminate du_0 ----
    {
      double Huu[NU * NU], g0u[NU], g1u[NU], ku[NU * NC];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        so.stat = dmax(so.stat, fabs(ru_c[a]));
        double cm = 0, cn = INFINITY;
        typename H::BV b = H::bound_terms(w.z[ui(o, 0, a)], w.lb[ui(o, 0, a)], w.ub[ui(o, 0, a)], w.zL[ui(o, 0, a)], w.zU[ui(o, 0, a)], cm, cn);
#pragma unroll
        for (int b2 = 0; b2 < NU; ++b2) Huu[a * NU + b2] = (a == b2) ? b.sigma + delta : 0.0;
        g0u[a] = 0.0; g1u[a] = b.g1;
      }
      so.nreg += os_first_point<Sys>(P, pc, Huu, g0u, g1u, o.reg_floor, so.Tnu, ku);
#pragma unroll
      for (int q = 0; q < NU * NC; ++q) w.st[q] = ku[q];
    }
  }

  MYR_HD static void forward(const HsWork& w, const HsSolveOpts& o, const double* p, double mu, const double* nu,
                             const bool* term_pinned, FwdOut& fo) {
    (void)p;
    const int I = o.N, cpi = o.cpi, S = I * cpi;
    const double tau = detail::dmax(o.tau_min, 1.0 - mu);
    fo.alpha_p = 1.0; fo.alpha_d = 1.0; fo.gphi = 0.0;
    double th[NC];
    th[0] = 1.0; th[1] = mu;
#pragma unroll
    for (int i = 0; i < NS; ++i) th[2 + i] = nu[i];
    double s[NW];
#pragma unroll
    for (int c = 0; c < NS; ++c) { s[c] = 0.0; w.dz[xi(0, c)] = 0.0; }
    auto setvar = [&](long i, double d) {
      w.dz[i] = d;
      H::step_limits(w.z[i], w.lb[i], w.ub[i], w.zL[i], w.zU[i], d, mu, 0.0, tau, fo);
    };
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      double v = 0.0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v -= w.st[a * NC + cc] * th[cc];
      s[NS + a] = v;
      setvar(ui(o, 0, a), v);
    }
    for (int i = 0; i < S; ++i) {
      const long base = (long)D::HEAD + (long)i * D::STAGE;
      const bool node_next = ((i + 1) % cpi) == 0;
      double y[NY];
#pragma unroll
      for (int c = 0; c < NW; ++c) y[c] = s[c];
#pragma unroll
      for (int r = 0; r < NQ; ++r) {
        double v = 0.0;
#pragma unroll
        for (int c = 0; c < NW; ++c) v -= w.st[base + D::O_K + r * NW + c] * s[c];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) v -= w.st[base + D::O_KC + r * NC + cc] * th[cc];
        y[NW + r] = v;
      }
#pragma unroll
      for (int c = 0; c < NY; ++c) fo.gphi += w.st[base + D::O_GS + c] * y[c];      // d(objective) along the lifted step
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double v = w.st[base + D::O_GE + r * NY1 + NY];
#pragma unroll
        for (int c = 0; c < NY; ++c) v += w.st[base + D::O_GE + r * NY1 + c] * y[c];
        s[r] = (i == S - 1 && term_pinned[r]) ? 0.0 : v;
      }
      if (node_next) {
        const int k1 = (i + 1) / cpi;
#pragma unroll
        for (int c = 0; c < NS; ++c) setvar(xi(k1, c), s[c]);
      }
      if constexpr (M > 1) {
#pragma unroll
        for (int q = 0; q < (M - 1) * NU; ++q) setvar(ui(o, M * i + 1, q), y[NW + q]);
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) { s[NS + a] = y[QN + a]; setvar(ui(o, M * i + M, a), y[QN + a]); }
    }
  }

  MYR_HD static bool trial(const HsWork& w, const HsSolveOpts& o, const double* p, double alpha, double mu,
                           double& f, double& bar, double& c1) {
    const int I = o.N, cpi = o.cpi, method = o.method;
    const double h = hstep(o);
    f = 0; bar = 0; c1 = 0;
    int bad = 0;
    auto val = [&](long i) -> double {
      const double v = w.z[i] + alpha * w.dz[i];
      const double l = w.lb[i], ub = w.ub[i];
      const bool fr = l < ub;
      const bool hl = fr && (l > -INFINITY), hu = fr && (ub < INFINITY);
      const double sl = hl ? v - l : 1.0, su = hu ? ub - v : 1.0;
      bad += (sl > 0.0 ? 0 : 1) + (su > 0.0 ? 0 : 1);
      bar -= log((sl > 0.0 ? sl : 1.0) * (su > 0.0 ? su : 1.0));     // one log per variable (sl + su = u - l: the pair cannot underflow)
      return v;
    };
    double uc[(M + 1) * NU];
#pragma unroll
    for (int a = 0; a < NU; ++a) uc[M * NU + a] = val(ui(o, 0, a));
    double xnode[NS];
#pragma unroll
    for (int c = 0; c < NS; ++c) xnode[c] = val(xi(0, c));
    for (int k = 0; k < I; ++k) {
      double x[NS];
#pragma unroll
      for (int c = 0; c < NS; ++c) x[c] = xnode[c];
      for (int i = k * cpi; i < (k + 1) * cpi; ++i) {
        double xn[NS], dc;
#pragma unroll
        for (int a = 0; a < NU; ++a) uc[a] = uc[M * NU + a];                       // the previous step's last control
#pragma unroll
        for (int a = 0; a < M * NU; ++a) uc[NU + a] = val(ui(o, M * i + 1, a));     // rows M i + 1 .. M i + M, each visited once
        sval(method, h, x, uc, p, xn, dc, h * i, i == I * cpi - 1);
        f += dc;
#pragma unroll
        for (int c = 0; c < NS; ++c) x[c] = xn[c];
      }
#pragma unroll
      for (int c = 0; c < NS; ++c) { xnode[c] = val(xi(k + 1, c)); c1 += fabs(x[c] - xnode[c]); }
    }
    bar *= mu;
    if (bad != 0) return false;
    if (!detail::finite_(f)) return false;
    if (!detail::finite_(c1)) return false;
    return detail::finite_(bar);
  }

  MYR_HD static void solve(const HsWork& w, const HsSolveOpts& o, const double* p, HsSolveResult& res) {
    IpLoop<ShootCore<Sys, M>>::run(w, o, p, res);
  }
};

}  // namespace myriad
