// hostsim.cpp -- TEST-ONLY host build of the solver core (myriad_amd/csrc/hs_solver.h) so the per-trajectory
// SQP algebra can be exercised by the CPU test-suite (`-m "not gpu"`) in a container without a GPU.
// It is never loaded by the myriad_amd package and is not a fallback: the product path is the HIP kernel
// that calls the very same HsSolver<Sys>::solve, one trajectory per lane.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../myriad_amd/csrc/hs_solver.h"
#include "../../myriad_amd/csrc/rollout.h"
#include "../../myriad_amd/csrc/os_solver.h"

using namespace myriad;

// test/dev knobs: every solver option can be overridden from the environment
static void apply_env(HsSolveOpts& o) {
  if (getenv("RHO")) o.rho_term = atof(getenv("RHO"));
  if (getenv("REGF")) o.reg_floor = atof(getenv("REGF"));
  if (getenv("NONM")) o.nonmono = atoi(getenv("NONM"));
  if (getenv("DUALF")) o.dual_follow = atoi(getenv("DUALF"));
  if (getenv("LM")) o.lm_init = atof(getenv("LM"));
  if (getenv("TAU")) o.tau_min = atof(getenv("TAU"));
  if (getenv("LMABS")) o.lm_abs = atoi(getenv("LMABS"));
  if (getenv("DWARM")) o.delta_warm = atoi(getenv("DWARM"));
  if (getenv("DWMIN")) o.delta_warm_min = atof(getenv("DWMIN"));
  if (getenv("KSIG")) o.kappa_sigma = atof(getenv("KSIG"));
  if (getenv("KMU")) o.kappa_mu = atof(getenv("KMU"));
  if (getenv("TMU")) o.theta_mu = atof(getenv("TMU"));
  if (getenv("KEPS")) o.kappa_eps = atof(getenv("KEPS"));
  if (getenv("RECN")) o.recenter = atoi(getenv("RECN"));
  if (getenv("RECA")) o.recenter_alpha = atof(getenv("RECA"));
}

template <class Sys>
static void solve_batch(int N, double T, int B, double* z, const double* lb, const double* ub, const double* params,
                        int pstride, int max_iter, double tol_feas, double tol_stat, double tol_compl, double mu_init,
                        double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt) {
  using S = HsSolver<Sys>;
  const int K = 2 * N + 1, n = K * Sys::NW, m = 2 * N * Sys::NS;
  HsSolveOpts o{N, T / N, max_iter, tol_feas, tol_stat, tol_compl, mu_init};
  apply_env(o);

#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    std::vector<double> zL(n), zU(n), dz(n), lbv(lb + (size_t)b * n, lb + (size_t)(b + 1) * n),
        ubv(ub + (size_t)b * n, ub + (size_t)(b + 1) * n), st(HsSol<Sys>::stage_doubles(N));
    if (getenv("POISON")) {   // the device scratch is not zero-initialised: NaN here exposes a read-before-write
      for (auto* v : {&zL, &zU, &dz, &st}) for (auto& x : *v) x = NAN;
      for (int i = 0; i < m; ++i) lam[(size_t)b * m + i] = NAN;
    }
    SysParams<Sys> pp;
    pp.load(params, b, pstride);
    const double* p = pp.get();
    HsWork w{{z + (size_t)b * n, 1}, {lbv.data(), 1}, {ubv.data(), 1}, {zL.data(), 1}, {zU.data(), 1},
             {lam + (size_t)b * m, 1}, {dz.data(), 1}, {st.data(), 1}};
    HsSolveResult r;
    S::solve(w, o, p, r);
    cost[b] = r.cost; status[b] = r.status; iters[b] = r.iters;
    if (kkt) { kkt[3 * b] = r.feas; kkt[3 * b + 1] = r.stat; kkt[3 * b + 2] = getenv("SWEEPS") ? (double)r.sweeps : r.compl_; }
  }
}

// One Newton/SQP step at a given (interior) iterate: for checking the Riccati recursion against a dense KKT solve.
template <class Sys>
static void one_step(int N, double T, double* z, const double* lb, const double* ub, double* zL, double* zU,
                     const double* nuT, double mu, double* lam, double* dz, double* nu_out, double* info) {
  using S = HsSolver<Sys>;
  const int K = 2 * N + 1, n = K * Sys::NW;
  HsSolveOpts o{N, T / N, 1, 1e-8, 1e-6, 1e-7, mu};
  if (getenv("RHO")) o.rho_term = atof(getenv("RHO"));
  if (getenv("REGF")) o.reg_floor = atof(getenv("REGF"));
  std::vector<double> st(HsSol<Sys>::stage_doubles(N)), lbv(lb, lb + n), ubv(ub, ub + n);
  double p[Sys::NPX];
  Sys::default_params(p);
  HsWork w{{z, 1}, {lbv.data(), 1}, {ubv.data(), 1}, {zL, 1}, {zU, 1}, {lam, 1}, {dz, 1}, {st.data(), 1}};
  typename S::SweepOut so;
  so.abort_on_reg = false;
  S::backward(w, o, p, nuT, 0.0, so);
  S::solve_nu(so, mu, nu_out);
  typename S::FwdOut fo;
  S::forward(w, o, p, mu, nu_out, so.term_pinned, fo);
  info[0] = so.f; info[1] = so.c1; info[2] = so.cinf; info[3] = so.stat; info[4] = so.compl_max;
  info[5] = fo.alpha_p; info[6] = fo.alpha_d; info[7] = fo.gphi; info[8] = so.nreg;
}

extern "C" int hostsim_solve(int system_id, int N, double T, int B, double* z, const double* lb, const double* ub,
                             const double* params, int pstride, int max_iter, double tol_feas, double tol_stat,
                             double tol_compl, double mu_init, double* lam, double* cost, int32_t* status,
                             int32_t* iters, double* kkt) {
#define GO(S) solve_batch<S>(N, T, B, z, lb, ub, params, pstride, max_iter, tol_feas, tol_stat, tol_compl, mu_init, lam, cost, status, iters, kkt)
  switch (system_id) {
    case 0: GO(SysCARTPOLE); return 0;
    case 1: GO(SysVANDERPOL); return 0;
    case 2: GO(SysCANCERTREATMENT); return 0;
    case 3: GO(SysSIMPLECASE); return 0;
    case 13: GO(SysPENDULUM); return 0;     // dev: solver traces of the later systems
    case 10: GO(SysEPIDEMICSEIRN); return 0;
  }
  return -1;
}

extern "C" int hostsim_step(int system_id, int N, double T, double* z, const double* lb, const double* ub, double* zL,
                            double* zU, const double* nuT, double mu, double* lam, double* dz, double* nu_out, double* info) {
#define ST(S) one_step<S>(N, T, z, lb, ub, zL, zU, nuT, mu, lam, dz, nu_out, info)
  switch (system_id) {
    case 0: ST(SysCARTPOLE); return 0;
    case 1: ST(SysVANDERPOL); return 0;
    case 2: ST(SysCANCERTREATMENT); return 0;
    case 3: ST(SysSIMPLECASE); return 0;
  }
  return -1;
}

extern "C" double hostsim_rollout(int system_id, int method, int num_steps, double h, int u_rows, const double* x0,
                                  const double* us, const double* params, double* xs) {
#define RL(S) { double p[S::NPX]; S::default_params(p); if (params) for (int i = 0; i < S::NP; ++i) p[i] = params[i]; \
                return Rollout<S>::run(method, num_steps, h, u_rows, x0, us, p, xs); }
  switch (system_id) {
    case 0: RL(SysCARTPOLE)
    case 1: RL(SysVANDERPOL)
    case 2: RL(SysCANCERTREATMENT)
    case 3: RL(SysSIMPLECASE)
  }
  return NAN;
}

// generic batch driver for the one-step cores (trapezoid / shooting)
template <class Core, class Sys>
static void os_batch(const HsSolveOpts& o0, int n, int m, long nst, int B, double* z, const double* lb, const double* ub,
                     const double* params, int pstride, double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt) {
#pragma omp parallel for schedule(dynamic)
  for (int b = 0; b < B; ++b) {
    std::vector<double> zL(n), zU(n), dz(n), lbv(lb + (size_t)b * n, lb + (size_t)(b + 1) * n),
        ubv(ub + (size_t)b * n, ub + (size_t)(b + 1) * n), st(nst);
    SysParams<Sys> pp;
    pp.load(params, b, pstride);
    const double* p = pp.get();
    HsWork w{{z + (size_t)b * n, 1}, {lbv.data(), 1}, {ubv.data(), 1}, {zL.data(), 1}, {zU.data(), 1},
             {lam + (size_t)b * m, 1}, {dz.data(), 1}, {st.data(), 1}};
    HsSolveResult r;
    HsSolveOpts o = o0;
    if (getenv("TAU")) o.tau_min = atof(getenv("TAU"));
    if (getenv("LMABS")) o.lm_abs = atoi(getenv("LMABS"));
    if (getenv("KSIG")) o.kappa_sigma = atof(getenv("KSIG"));
    if (getenv("MU0")) o.mu_init = atof(getenv("MU0"));
    if (getenv("LM")) o.lm_init = atof(getenv("LM"));
    Core::solve(w, o, p, r);
    cost[b] = r.cost; status[b] = r.status; iters[b] = r.iters;
    if (kkt) { kkt[3 * b] = r.feas; kkt[3 * b + 1] = r.stat; kkt[3 * b + 2] = r.compl_; }
  }
}

extern "C" int hostsim_solve_trap(int system_id, int N, double T, int B, double* z, const double* lb, const double* ub,
                                  const double* params, int pstride, int max_iter, double* lam, double* cost,
                                  int32_t* status, int32_t* iters, double* kkt) {
  HsSolveOpts o{N, T / N, max_iter, 1e-8, 1e-6, 1e-7, 0.1};
  apply_env(o);
#define TR(S) os_batch<TrapCore<S>, S>(o, (N + 1) * S::NW, N * S::NS, TrapCore<S>::stage_doubles(N), B, z, lb, ub, params, pstride, lam, cost, status, iters, kkt)
  switch (system_id) {
    case 0: TR(SysCARTPOLE); return 0;
    case 1: TR(SysVANDERPOL); return 0;
    case 2: TR(SysCANCERTREATMENT); return 0;
    case 3: TR(SysSIMPLECASE); return 0;
  }
  return -1;
}

extern "C" int hostsim_solve_shoot(int system_id, int I, int cpi, int method, double T, int B, double* z, const double* lb,
                                   const double* ub, const double* params, int pstride, int max_iter, double* lam,
                                   double* cost, int32_t* status, int32_t* iters, double* kkt) {
  HsSolveOpts o{I, T / I, max_iter, 1e-8, 1e-6, 1e-7, 0.1};
  o.cpi = cpi; o.method = method;
  apply_env(o);
#define SH(S) { if (method == 3) os_batch<ShootCore<S, 2>, S>(o, (I + 1) * S::NS + (2 * I * cpi + 1) * S::NU, I * S::NS, ShootCore<S, 2>::stage_doubles(I, cpi), B, z, lb, ub, params, pstride, lam, cost, status, iters, kkt); \
                else os_batch<ShootCore<S>, S>(o, (I + 1) * S::NS + (I * cpi + 1) * S::NU, I * S::NS, ShootCore<S>::stage_doubles(I, cpi), B, z, lb, ub, params, pstride, lam, cost, status, iters, kkt); }
  switch (system_id) {
    case 0: SH(SysCARTPOLE); return 0;
    case 1: SH(SysVANDERPOL); return 0;
    case 2: SH(SysCANCERTREATMENT); return 0;
    case 3: SH(SysSIMPLECASE); return 0;
  }
  return -1;
}
