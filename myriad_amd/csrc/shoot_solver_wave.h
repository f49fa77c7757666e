// shoot_solver_wave.h -- the shooting solver (ShootCore, os_solver.h) mapped ONE TRAJECTORY PER WAVEFRONT.
//
// The lane-per-trajectory kernel runs a whole shooting solve in one lane: at the batch sizes of the reference's
// workloads (config 3: 8192 trajectories per GPU) that is 8192 busy lanes on a machine that retires 16 k fp64 lanes per
// cycle, and the launch lasts as long as its slowest trajectory (119 iterations x 0.7 ms).  Here a 64-lane workgroup owns
// one trajectory, the whole iterate lives in LDS (variables, bounds, bound multipliers, rollout states, stage
// records, gains -- 18 KB for VANDERPOL 1 x 50, eight workgroups per CU), and the workgroups are persistent (ticket
// counter, as hs_solver_wave.h):
//   * lanes over VARIABLES: starting point, bound terms, step limits, trial point + barrier, update;
//   * lanes over STEPS (64 at a time): linearise -> costates -> step Hessians in one pass -- the costate recursion is affine and
//     runs as a suffix scan of map compositions across the wave (affine_scan, hs_solver_wave.h), after which every lane
//     evaluates its step Hessian with the costate it received; and the forward sweep in one pass -- closed-loop map of the
//     step, prefix scan, the lane's own step and its share of the directional derivative;
//   * lanes over INTERVALS: rollouts (states, continuity defects, objective) of every trial point; the accepted trial point
//     IS the next iterate and its rollout is kept for the next sweep;
//   * sequential over the steps: only the Riccati recursion -- on the matrix cores (three v_mfma_f64_16x16x4_f64 per step,
//     riccati_mfma below) for one-control systems with NS <= 4, else os_riccati_stage (the code of the lane kernel) in
//     one lane out of LDS.  An inertia-correction retry (W + delta I) repeats only that recursion: the rollout, the
//     linearisation and the costates do not depend on delta.
//   * a solve that ends without a KKT point is started again (another initial barrier parameter), see the kernel.
// The algorithm, its constants and its control flow are ShootCore's: the outer loop IS IpLoop<> (hs_solver.h), executed
// redundantly by all 64 lanes on wave-uniform scalars, with this struct as its `Core`.  Only the association of the sums
// (objective, defect norms, barrier, directional derivative: wave reductions instead of one running sum) and of the two
// recursions (compositions of affine maps instead of step-by-step application) differ from the lane kernel, against which -- and against
// the same oracle / golden solutions -- it is tested.
//
// Replaces, per trajectory, IPOPT on /root/reference/myriad/trajectory_optimizers/shooting.py:169-241 (objective :169-210,
// constraints :230-241) as called from nlp_solvers/__init__.py:32-96.
#pragma once
#include <hip/hip_runtime.h>
#include "os_solver.h"
#include "hs_solver_wave.h"

namespace myriad {

typedef __attribute__((address_space(3))) double sw_lds;     // LDS-typed pointers: ds_read / ds_write instead of flat accesses

#ifndef MYR_SHOOT_RESTARTS
#define MYR_SHOOT_RESTARTS 2       // default of myr_solve_opts.restarts for this kernel: second / third start of a failed solve
#endif
#ifndef MYR_SHOOT_MIN_WAVES
#define MYR_SHOOT_MIN_WAVES 2     // waves per SIMD the register allocation must allow (LDS allows 8 workgroups per CU)
#endif

template <class Sys, int M = 1>
struct ShootWave {
  using SC = ShootCore<Sys, M>;
  using H = HsSolver<Sys>;
  using D = OsDims<Sys, M>;
  static constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NY = D::NY, NQ = D::NQ, NC = D::NC, NY1 = D::NY1, QN = D::QN;
  static constexpr bool PEN_LAM_FLOOR = SC::PEN_LAM_FLOOR;
  using SweepOut = typename H::SweepOut;
  using FwdOut = typename H::FwdOut;

  // stage record: gy (NY), Fy | c (NS x NY1, the step map dx_next = Fy y + c), Hs (NY x NY, lower triangle packed)
  static constexpr int HSP = NY * (NY + 1) / 2;
  static constexpr int R_GY = 0, R_GE = R_GY + NY, R_HS = R_GE + NS * NY1, REC = R_HS + HSP;
  __host__ __device__ static constexpr int hsp(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
  static constexpr int KG = NQ * NW + NQ * NC;        // gains K | kc of a stage
  // exchange block (first in LDS): results of the one-lane phases and of the last linearisation, for all lanes
  static constexpr int X_VALID = 0, X_F = 1, X_C1 = 2, X_CINF = 3, X_STAT = 4, X_CMAX = 5, X_CMIN = 6, X_LAMINF = 7, X_SUMMULT = 8,
                       X_NMULT = 9, X_NREG = 10, X_GPHI = 11, X_TNU = 12, X_TP = X_TNU + NS * NC,
                       X_T = X_TP + NS /* 8 phase timers (developer knob MYR_SW_TIMING) */,
                       X_Z = X_T + 8 /* a zero and a write-only slot */, X_P = X_Z + 2, X_PC = X_P + NW * NW,
                       // the last trial point: step length, objective, defect norms; 1 if its rollout is the current iterate's;
                       // problem dimensions for the phases that are not handed the options (update)
                       X_TA = X_PC + NW * NC, X_TF = X_TA + 1, X_TC1 = X_TF + 1, X_TCINF = X_TC1 + 1, X_ROLL = X_TCINF + 1,
                       X_DN = X_ROLL + 1, X_DCPI = X_DN + 1, X_SCR = X_DCPI + 1 /* network systems: this slot's global scratch (a pointer) */,
                       X_N = (X_SCR + 1 + 7) / 8 * 8;

  // ---- network dynamics (node_system.h) on the matrix-core passes of node_mfma.h (round 6) --------------------------------------------------
  // The per-lane form of the network (SysNODE::f / lin / hessian: every lane walks the 64 x 64 layers of ITS point with the hidden vectors in
  // private memory) made a shooting solve of the network system thousands of times slower than a collocation solve.  Here the evaluations of a
  // phase are gathered into point lists and run as tiles of 16 points, as in the collocation kernel:
  //   * rollouts: the steps of an interval are sequential, the intervals are not -- per step and stage ONE value pass (MODE 0) over the stage
  //     points of all intervals (lane k = interval k);
  //   * linearisation: after the rollout every step's first stage point is known; per stage one MODE 3 pass over the S steps (value, A, B into
  //     global records; the hidden activations and tangents stay in the slot's scratch for MODE 4), the next stage's points follow from the
  //     records; the step algebra (os_solver.h: step_lin / rk4_lin) then reads the records through an evaluator instead of calling the system;
  //   * second derivatives: a dry run of the step algebra with the costates records the stage weights (the contraction vectors), one MODE 4
  //     pass per stage contracts the network's second derivatives with them, a last run of the step algebra assembles the step Hessians.
  static constexpr bool MLP = NodeTraits<Sys>::mlp;
  static constexpr int NSTM = (M == 2) ? 4 : 2;                    // stage points per step at most
  static constexpr int PT_F = 0, PT_A = PT_F + NS, PT_B = PT_A + NS * NS, PT_D2 = PT_B + NS * NU, PT_N = PT_D2 + NW * (NW + 1) / 2;
  __host__ __device__ static long mlp_kmax(long S) { return S > 64 ? S : 64; }
  __host__ __device__ static long mlp_lds_doubles(long S) {       // weights | point list | values (one rollout pass) | costates per step | stage weights
    return MLP ? (long)NodeTraits<Sys>::lds_doubles + NW * mlp_kmax(S) + NS * 64 + S * NS + (long)NSTM * S * NS : 0;
  }
  __host__ __device__ static long mlp_scratch_doubles(int I, int cpi) {      // per resident workgroup (global): records, activations, tangents of every stage
    const long S = (long)I * cpi, nt = (S + 15) / 16;
    return MLP ? (long)NSTM * (PT_N * S + nt * (NodeMfma64::HB_TILE + NodeMfma64::MB_TILE)) : 0;
  }

  __host__ __device__ static inline int steps(const HsSolveOpts& o) { return o.N * o.cpi; }
  __host__ __device__ static inline int nvars(const HsSolveOpts& o) { return (o.N + 1) * NS + (M * steps(o) + 1) * NU; }
  __host__ __device__ static inline long xi(int k, int c) { return SC::xi(k, c); }
  __host__ __device__ static inline long ui(const HsSolveOpts& o, int i, int a) { return SC::ui(o, i, a); }
  // rollout states of the iterate, rollout states of the last trial point, the trial point
  __host__ __device__ static long tmp_doubles(long S, long n) { return 2 * (S + 1) * NS + n; }
  __host__ __device__ static long lds_doubles(int I, int cpi) {
    const long S = (long)I * cpi, n = (long)(I + 1) * NS + (M * S + 1) * NU;
    return X_N + 9 * n + tmp_doubles(S, n) + S * (REC + KG) + NU * NC + 2 * (long)I * NS + mlp_lds_doubles(S);
  }
  __host__ __device__ static size_t lds_bytes(int I, int cpi) { return (size_t)lds_doubles(I, cpi) * 8; }

  struct Lds {
    sw_lds *ex, *z, *lb, *ub, *zL, *zU, *dz, *sig, *g1, *rec, *kg, *ku, *lam;
    sw_lds *z0;              // the caller's starting point (restart after a failed solve, see the kernel)
    sw_lds *zlu;             // = dz  (bound-multiplier difference, dead before the step is written)
    sw_lds *xs, *xt, *zt;    // rollout states of the iterate / of the last trial point, the trial point
    sw_lds *ct;              // continuity defects of the last trial point
    sw_lds *wl, *pts, *sF, *pinS, *avec;      // network systems: weights, point list, values of a rollout pass, costates per step, stage weights
  };
  __device__ static inline sw_lds* lds_base() {
    extern __shared__ __attribute__((aligned(16))) char smem_wave[];
    return (sw_lds*)reinterpret_cast<double*>(smem_wave);
  }
  __device__ static inline Lds lds(const HsSolveOpts& o) {
    const long S = steps(o), n = nvars(o);
    Lds l;
    sw_lds* s = lds_base();
    l.ex = s; s += X_N;
    l.z = s; s += n; l.lb = s; s += n; l.ub = s; s += n; l.zL = s; s += n; l.zU = s; s += n; l.dz = s; s += n;
    l.sig = s; s += n; l.g1 = s; s += n; l.z0 = s; s += n;
    l.zlu = l.dz;
    l.xs = s; l.xt = s + (S + 1) * NS; l.zt = s + 2 * (S + 1) * NS; s += tmp_doubles(S, n);
    l.rec = s; s += S * REC; l.kg = s; s += S * KG; l.ku = s; s += NU * NC; l.lam = s; s += steps(o) / o.cpi * NS; l.ct = s; s += steps(o) / o.cpi * NS;
    l.wl = s; s += NodeTraits<Sys>::lds_doubles; l.pts = s; s += NW * mlp_kmax(S); l.sF = s; s += NS * 64; l.pinS = s; s += S * NS; l.avec = s;
    return l;
  }
#ifdef MYR_SW_TIMING
#define MYR_SWT(k) { const long long t1_ = wall_clock64(); if (threadIdx.x == 0) lds_base()[X_T + k] += (double)(t1_ - t0_); t0_ = t1_; }
#define MYR_SWT0 long long t0_ = wall_clock64();
#define MYR_SWT0F MYR_SWT0
#define MYR_SWTF(k) MYR_SWT(k)
#else
#define MYR_SWT0F
#define MYR_SWTF(k)
#define MYR_SWT(k)
#define MYR_SWT0
#endif

  // ---- lanes over variables: starting point and accepted step (HsSolver::init / update, one variable per lane) ----
  __device__ static void init(const HsWork& w, int n) {
    const double k1 = 1e-2, k2 = 1e-2;
    sw_lds* z = (sw_lds*)w.z.p; sw_lds* lb = (sw_lds*)w.lb.p; sw_lds* ub = (sw_lds*)w.ub.p; sw_lds* zL = (sw_lds*)w.zL.p; sw_lds* zU = (sw_lds*)w.zU.p;
    for (int i = threadIdx.x; i < n; i += 64) {
      const double l = lb[i], u = ub[i], v0 = z[i];
      const bool fr = l < u;
      const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
      const double width = (hl && hu) ? (u - l) : INFINITY;
      const double pl = detail::dmin(k1 * detail::dmax(1.0, fabs(l)), k2 * width);
      const double pu = detail::dmin(k1 * detail::dmax(1.0, fabs(u)), k2 * width);
      double v = v0;
      v = hl ? detail::dmax(v, l + pl) : v;
      v = hu ? detail::dmin(v, u - pu) : v;
      v = fr ? v : l;
      z[i] = v;
      zL[i] = hl ? 1.0 : 0.0;
      zU[i] = hu ? 1.0 : 0.0;
    }
    if (threadIdx.x == 0) { lds_base()[X_VALID] = 0.0; lds_base()[X_ROLL] = 0.0; }
    __syncthreads();
  }
  // The accepted step is the last trial point of the line search: its values ARE the new iterate (bit for bit), and its
  // rollout -- states, defects, objective -- is kept for the next sweep instead of being integrated again.
  __device__ static void update(const HsWork& w, int n, double ap, double ad, double mu, double ksig) {
    const double iks = 1.0 / ksig;
    sw_lds* ex = lds_base();
    HsSolveOpts od; od.N = (int)ex[X_DN]; od.cpi = (int)ex[X_DCPI];
    const Lds l = lds(od);
    const bool reuse = ex[X_TA] == ap && ap > 0.0;
    for (int i = threadIdx.x; i < n; i += 64) {
      const double lo = l.lb[i], u = l.ub[i], zv = l.z[i], d = l.dz[i], zl = l.zL[i], zu = l.zU[i];
      const bool fr = lo < u;
      const bool hl = fr && (lo > -INFINITY), hu = fr && (u < INFINITY);
      const double zn = fr ? (reuse ? l.zt[i] : zv + ap * d) : zv;
      const double sl = hl ? zv - lo : 1.0, su = hu ? u - zv : 1.0;
      const double snl = hl ? zn - lo : 1.0, snu = hu ? u - zn : 1.0;
      double vl = zl + ad * (-zl + (mu - zl * d) / sl);
      double vu = zu + ad * (-zu + (mu + zu * d) / su);
      const double ml = mu / snl, mu_ = mu / snu;
      vl = detail::dmax(detail::dmin(vl, ksig * ml), ml * iks);
      vu = detail::dmax(detail::dmin(vu, ksig * mu_), mu_ * iks);
      l.z[i] = zn;
      l.zL[i] = hl ? vl : 0.0;
      l.zU[i] = hu ? vu : 0.0;
    }
    if (reuse) {
      const int S = steps(od);
      for (int i = threadIdx.x; i < S * NS; i += 64) l.xs[i] = l.xt[i];
      for (int i = threadIdx.x; i < od.N * NS; i += 64) l.lam[i] = l.ct[i];
    }
    if (threadIdx.x == 0) { ex[X_VALID] = 0.0; ex[X_ROLL] = reuse ? 1.0 : 0.0; }
    __syncthreads();
  }
  __device__ static void solve_nu(const SweepOut& so, double mu, double* nu) { H::solve_nu(so, mu, nu); }

  // one interval's rollout from x (values of the variables in `v`): states -> xs (if given), end state -> x.
  // The control rows of step i+1 are fetched before step i is integrated (the chain of dependent steps never waits for LDS).
  __device__ static inline void roll_interval(const HsSolveOpts& o, const double* p, const sw_lds* v, int k, double* x, sw_lds* xs, double& f) {
    const int cpi = o.cpi, S = steps(o);
    const double h = SC::hstep(o);
    const int i0 = k * cpi, i1 = (k + 1) * cpi;
    double uc[(M + 1) * NU], un[M * NU];
#pragma unroll
    for (int a = 0; a < NU; ++a) uc[M * NU + a] = v[ui(o, M * i0, a)];
#pragma unroll
    for (int a = 0; a < M * NU; ++a) un[a] = v[ui(o, M * i0 + 1, a)];
    for (int i = i0; i < i1; ++i) {
      double xn[NS], dc;
#pragma unroll
      for (int a = 0; a < NU; ++a) uc[a] = uc[M * NU + a];                 // the previous step's last row
#pragma unroll
      for (int a = 0; a < M * NU; ++a) uc[NU + a] = un[a];
      if (i + 1 < i1) {
#pragma unroll
        for (int a = 0; a < M * NU; ++a) un[a] = v[ui(o, M * (i + 1) + 1, a)];
      }
      if (xs) {
#pragma unroll
        for (int c = 0; c < NS; ++c) xs[(long)i * NS + c] = x[c];
      }
      SC::sval(o.method, h, x, uc, p, xn, dc, h * i, i == S - 1);
      f += dc;
#pragma unroll
      for (int c = 0; c < NS; ++c) x[c] = xn[c];
    }
  }

  // ---- network systems: the phases on the matrix-core passes ------------------------------------------------------------------------------
  struct MlpScr { nd_glb *pt, *hb, *mb; long nt; };
  __device__ static inline MlpScr mlp_scr(const Lds& l, long S) {
    MlpScr m;
    nd_glb* g = (nd_glb*)(double*)__builtin_bit_cast(unsigned long long, (double)l.ex[X_SCR]);
    m.nt = (S + 15) / 16;
    m.pt = g; m.hb = g + (long)NSTM * PT_N * S; m.mb = m.hb + (long)NSTM * m.nt * NodeMfma64::HB_TILE;
    return m;
  }
  // f at the K points of the list l.pts (states [K][NS], then the controls [K]) -> l.sF[k * NS + r]
  __device__ static inline void mlp_values(const Lds& l, int K) {
    if constexpr (MLP) {
      NodeMfma64::ArgsT<nd_lds> a;
      a.z = (const nd_lds*)l.pts; a.dz = (const nd_lds*)l.pts; a.lam = nullptr; a.pt = nullptr; a.sF = (nd_lds*)l.sF;
      a.alpha = 0.0; a.h6 = 0.0; a.h8 = 0.0; a.K = K; a.N = 0; a.pf_f = 0; a.pf_a = 0; a.pf_b = 0; a.pf_d2 = 0;
      a.t0 = 0; a.ts = 1; a.hb = nullptr; a.mb = nullptr; a.h_valid = 0;
      __syncthreads();
      NodeMfma64::pass<0, nd_lds>((const nd_lds*)l.wl, a, (int)threadIdx.x);
      __syncthreads();
    } else { (void)l; (void)K; }
  }
  // Rollouts of ALL intervals from the variables in `v`: step states -> xs, continuity defects -> dfc, objective and defect norms summed.  Lane k
  // rolls interval k out; the network is evaluated for the stage points of all lanes together (mlp_values).  The step formulas are step_val's /
  // rk4_val's (os_solver.h; utils.py:31-54).
  __device__ static void roll_all_mlp(const HsSolveOpts& o, const double* p, const Lds& l, const sw_lds* v, sw_lds* xs, sw_lds* dfc, double& f, double& c1, double& cinf) {
    if constexpr (MLP) {
      const int I = o.N, cpi = o.cpi, lane = threadIdx.x;
      const double h = SC::hstep(o);
      for (int kb = 0; kb < I; kb += 64) {
        const int K = (I - kb) < 64 ? (I - kb) : 64;
        const bool on = lane < K;
        const int k = on ? kb + lane : kb;
        double x[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) x[c] = v[xi(k, c)];
        auto F = [&](const double* X, const double* U, double* out) {      // (every lane calls it: the pass is the whole wavefront's)
          if (on) {
#pragma unroll
            for (int c = 0; c < NS; ++c) l.pts[lane * NS + c] = X[c];
            l.pts[K * NS + lane] = U[0];
          }
          mlp_values(l, K);
#pragma unroll
          for (int c = 0; c < NS; ++c) out[c] = l.sF[(on ? lane : 0) * NS + c];
        };
        for (int s2 = 0; s2 < cpi; ++s2) {
          const int i = k * cpi + s2;
          double uc[(M + 1) * NU];
#pragma unroll
          for (int a = 0; a < (M + 1) * NU; ++a) uc[a] = v[ui(o, M * i, a)];
          if (on) {
#pragma unroll
            for (int c = 0; c < NS; ++c) xs[(long)i * NS + c] = x[c];
          }
          double xn[NS], dc;
          if constexpr (M == 2) {
            const double* U[4] = {uc, uc + NU, uc + NU, uc + 2 * NU};
            const double aj[4] = {0.0, 0.5, 0.5, 1.0}, bj[4] = {1.0 / 6.0, 2.0 / 6.0, 2.0 / 6.0, 1.0 / 6.0};
            double kk[NS], X[NS], acc[NS], g = 0.0;
#pragma unroll
            for (int c = 0; c < NS; ++c) { acc[c] = 0.0; kk[c] = 0.0; }
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int c = 0; c < NS; ++c) X[c] = x[c] + aj[j] * h * kk[c];
              F(X, U[j], kk);
              g += bj[j] * Sys::g(X, U[j], p);
#pragma unroll
              for (int c = 0; c < NS; ++c) acc[c] += bj[j] * kk[c];
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) xn[c] = x[c] + h * acc[c];
            dc = h * g;
          } else {
            const double* u = uc; const double* un = uc + NU;
            double f1[NS];
            F(x, u, f1);
            const double g1 = Sys::g(x, u, p);
            if (o.method == 0) {          // (wave-uniform)
#pragma unroll
              for (int c = 0; c < NS; ++c) xn[c] = x[c] + h * f1[c];
              dc = h * g1;
            } else {
              double xt[NS], f2[NS], u2[NU];
#pragma unroll
              for (int c = 0; c < NS; ++c) xt[c] = x[c] + h * f1[c];
              const bool mid = (o.method == 2);
#pragma unroll
              for (int a = 0; a < NU; ++a) u2[a] = mid ? 0.5 * (u[a] + un[a]) : un[a];
              F(xt, u2, f2);
              const double g2 = Sys::g(xt, u2, p);
              if (mid) {
#pragma unroll
                for (int c = 0; c < NS; ++c) xn[c] = x[c] + h * f2[c];
                dc = h * g2;
              } else {
#pragma unroll
                for (int c = 0; c < NS; ++c) xn[c] = x[c] + 0.5 * h * (f1[c] + f2[c]);
                dc = 0.5 * h * (g1 + g2);
              }
            }
          }
          if (on) f += dc;
#pragma unroll
          for (int c = 0; c < NS; ++c) x[c] = xn[c];
        }
        if (on) {
#pragma unroll
          for (int c = 0; c < NS; ++c) {
            const double ck = x[c] - v[xi(k + 1, c)];                          // shooting.py:239-241
            dfc[(long)k * NS + c] = ck;
            c1 += fabs(ck);
            cinf = detail::dmax(cinf, fabs(ck));
          }
        }
      }
    } else { (void)o; (void)p; (void)l; (void)v; (void)xs; (void)dfc; (void)f; (void)c1; (void)cinf; }
  }
  // what the step algebra reads instead of calling the system (os_solver.h: SysEval): stage j of step i from the records of the passes
  struct RecEval {
    const nd_glb* pt; sw_lds* avec; long S; int i, mode;      // mode 0: first derivatives; 1: + record the stage weights; 2: + second derivatives from the records
    __device__ inline void lin(int j, HsPoint<Sys>& P, const double* p) const {
      const nd_glb* r = pt + (long)j * PT_N * S + i;
#pragma unroll
      for (int c = 0; c < NS; ++c) P.f[c] = r[(long)(PT_F + c) * S];
#pragma unroll
      for (int q = 0; q < NS * NS; ++q) P.A[q] = r[(long)(PT_A + q) * S];
#pragma unroll
      for (int q = 0; q < NS * NU; ++q) P.B[q] = r[(long)(PT_B + q) * S];
      Sys::cost_grad(P.x, P.u, p, &P.g, P.gw);
    }
    __device__ inline void hess(int j, const HsPoint<Sys>& P, const double* p, const double* mu, double w, double* W) const {
      (void)p;
      if (mode == 1) {
#pragma unroll
        for (int c = 0; c < NS; ++c) avec[((long)j * S + i) * NS + c] = mu[c];
      }
      if (mode == 2) {
        double Wm[NW * (NW + 1) / 2];
        const nd_glb* r = pt + (long)j * PT_N * S + i;
#pragma unroll
        for (int e = 0; e < NW * (NW + 1) / 2; ++e) Wm[e] = r[(long)(PT_D2 + e) * S];
        Sys::hessian_packed(P.x, P.u, Wm, w, W);
      } else {
#pragma unroll
        for (int q = 0; q < NW * NW; ++q) W[q] = 0.0;
      }
    }
  };
  __device__ static inline int mlp_stages(const HsSolveOpts& o) { return M == 2 ? 4 : (o.method == 0 ? 1 : 2); }
  // value + first derivatives at every stage point of every step (the iterate's rollout states are in l.xs): one MODE 3 pass per stage
  __device__ static void mlp_lin_passes(const HsSolveOpts& o, const Lds& l) {
    if constexpr (MLP) {
      const int S = steps(o), lane = threadIdx.x;
      const double h = SC::hstep(o);
      const MlpScr m = mlp_scr(l, S);
      const int nst = mlp_stages(o);
      for (int j = 0; j < nst; ++j) {
        for (int i0 = 0; i0 < S; i0 += 64) {
          const int i = i0 + lane < S ? i0 + lane : S - 1;
          double X[NS], U[NU];
#pragma unroll
          for (int c = 0; c < NS; ++c) X[c] = l.xs[(long)i * NS + c];
          if (j > 0) {       // X_j = x + a_j h k_{j-1}  (rk4_lin) / x + h f_1 (step_lin), with k from the previous stage's records
            const double aj = (M == 2) ? (j == 3 ? 1.0 : 0.5) : 1.0;
            const nd_glb* r = m.pt + (long)(j - 1) * PT_N * S + i;
#pragma unroll
            for (int c = 0; c < NS; ++c) X[c] = X[c] + aj * h * r[(long)(PT_F + c) * S];
          }
          if constexpr (M == 2) {
            const int row = (j == 0) ? 0 : (j == 3 ? 2 : 1);
#pragma unroll
            for (int a = 0; a < NU; ++a) U[a] = l.z[ui(o, M * i + row, a)];
          } else {
            const bool mid = (o.method == 2);
            const double cu = (j == 0) ? 1.0 : (mid ? 0.5 : 0.0), cun = (j == 0) ? 0.0 : (mid ? 0.5 : 1.0);
#pragma unroll
            for (int a = 0; a < NU; ++a) U[a] = cu * l.z[ui(o, i, a)] + cun * l.z[ui(o, i + 1, a)];
          }
          if (i0 + lane < S) {
#pragma unroll
            for (int c = 0; c < NS; ++c) l.pts[(long)i * NS + c] = X[c];
            l.pts[(long)S * NS + i] = U[0];
          }
        }
        __syncthreads();
        NodeMfma64::ArgsT<nd_lds> a;
        a.z = (const nd_lds*)l.pts; a.dz = (const nd_lds*)l.pts; a.lam = nullptr; a.pt = m.pt + (long)j * PT_N * S; a.sF = (nd_lds*)l.sF;
        a.alpha = 0.0; a.h6 = 0.0; a.h8 = 0.0; a.K = S; a.N = 0; a.pf_f = PT_F; a.pf_a = PT_A; a.pf_b = PT_B; a.pf_d2 = PT_D2;
        a.t0 = 0; a.ts = 1; a.hb = m.hb + (long)j * m.nt * NodeMfma64::HB_TILE; a.mb = m.mb + (long)j * m.nt * NodeMfma64::MB_TILE; a.h_valid = 0;
        NodeMfma64::pass<3, nd_lds>((const nd_lds*)l.wl, a, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
      }
    } else { (void)o; (void)l; }
  }
  // the network's second derivatives contracted with the stage weights in l.avec: one MODE 4 pass per stage
  __device__ static void mlp_hess_passes(const HsSolveOpts& o, const Lds& l) {
    if constexpr (MLP) {
      const int S = steps(o), lane = threadIdx.x;
      const MlpScr m = mlp_scr(l, S);
      const int nst = mlp_stages(o);
      __syncthreads();
      for (int j = 0; j < nst; ++j) {
        NodeMfma64::ArgsT<nd_lds> a;
        a.z = (const nd_lds*)l.pts; a.dz = (const nd_lds*)l.pts; a.lam = nullptr; a.pt = m.pt + (long)j * PT_N * S; a.sF = (nd_lds*)l.sF;
        a.alpha = 0.0; a.h6 = 0.0; a.h8 = 0.0; a.K = S; a.N = 0; a.pf_f = PT_F; a.pf_a = PT_A; a.pf_b = PT_B; a.pf_d2 = PT_D2;
        a.t0 = 0; a.ts = 1; a.hb = m.hb + (long)j * m.nt * NodeMfma64::HB_TILE; a.mb = m.mb + (long)j * m.nt * NodeMfma64::MB_TILE; a.h_valid = 1;
        a.avec = (const nd_lds*)(l.avec + (long)j * S * NS);
        NodeMfma64::pass<4, nd_lds>((const nd_lds*)l.wl, a, lane);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
    } else { (void)o; (void)l; }
  }

  // ---- Riccati recursion on the matrix cores (one control, NS <= 4, one control row per step) --------------------------
  // The stage algebra of os_riccati_stage as three v_mfma_f64_16x16x4_f64 per step -- the sweep of the trapezoidal wavefront
  // solver (HsWave<Sys, 1>::riccati_mfma_trap: same tile slots, same chaining of the products through the result
  // registers, same pivot and gain rule) with the step map Fy | c in the place of the eliminated collocation rows and
  // the FULL step Hessian Hs | gy as the C operand of the second product (where Hermite-Simpson feeds its midpoint terms).
  // Own (bound) terms are diagonal here: sigma + delta of the control row of every point, of the state rows at nodes only.
  static constexpr bool MFMA_RICCATI = (M == 1 && NU == 1 && NS <= 4);
  using HW = HsWave<Sys, NodeTraits<Sys>::mlp ? 0 : 1>;      // (only its lane-movement helpers are used; the trapezoidal form of HsWave is not built for network dynamics)
  typedef double mfma_d4 __attribute__((ext_vector_type(4)));
#ifndef MYR_SHOOT_RICCATI_PF
#define MYR_SHOOT_RICCATI_PF 2
#endif
  __device__ static int riccati_mfma(const Lds& l, const HsSolveOpts& o, double delta, bool abort_on_reg, bool& aborted) {
    using namespace detail;
    const int lane = threadIdx.x, S = steps(o), cpi = o.cpi;
    const int g = lane >> 4, j = lane & 15;
    const int scol = j < 4 ? (j < NS ? j : -1) : (j < 6 ? NS : -1);
    const int ycol = scol >= 0 ? scol : ((j == 8 || j == 9) ? NW : -1);
    const int cc = j == 6 ? 0 : (j == 7 ? 1 : (j == 10 ? 2 : (j == 11 ? 3 : (j == 14 ? 4 : (j == 15 ? 5 : -1)))));
    const int rcc = (cc >= 0 && cc < NC) ? cc : -1;
    const bool rowx = g < NS;
    const sw_lds* zr = l.ex + X_Z;
    // per-lane input streams of step i (whose end point is point i+1):
    //   0 own terms of the control row of point i+1 (row du, lane groups 0, 1)   1 Fy | c   2..4 Hs | gy rows dx, du, du_next
    // and, at nodes only, the own terms of the node's state rows
    const sw_lds* ptr[5]; int stp[5];
    ptr[0] = (g < 2 && scol == NS) ? l.sig + ui(o, S, 0) : ((g < 2 && rcc == 1) ? l.g1 + ui(o, S, 0) : zr);
    stp[0] = (ptr[0] == zr) ? 0 : NU;
    const sw_lds* rec = l.rec + (long)(S - 1) * REC;
    ptr[1] = !rowx ? zr : (ycol >= 0 ? rec + R_GE + g * NY1 + ycol : (rcc == 0 ? rec + R_GE + g * NY1 + NY : zr));
    auto hsel = [&](int row, bool on) -> const sw_lds* {
      if (!on) return zr;
      if (ycol >= 0) return rec + R_HS + hsp(row, ycol);
      if (rcc == 0) return rec + R_GY + row;
      return zr;
    };
    ptr[2] = hsel(g, rowx); ptr[3] = hsel(NS, g < 2); ptr[4] = hsel(NW, g < 2);
#pragma unroll
    for (int q = 1; q < 5; ++q) stp[q] = (ptr[q] == zr) ? 0 : REC;
    const sw_lds* nptr = (rowx && scol == g) ? l.sig + g : ((rowx && rcc == 1) ? l.g1 + g : zr);   // + NS * node index
    const int nstr = (nptr == zr) ? 0 : NS;
    const bool pinr = rowx && (l.ex[X_TP + (rowx ? g : 0)] != 0.0);
    const double X0i = (pinr && scol == g) ? o.rho_term - delta : ((pinr && rcc == 2 + g) ? 1.0 : 0.0);
    const double dv0 = (rowx && scol == g) ? delta : 0.0, dv1 = (g < 2 && scol == NS) ? delta : 0.0;
    const double f_a1 = j < 6 ? 1.0 : 0.0, f_keep = rcc >= 0 ? 1.0 : 0.0, f_she = (j == 8 || j == 9) ? 1.0 : 0.0;
    const double f_x1 = g < 2 ? 1.0 : 0.0, f_t1 = g == 2 ? 1.0 : 0.0, f_t23 = g >= 2 ? 1.0 : 0.0;
    const double f_a3 = (g == 0 && (j < 6 || j == 10 || j == 11 || j == 14 || j == 15)) ? -1.0 : 0.0;
    const int k_off = (g == 0 && scol >= 0 && j != 5) ? scol : ((g == 0 && rcc >= 0) ? NQ * NW + rcc : -1);
    sw_lds* k_ptr = k_off >= 0 ? l.kg + (long)(S - 1) * KG + k_off : l.ex + X_Z + 1;
    const int k_step = k_off >= 0 ? KG : 0;
    const double reg_floor = o.reg_floor;
    int nreg = 0;
    aborted = false;
    constexpr int PF = MYR_SHOOT_RICCATI_PF;
    double in[PF][5];
#pragma unroll
    for (int u = 0; u < PF; ++u) {         // (steps below 0 are read too -- valid LDS in front of the records -- and never used)
#pragma unroll
      for (int q = 0; q < 5; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
    }
    mfma_d4 D3 = {X0i, 0.0, 0.0, 0.0};
    int to_node = 0, node = o.N;               // steps until the end point is a node again (no integer division in the loop)
    for (int ib = S - 1; ib >= 0; ib -= PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int i = ib - u;
        if (i < 0) break;
        double own0 = 0.0;
        if (to_node == 0) { own0 = nptr[node * nstr] + dv0; to_node = cpi; --node; }      // wave-uniform: the end point is a node
        --to_node;
        const double X0 = D3[0] + own0, X1 = fma(D3[1], f_x1, in[u][0] + dv1);
        const double G = in[u][1], H0 = in[u][2], H1 = in[u][3], H2 = in[u][4];
#pragma unroll
        for (int q = 0; q < 5; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
        const double sh0 = HW::dpp_row_shr4(X0), sh1 = HW::dpp_row_shr4(X1);
        mfma_d4 C1;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        C1[2] = 0.0; C1[3] = 0.0;
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
        mfma_d4 C2;
        C2[0] = H0; C2[1] = fma(D3[1], f_t1, H1); C2[2] = fma(D3[2], f_t23, D1[1]) + H2; C2[3] = D3[3] * f_t23;
        const mfma_d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(G, D1[0], C2, 0, 0, 0);
        const double q11 = HW::rdlane(D2[2], 8);
        const bool bad = !(q11 > reg_floor);                           // wave-uniform, rare (same pivot rule as chol_reg)
        if (bad) {
          ++nreg;
          if (abort_on_reg) { aborted = true; return nreg; }
        }
        const double d = bad ? dmax(fabs(q11), reg_floor) : q11;
        const double kk = D2[2] * fast_rcp(d);
        k_ptr[0] = kk;
        k_ptr -= k_step;
        const double A3 = D2[2] * f_a3;
        const double B3 = g == 0 ? kk : 0.0;
        D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, B3, D2, 0, 0, 0);
      }
    }
    const double X0 = D3[0], X1 = D3[1], T1 = D3[1], T2 = D3[2], T3 = D3[3];
    sw_lds* sP = l.ex + X_P; sw_lds* sPc = l.ex + X_PC; sw_lds* sTnu = l.ex + X_TNU;
    if (scol >= 0 && j != 5) {
      if (rowx) sP[g * NW + scol] = X0;
      if (g == 0) sP[NS * NW + scol] = X1;
    }
    if (rcc >= 0) {
      if (rowx) sPc[g * NC + rcc] = X0;
      if (g == 0) sPc[NS * NC + rcc] = X1;
      if (g >= 2 && g - 2 < NS) sTnu[(g - 2) * NC + rcc] = T2;
      if (g >= 2 && g < NS) sTnu[g * NC + rcc] = T3;
    }
    __syncthreads();
    if (g == 2 && rcc >= 2) sTnu[(rcc - 2) * NC + 0] += T1;          // row 6: c^T pc'[:, nu_i], summed over the steps
    __syncthreads();
    return nreg;
  }

  // ---- backward sweep (ShootCore::backward): linearisation at the current iterate (once per iterate), Riccati recursion ----
  __device__ static void backward(const HsWork& w, const HsSolveOpts& o, const double* p, const double* nuT, double delta, SweepOut& so) {
    using namespace detail;
    (void)w;
    const Lds l = lds(o);
    const int lane = threadIdx.x, I = o.N, cpi = o.cpi, S = I * cpi, n = nvars(o);
    const double h = SC::hstep(o);
    MYR_SWT0
    if (l.ex[X_VALID] == 0.0) {
      // own (bound) terms of every variable
      double cmax = 0, cmin = INFINITY;
      for (int v = lane; v < n; v += 64) {
        typename H::BV b = H::bound_terms(l.z[v], l.lb[v], l.ub[v], l.zL[v], l.zU[v], cmax, cmin);
        l.sig[v] = b.sigma; l.g1[v] = b.g1; l.zlu[v] = b.zlu;
        // terminal state pinned?  Decided HERE, lane-divergently: as a wave-uniform select in the one-lane phase below,
        // ROCm 7.2's hipcc lowered `pinned ? 1.0 : 0.0` to v_cmp_nlt_f64 vcc ; s_cselect_b32 -- a select on SCC, which the
        // compare does not write (tools/dev/scan_scc.py finds the pattern in a listing)
        if (v >= xi(I, 0) && v < xi(I, 0) + NS) l.ex[X_TP + (v - xi(I, 0))] = b.pinned ? 1.0 : 0.0;
      }
      // rollouts: states at every step, continuity defects (parked in lam), objective -- unless the accepted trial point's
      // rollout is this iterate's (update())
      double f = 0, c1 = 0, cinf = 0;
      const bool have_roll = l.ex[X_ROLL] != 0.0;
      if (!have_roll) {
        if constexpr (MLP) {
          __syncthreads();
          roll_all_mlp(o, p, l, l.z, l.xs, l.lam, f, c1, cinf);
        } else
        for (int k = lane; k < I; k += 64) {
          double x[NS], xe[NS];
#pragma unroll
          for (int c = 0; c < NS; ++c) { x[c] = l.z[xi(k, c)]; xe[c] = l.z[xi(k + 1, c)]; }
          roll_interval(o, p, l.z, k, x, l.xs, f);
#pragma unroll
          for (int c = 0; c < NS; ++c) {
            const double ck = x[c] - xe[c];                                  // shooting.py:239-241
            l.lam[(long)k * NS + c] = ck;
            c1 += fabs(ck);
            cinf = dmax(cinf, fabs(ck));
          }
        }
      }
      f = wv_sum(f); c1 = wv_sum(c1); cinf = wv_max(cinf); cmax = wv_max(cmax); cmin = wv_min(cmin);
      if (have_roll) { f = l.ex[X_TF]; c1 = l.ex[X_TC1]; cinf = l.ex[X_TCINF]; }
      __syncthreads();
      MYR_SWT(0)
      if constexpr (MLP) mlp_lin_passes(o, l);      // network systems: f, A, B at every stage point of every step -> records
      // Step linearisations, costates, step Hessians in ONE pass, lanes over steps (64 steps at a time, from the end).
      // The costate recursion pi_i = Fx_i^T pi_{i+1} + gx_i (+ own terms of a node state) is affine: a suffix scan of map
      // compositions over the wave (6 rounds) gives every lane the costate behind its step, with which it evaluates its
      // step Hessian right away; multipliers of the node defects and the stationarity residuals follow lane-wise.
      {
        double piS[NS], ruS[NU];          // costate / control-row residual of the point behind the current block (uniform)
#pragma unroll
        for (int c = 0; c < NS; ++c) piS[c] = (l.ex[X_TP + c] != 0.0) ? nuT[c] : l.zlu[xi(I, c)];
#pragma unroll
        for (int a = 0; a < NU; ++a) ruS[a] = l.zlu[ui(o, M * S, a)];
        double stat = 0, lam_inf = 0, sum_mult = 0;
        const double zero[NS] = {0};
        for (int base = ((S - 1) / 64) * 64; base >= 0; base -= 64) {
          const int i = base + lane;
          const bool on = i < S;
          const int k = on ? i / cpi : 0;
          const bool node_next = on && (i + 1 == (k + 1) * cpi), node_here = on && (i == k * cpi);
          double x[NS], uc[(M + 1) * NU], Fy[NS * NY], gy[NY], Hs[NY * NY], A[NS * NS], bb[NS];
#pragma unroll
          for (int q = 0; q < NS * NS; ++q) A[q] = ((q / NS) == (q % NS)) ? 1.0 : 0.0;     // identity beyond the last step
#pragma unroll
          for (int q = 0; q < NS; ++q) bb[q] = 0.0;
#pragma unroll
          for (int q = 0; q < NS * NY; ++q) Fy[q] = 0.0;
#pragma unroll
          for (int q = 0; q < NY; ++q) gy[q] = 0.0;
          if (on) {
#pragma unroll
            for (int c = 0; c < NS; ++c) x[c] = l.xs[(long)i * NS + c];
#pragma unroll
            for (int a = 0; a < (M + 1) * NU; ++a) uc[a] = l.z[ui(o, M * i, a)];
            double caff[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) caff[t] = node_next ? l.lam[(long)k * NS + t] : 0.0;
            if constexpr (MLP) {
              const RecEval ev{mlp_scr(l, S).pt, l.avec, (long)S, i, 0};
              SC::slin(o.method, h, x, uc, p, zero, Fy, gy, Hs, h * i, i == S - 1, ev);
            } else
              SC::slin(o.method, h, x, uc, p, zero, Fy, gy, Hs, h * i, i == S - 1);
            sw_lds* r = l.rec + (long)i * REC;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
#pragma unroll
              for (int c = 0; c < NY; ++c) r[R_GE + t * NY1 + c] = Fy[t * NY + c];
              r[R_GE + t * NY1 + NY] = caff[t];
            }
#pragma unroll
            for (int c = 0; c < NY; ++c) r[R_GY + c] = gy[c];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
#pragma unroll
              for (int t = 0; t < NS; ++t) A[c * NS + t] = Fy[t * NY + c];
              bb[c] = gy[c] + ((node_here && i > 0) ? l.zlu[xi(k, c)] : 0.0);
            }
          }
          affine_scan<NS, true>(A, bb);
          double pi_i[NS], pin[NS];
#pragma unroll
          for (int c = 0; c < NS; ++c) {
            double v = bb[c];
#pragma unroll
            for (int t = 0; t < NS; ++t) v += A[c * NS + t] * piS[t];
            pi_i[c] = v;
          }
#pragma unroll
          for (int c = 0; c < NS; ++c) { const double t = wv_down(pi_i[c], 1); pin[c] = (lane == 63) ? piS[c] : t; }
          // residual of the step's first control row (own part), then the stationarity of its last row
          double ru[NU], run[NU];
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            double v = ruS[a];
            if (on) {
              v = gy[NS + a] + l.zlu[ui(o, M * i, a)];
#pragma unroll
              for (int t = 0; t < NS; ++t) v += Fy[t * NY + NS + a] * pin[t];
            }
            ru[a] = v;
          }
#pragma unroll
          for (int a = 0; a < NU; ++a) { const double t = wv_down(ru[a], 1); run[a] = (lane == 63) ? ruS[a] : t; }
          if (on) {
#pragma unroll
            for (int a = 0; a < NU; ++a) {
              double rr = run[a] + gy[QN + a];
#pragma unroll
              for (int t = 0; t < NS; ++t) rr += Fy[t * NY + QN + a] * pin[t];
              stat = dmax(stat, fabs(rr));
            }
            if constexpr (M > 1) {
#pragma unroll
              for (int q = 0; q < (M - 1) * NU; ++q) {
                double rr = gy[NW + q] + l.zlu[ui(o, M * i + 1, q)];
#pragma unroll
                for (int t = 0; t < NS; ++t) rr += Fy[t * NY + NW + q] * pin[t];
                stat = dmax(stat, fabs(rr));
              }
            }
            if (node_next) {
#pragma unroll
              for (int c = 0; c < NS; ++c) {
                l.lam[(long)k * NS + c] = pin[c];            // lam_k = costate of the node
                lam_inf = dmax(lam_inf, fabs(pin[c]));
                sum_mult += fabs(pin[c]);
              }
            }
            // step Hessian of the Lagrangian with the costate of the step's end state
            if constexpr (MLP) {
              // (network systems: this run of the step algebra only records the stage weights; the Hessians follow behind the loop, after the MODE 4 passes)
              const RecEval ev{mlp_scr(l, S).pt, l.avec, (long)S, i, 1};
              SC::slin(o.method, h, x, uc, p, pin, Fy, gy, Hs, h * i, i == S - 1, ev);
#pragma unroll
              for (int c = 0; c < NS; ++c) l.pinS[(long)i * NS + c] = pin[c];
            } else {
            SC::slin(o.method, h, x, uc, p, pin, Fy, gy, Hs, h * i, i == S - 1);
            sw_lds* r = l.rec + (long)i * REC;
#pragma unroll
            for (int a = 0; a < NY; ++a)
#pragma unroll
              for (int b2 = 0; b2 <= a; ++b2) r[R_HS + hsp(a, b2)] = Hs[a * NY + b2];
            }
          }
#pragma unroll
          for (int c = 0; c < NS; ++c) piS[c] = __shfl(pi_i[c], 0, 64);
#pragma unroll
          for (int a = 0; a < NU; ++a) ruS[a] = __shfl(ru[a], 0, 64);
        }
        if constexpr (MLP) {
          mlp_hess_passes(o, l);
          for (int i0 = 0; i0 < S; i0 += 64) {
            const int i = i0 + lane < S ? i0 + lane : S - 1;
            double x[NS], uc[(M + 1) * NU], pin[NS], Fy[NS * NY], gy[NY], Hs[NY * NY];
#pragma unroll
            for (int c = 0; c < NS; ++c) { x[c] = l.xs[(long)i * NS + c]; pin[c] = l.pinS[(long)i * NS + c]; }
#pragma unroll
            for (int a = 0; a < (M + 1) * NU; ++a) uc[a] = l.z[ui(o, M * i, a)];
            const RecEval ev{mlp_scr(l, S).pt, l.avec, (long)S, i, 2};
            SC::slin(o.method, h, x, uc, p, pin, Fy, gy, Hs, h * i, i == S - 1, ev);
            if (i0 + lane < S) {
              sw_lds* r = l.rec + (long)i * REC;
#pragma unroll
              for (int a = 0; a < NY; ++a)
#pragma unroll
                for (int b2 = 0; b2 <= a; ++b2) r[R_HS + hsp(a, b2)] = Hs[a * NY + b2];
            }
          }
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) stat = dmax(stat, fabs(ruS[a]));          // first point
        stat = wv_max(stat); lam_inf = wv_max(lam_inf); sum_mult = wv_sum(sum_mult);
        if (lane == 0) {
          l.ex[X_F] = f; l.ex[X_C1] = c1; l.ex[X_CINF] = cinf; l.ex[X_STAT] = stat; l.ex[X_CMAX] = cmax; l.ex[X_CMIN] = cmin;
          l.ex[X_LAMINF] = lam_inf; l.ex[X_SUMMULT] = sum_mult; l.ex[X_NMULT] = (double)(I * NS);
          l.ex[X_VALID] = 1.0;
        }
      }
      __syncthreads();
      MYR_SWT(3)
    }
    // ---- Riccati recursion: matrix cores where the stage fits a tile, else one lane out of LDS ----
    if constexpr (MFMA_RICCATI) {
      bool aborted;
      int nreg = riccati_mfma(l, o, delta, so.abort_on_reg, aborted);
      if (lane == 0) {
        if (!aborted) {     // first point: x_0 pinned, eliminate du_0
          double P[NW * NW], pc[NW * NC], Tnu[NS * NC], Huu[NU * NU], g0u[NU], g1u[NU], ku[NU * NC];
#pragma unroll
          for (int q = 0; q < NW * NW; ++q) P[q] = l.ex[X_P + q];
#pragma unroll
          for (int q = 0; q < NW * NC; ++q) pc[q] = l.ex[X_PC + q];
#pragma unroll
          for (int q = 0; q < NS * NC; ++q) Tnu[q] = l.ex[X_TNU + q];
          const long v = ui(o, 0, 0);
          Huu[0] = l.sig[v] + delta; g0u[0] = 0.0; g1u[0] = l.g1[v];
          nreg += os_first_point<Sys>(P, pc, Huu, g0u, g1u, o.reg_floor, Tnu, ku);
#pragma unroll
          for (int q = 0; q < NU * NC; ++q) l.ku[q] = ku[q];
#pragma unroll
          for (int q = 0; q < NS * NC; ++q) l.ex[X_TNU + q] = Tnu[q];
        }
        l.ex[X_NREG] = (double)nreg;
      }
    } else
    if (lane == 0) {
      double P[NW * NW], pc[NW * NC], Tnu[NS * NC];
      int nreg = 0;
#pragma unroll
      for (int i = 0; i < NW * NW; ++i) P[i] = 0.0;
#pragma unroll
      for (int i = 0; i < NW * NC; ++i) pc[i] = 0.0;
#pragma unroll
      for (int i = 0; i < NS * NC; ++i) Tnu[i] = 0.0;
#pragma unroll
      for (int c = 0; c < NS; ++c) {
        const long v = xi(I, c);
        if (l.ex[X_TP + c] != 0.0) { pc[c * NC + 2 + c] = 1.0; P[c * NW + c] = o.rho_term; }
        else { P[c * NW + c] = l.sig[v] + delta; pc[c * NC + 1] = l.g1[v]; }
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        const long v = ui(o, M * S, a);
        P[(NS + a) * NW + NS + a] = l.sig[v] + delta; pc[(NS + a) * NC + 1] = l.g1[v];
      }
      bool aborted = false;
      for (int i = S - 1; i >= 0; --i) {
        const int k = i / cpi;
        const bool node_here = (i % cpi) == 0;
        const sw_lds* r = l.rec + (long)i * REC;
        double Ge[NS * NY1], gy[NY], Hs[NY * NY], qdiag[NQ], qg1[NQ], Kk[NQ * NW], kc[NQ * NC];
#pragma unroll
        for (int q = 0; q < NS * NY1; ++q) Ge[q] = r[R_GE + q];
#pragma unroll
        for (int q = 0; q < NY; ++q) gy[q] = r[R_GY + q];
#pragma unroll
        for (int a = 0; a < NY; ++a)
#pragma unroll
          for (int b = 0; b < NY; ++b) Hs[a * NY + b] = r[R_HS + hsp(a, b)];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { qdiag[q] = 0.0; qg1[q] = 0.0; }
        if constexpr (M > 1) {
#pragma unroll
          for (int q = 0; q < (M - 1) * NU; ++q) {
            const long v = ui(o, M * i + 1, q);
            qdiag[q] = l.sig[v] + delta; qg1[q] = l.g1[v];
          }
        }
        nreg += os_riccati_stage<Sys, M>(P, pc, Tnu, Ge, Hs, gy, o.reg_floor, Kk, kc, qdiag, qg1);
        if (nreg > 0 && so.abort_on_reg) { aborted = true; break; }
        sw_lds* g = l.kg + (long)i * KG;
#pragma unroll
        for (int q = 0; q < NQ * NW; ++q) g[q] = Kk[q];
#pragma unroll
        for (int q = 0; q < NQ * NC; ++q) g[NQ * NW + q] = kc[q];
        if (i > 0) {
#pragma unroll
          for (int a = 0; a < NU; ++a) {
            const long v = ui(o, M * i, a);
            P[(NS + a) * NW + NS + a] += l.sig[v] + delta; pc[(NS + a) * NC + 1] += l.g1[v];
          }
          if (node_here) {
#pragma unroll
            for (int c = 0; c < NS; ++c) {
              const long v = xi(k, c);
              P[c * NW + c] += l.sig[v] + delta; pc[c * NC + 1] += l.g1[v];
            }
          }
        }
      }
      if (!aborted) {       // first point: x_0 pinned, eliminate du_0
        double Huu[NU * NU], g0u[NU], g1u[NU], ku[NU * NC];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          const long v = ui(o, 0, a);
#pragma unroll
          for (int b2 = 0; b2 < NU; ++b2) Huu[a * NU + b2] = (a == b2) ? l.sig[v] + delta : 0.0;
          g0u[a] = 0.0; g1u[a] = l.g1[v];
        }
        nreg += os_first_point<Sys>(P, pc, Huu, g0u, g1u, o.reg_floor, Tnu, ku);
#pragma unroll
        for (int q = 0; q < NU * NC; ++q) l.ku[q] = ku[q];
      }
      l.ex[X_NREG] = (double)nreg;
#pragma unroll
      for (int q = 0; q < NS * NC; ++q) l.ex[X_TNU + q] = Tnu[q];
    }
    __syncthreads();
    so.f = l.ex[X_F]; so.c1 = l.ex[X_C1]; so.cinf = l.ex[X_CINF]; so.stat = l.ex[X_STAT]; so.compl_max = l.ex[X_CMAX];
    so.compl_min = l.ex[X_CMIN]; so.lam_inf = l.ex[X_LAMINF]; so.sum_mult = l.ex[X_SUMMULT]; so.n_mult = (int)l.ex[X_NMULT];
    so.nreg = (int)l.ex[X_NREG];
#pragma unroll
    for (int q = 0; q < NS * NC; ++q) so.Tnu[q] = l.ex[X_TNU + q];
#pragma unroll
    for (int c = 0; c < NS; ++c) so.term_pinned[c] = l.ex[X_TP + c] != 0.0;
    __syncthreads();
    MYR_SWT(4)
  }

  // ---- forward sweep (ShootCore::forward).  s_{i+1} = Phi_i s_i + phi_i, s = (dx, du) of a point: every lane forms the
  // closed-loop map of its step (gains applied to the step map, for the multipliers theta = (1, mu, nu)), a prefix scan
  // of map compositions over the wave gives it the state in front of its step, from which it takes its step
  // y_i = (s_i, -K_i s_i - kc_i theta) and its share of the directional derivative; step limits lanes-over-variables.
  __device__ static void forward(const HsWork& w, const HsSolveOpts& o, const double* p, double mu, const double* nu,
                                 const bool* term_pinned, FwdOut& fo) {
    (void)w; (void)p;
    const Lds l = lds(o);
    const int lane = threadIdx.x, cpi = o.cpi, S = steps(o), n = nvars(o);
    const double tau = detail::dmax(o.tau_min, 1.0 - mu);
    MYR_SWT0F
    double th[NC];
    th[0] = 1.0; th[1] = mu;
#pragma unroll
    for (int i = 0; i < NS; ++i) th[2 + i] = nu[i];
    // 64 steps at a time: closed-loop maps in registers, prefix scan of their compositions, then every lane's own step
    double s0[NW], gphi = 0.0;          // state (dx, du) of the first point of the block (uniform)
#pragma unroll
    for (int c = 0; c < NS; ++c) s0[c] = 0.0;
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      double v = 0.0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v -= l.ku[a * NC + cc] * th[cc];
      s0[NS + a] = v;
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < NS; ++c) l.dz[xi(0, c)] = 0.0;
#pragma unroll
      for (int a = 0; a < NU; ++a) l.dz[ui(o, 0, a)] = s0[NS + a];
    }
    for (int base = 0; base < S; base += 64) {
      const int i = base + lane;
      const bool on = i < S;
      double Phi[NW * NW], phi[NW], K[NQ * NW], kt[NQ], gy[NY];
#pragma unroll
      for (int q = 0; q < NW * NW; ++q) Phi[q] = ((q / NW) == (q % NW)) ? 1.0 : 0.0;       // identity beyond the last step
#pragma unroll
      for (int q = 0; q < NW; ++q) phi[q] = 0.0;
      if (on) {
        const sw_lds* r = l.rec + (long)i * REC;
        const sw_lds* g = l.kg + (long)i * KG;
        double Ge[NS * NY1];
#pragma unroll
        for (int q = 0; q < NS * NY1; ++q) Ge[q] = r[R_GE + q];
#pragma unroll
        for (int q = 0; q < NY; ++q) gy[q] = r[R_GY + q];
#pragma unroll
        for (int q = 0; q < NQ * NW; ++q) K[q] = g[q];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          double v = 0.0;
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) v += g[NQ * NW + q * NC + cc] * th[cc];
          kt[q] = v;
        }
#pragma unroll
        for (int t = 0; t < NS; ++t) {
          const bool zero_ = (i == S - 1) && term_pinned[t];
#pragma unroll
          for (int c = 0; c < NW; ++c) {
            double v = Ge[t * NY1 + c];
#pragma unroll
            for (int q = 0; q < NQ; ++q) v -= Ge[t * NY1 + NW + q] * K[q * NW + c];
            Phi[t * NW + c] = zero_ ? 0.0 : v;
          }
          double v = Ge[t * NY1 + NY];
#pragma unroll
          for (int q = 0; q < NQ; ++q) v -= Ge[t * NY1 + NW + q] * kt[q];
          phi[t] = zero_ ? 0.0 : v;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          const int q = (M - 1) * NU + a;                   // the step's last control row is the next point's control
#pragma unroll
          for (int c = 0; c < NW; ++c) Phi[(NS + a) * NW + c] = -K[q * NW + c];
          phi[NS + a] = -kt[q];
        }
      }
      affine_scan<NW, false>(Phi, phi);
      double sn[NW], sc[NW];                // state behind / in front of this lane's step
#pragma unroll
      for (int t = 0; t < NW; ++t) {
        double v = phi[t];
#pragma unroll
        for (int c = 0; c < NW; ++c) v += Phi[t * NW + c] * s0[c];
        sn[t] = v;
      }
#pragma unroll
      for (int c = 0; c < NW; ++c) { const double t = wv_up(sn[c], 1); sc[c] = (lane == 0) ? s0[c] : t; }
      if (on) {
        double y[NY];
#pragma unroll
        for (int c = 0; c < NW; ++c) y[c] = sc[c];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          double v = -kt[q];
#pragma unroll
          for (int c = 0; c < NW; ++c) v -= K[q * NW + c] * y[c];
          y[NW + q] = v;
        }
#pragma unroll
        for (int c = 0; c < NY; ++c) gphi += gy[c] * y[c];          // d(objective) along the lifted step
        if (((i + 1) % cpi) == 0) {
#pragma unroll
          for (int c = 0; c < NS; ++c) l.dz[xi((i + 1) / cpi, c)] = sn[c];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) l.dz[ui(o, M * i + 1, q)] = y[NW + q];     // rows M i + 1 .. M i + M: (mid,) next control
      }
#pragma unroll
      for (int c = 0; c < NW; ++c) s0[c] = __shfl(sn[c], 63, 64);
    }
    __syncthreads();
    FwdOut fl; fl.alpha_p = 1.0; fl.alpha_d = 1.0; fl.gphi = 0.0;
    for (int v = NS + lane; v < n; v += 64)       // every variable but x_0 (ShootCore::forward sets dz = 0 there, no limits)
      H::step_limits(l.z[v], l.lb[v], l.ub[v], l.zL[v], l.zU[v], l.dz[v], mu, 0.0, tau, fl);
    fo.alpha_p = wv_min(fl.alpha_p); fo.alpha_d = wv_min(fl.alpha_d); fo.gphi = wv_sum(gphi + fl.gphi);
    __syncthreads();
    MYR_SWTF(5)
  }

  // ---- merit function at z + alpha dz: trial point and barrier (lanes over variables), rollouts (lanes over intervals).
  // At alpha = 0 (the reference value of every line search) objective and defects are those of the sweep just done.
  __device__ static bool trial(const HsWork& w, const HsSolveOpts& o, const double* p, double alpha, double mu,
                               double& f, double& bar, double& c1) {
    (void)w;
    const Lds l = lds(o);
    const int lane = threadIdx.x, I = o.N, n = nvars(o);
    double fl = 0, bl = 0, cl = 0; int bad = 0;
    MYR_SWT0
    const bool at_z = (alpha == 0.0) && (l.ex[X_VALID] != 0.0);
    for (int i = lane; i < n; i += 64) {
      const double v = l.z[i] + alpha * l.dz[i];
      const double lo = l.lb[i], ub = l.ub[i];
      const bool fr = lo < ub;
      const bool hl = fr && (lo > -INFINITY), hu = fr && (ub < INFINITY);
      const double sl = hl ? v - lo : 1.0, su = hu ? ub - v : 1.0;
      bad += (sl > 0.0 ? 0 : 1) + (su > 0.0 ? 0 : 1);
      bl -= log((sl > 0.0 ? sl : 1.0) * (su > 0.0 ? su : 1.0));
      if (!at_z) l.zt[i] = v;
    }
    if (at_z) {
      f = l.ex[X_F]; c1 = l.ex[X_C1];
    } else {
      __syncthreads();
      double cm = 0;
      if constexpr (MLP) roll_all_mlp(o, p, l, l.zt, l.xt, l.ct, fl, cl, cm);
      else
      for (int k = lane; k < I; k += 64) {
        double x[NS], xe[NS];
#pragma unroll
        for (int c = 0; c < NS; ++c) { x[c] = l.zt[xi(k, c)]; xe[c] = l.zt[xi(k + 1, c)]; }
        roll_interval(o, p, l.zt, k, x, l.xt, fl);
#pragma unroll
        for (int c = 0; c < NS; ++c) {
          const double ck = x[c] - xe[c];
          l.ct[(long)k * NS + c] = ck;
          cl += fabs(ck);
          cm = detail::dmax(cm, fabs(ck));
        }
      }
      f = wv_sum(fl); c1 = wv_sum(cl); cm = wv_max(cm);
      if (lane == 0) { l.ex[X_TA] = alpha; l.ex[X_TF] = f; l.ex[X_TC1] = c1; l.ex[X_TCINF] = cm; }
    }
    bar = mu * wv_sum(bl); bad = wv_isum(bad);
    __syncthreads();
    MYR_SWT(6)
    if (bad != 0) return false;
    if (!detail::finite_(f)) return false;
    if (!detail::finite_(c1)) return false;
    return detail::finite_(bar);
  }
};

// Persistent: grid = resident workgroups (one wavefront each); each pulls trajectories from `ticket`.  z / lb / ub / lam are
// the caller's instance-major rows ([B][n], [B][I*NS]); the iterate is copied into LDS, solved there and copied back.
template <class Sys, int M = 1>
__global__ __launch_bounds__(64, NodeTraits<Sys>::mlp ? 1 : MYR_SHOOT_MIN_WAVES)      // (network systems: one workgroup per CU by LDS anyway -- the whole register file for the matrix-core passes)
void shoot_solve_wave_kernel(int B, int* ticket, HsSolveOpts o, VarScale vs, double* __restrict__ z, const double* __restrict__ lb,
                             const double* __restrict__ ub, double* __restrict__ lam, const double* __restrict__ params, int params_stride,
                             double* cost, int32_t* status, int32_t* iters, double* kkt, unsigned long long poison, double* scratch, long scratch_stride) {
  using W = ShootWave<Sys, M>;
  const typename W::Lds l = W::lds(o);
  const int n = W::nvars(o), m = o.N * W::NS;
  (void)scratch; (void)scratch_stride;
  for (;;) {
    int t = 0;
    if (threadIdx.x == 0) t = atomicAdd(ticket, 1);
    const long b = __builtin_amdgcn_readfirstlane(t);
    if (b >= B) break;
    if (poison) {      // MYRIAD_POISON (tests/test_gpu_poison.py): the whole LDS of the workgroup, which is all a trajectory inherits here
      extern __shared__ __attribute__((aligned(16))) char smem_poison[];
      double* l0 = reinterpret_cast<double*>(smem_poison);
      const int nl = (int)(W::lds_bytes(o.N, o.cpi) / 8);
      for (int i = threadIdx.x; i < nl; i += 64) l0[i] = poison_value(poison, (unsigned long long)i, (unsigned long long)b * 1315423911ULL + blockIdx.x);
      __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 64) { const double v = z[b * n + i]; l.z[i] = v; l.z0[i] = v; l.lb[i] = lb[b * n + i]; l.ub[i] = ub[b * n + i]; }
    if (threadIdx.x == 0) { l.ex[W::X_Z] = 0.0; l.ex[W::X_DN] = (double)o.N; l.ex[W::X_DCPI] = (double)o.cpi; l.ex[W::X_TA] = -1.0; }
    SysParams<Sys> pp;
    pp.load(params, b, params_stride);
    pp.set_scale(vs.s);
    if constexpr (W::MLP) {      // network systems: the weights (matrix-core operand layout) and this workgroup's global scratch
      NodeMfma64::load_weights(pp.get(), (double*)l.wl, (int)threadIdx.x, 64);
      if (threadIdx.x == 0) l.ex[W::X_SCR] = __builtin_bit_cast(double, (unsigned long long)(scratch + (long)blockIdx.x * scratch_stride));
    }
    __syncthreads();
    HsWork w{{(double*)l.z, 1}, {(double*)l.lb, 1}, {(double*)l.ub, 1}, {(double*)l.zL, 1}, {(double*)l.zU, 1}, {(double*)l.lam, 1}, {(double*)l.dz, 1},
             {(double*)l.rec, 1}};
    HsSolveResult r;
#ifdef MYR_SW_TIMING
    if (threadIdx.x < 8) l.ex[W::X_T + threadIdx.x] = 0.0;
    const long long tall_ = wall_clock64();
#endif
    // first start: an eighth of the iteration limit (at least 100) -- a solve that has not converged by then is almost
    // always one that crawls to the limit (config 3: p99 56 iterations, slowest converging solve 180, stuck ones 1000),
    // a second start typically needs 30 .. 50 iterations, and a stuck trajectory is what the launch otherwise waits for
    // (seeds 2020 / 2021 of config 3: 170 / 105 ms against 20 ms for the other 8191 trajectories)
    HsSolveOpts o1 = o;
    o1.max_iter = o.max_iter / 8 > 100 ? o.max_iter / 8 : (o.max_iter < 100 ? o.max_iter : 100);
    IpLoop<W>::run(w, o.restarts > 0 ? o1 : o, pp.get(), r);
    __syncthreads();
    // A solve that ends without a KKT point (line search stalled on a non-descent direction, iteration limit, non-finite
    // values) is restarted from the caller's point with another initial barrier parameter (x3, then /3): the iterates of
    // single shooting over a long horizon are sensitive enough that the slowest 0.01 % of a batch depend on rounding --
    // config 3's trajectory 3985 stalls after 58 iterations from mu = 0.1 with this kernel's summation order and needs
    // 31 .. 49 iterations from any of 0.01, 0.03, 0.3, 0.5, 1.  There is no restoration phase to fall back on;
    // a second start is the cheap substitute.  The restarts get the full iteration limit; iterations and sweeps of all
    // attempts are reported (so `iters` can exceed max_iter).
    for (int attempt = 0; attempt < o.restarts && r.status != 0; ++attempt) {
      for (int i = threadIdx.x; i < n; i += 64) l.z[i] = l.z0[i];
      __syncthreads();
      HsSolveOpts o2 = o;
      // a different initial barrier parameter per attempt: x 3, / 3, x 9, / 9 (a repeated factor would repeat the failed solve)
      const double fac = (attempt >> 1) == 0 ? 3.0 : 9.0;
      o2.mu_init = o.mu_init * ((attempt & 1) == 0 ? fac : 1.0 / fac);
      HsSolveResult r2;
      IpLoop<W>::run(w, o2, pp.get(), r2);
      __syncthreads();
      r2.iters += r.iters; r2.sweeps += r.sweeps;
      r = r2;
    }
#ifdef MYR_SW_TIMING
    if (threadIdx.x == 0 && b < 3)
      printf("traj %ld it %d sweeps %d, x10ns: total %lld | own+rollout %.0f lin1 %.0f costate %.0f hess %.0f riccati %.0f forward %.0f trial %.0f\n", b, r.iters,
             r.sweeps, wall_clock64() - tall_, l.ex[W::X_T], l.ex[W::X_T + 1], l.ex[W::X_T + 2], l.ex[W::X_T + 3], l.ex[W::X_T + 4], l.ex[W::X_T + 5], l.ex[W::X_T + 6]);
#endif
    for (int i = threadIdx.x; i < n; i += 64) z[b * n + i] = l.z[i];
    if (lam) for (int i = threadIdx.x; i < m; i += 64) lam[b * m + i] = l.lam[i];
    if (threadIdx.x == 0) {
      if (cost) cost[b] = r.cost;
      if (status) status[b] = r.status;
      if (iters) iters[b] = r.iters;
      if (kkt) { kkt[3 * b] = r.feas; kkt[3 * b + 1] = r.stat; kkt[3 * b + 2] = r.compl_; }
    }
    __syncthreads();      // LDS is handed to the next trajectory
  }
}

}  // namespace myriad
