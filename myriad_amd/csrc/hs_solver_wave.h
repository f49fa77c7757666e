// hs_solver_wave.h -- the SAME interior-point SQP as hs_solver.h, mapped ONE TRAJECTORY PER WAVEFRONT.
//
// hs_solver.h runs a whole trajectory in one lane: simple, host-testable, and right when the batch is large enough
// to fill the machine with lanes (B >~ 64k); but its time is max_iterations x (sequential work of one lane), and at
// the BASELINE batch (4096) most of the chip idles.  Here the 64 lanes of a wavefront share one trajectory:
//   * everything that is independent across collocation points or intervals is done lanes-over-points /
//     lanes-over-intervals (dynamics + derivatives with their sin/cos, bound terms, interval eliminations
//     (LU of E, G_e, G_m), adjoint maps, Lagrangian Hessians, midpoint Schur terms, step limits, merit trials,
//     updates);
//   * only two recursions stay sequential over the N stages and are executed cooperatively through LDS:
//     the Riccati sweep (one column of the stage blocks per lane, see riccati()) and the forward state recursion;
//     the adjoint recursion is reduced to an affine recurrence Pi_{k-1} = M_k Pi_k + v_k with M, v staged in LDS.
// Trajectory data stays instance-major in HBM (the caller's z/lb/ub rows are used in place, no transposes);
// per-trajectory scratch is one contiguous block, so every parallel phase reads/writes it with unit stride.
// The algorithm, its constants and its control flow are those of HsSolver<Sys>::solve (hs_solver.h) -- the two
// kernels are tested against each other and against the same golden trajectories.
#pragma once
#include <hip/hip_runtime.h>
#include "hs_solver.h"
#include "node_mfma.h"

namespace myriad {

// network dynamics (node_system.h) are evaluated by the matrix-core passes of node_mfma.h inside the wavefront solver (NodeTraits there)

// Barrier of the phases of ONE wavefront.  A workgroup of a single wavefront uses the hardware barrier; network systems pack
// several independent wavefronts into a workgroup (they share the weights in LDS) and may not meet at a workgroup barrier:
// there the phases of a wavefront are ordered by a fence (all its LDS / global accesses retired) + the wave barrier.
template <bool MULTI>
__device__ inline void wave_sync() {
  if constexpr (MULTI) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
  else __syncthreads();
}
__device__ inline double wv_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ inline double wv_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o, 64); v = v > t ? v : t; }
  return v;
}
__device__ inline double wv_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o, 64); v = v < t ? v : t; }
  return v;
}
// reciprocal from v_rcp_f64 + two Newton steps (the operands here are pivots already known to exceed reg_floor)
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}
// L D L^T of a small SPD block with the SAME pivot rule as detail::chol_reg (the pivots d_j are the squares of the
// Cholesky diagonal): a <- unit lower factor (strict lower part), dinv <- 1 / d.  No sqrt, one reciprocal per pivot.
template <int n>
__device__ inline int ldl_reg(double* a, double* dinv, double floor_) {
  int nreg = 0;
  double d[n];
#pragma unroll
  for (int j = 0; j < n; ++j) {
    double dj = a[j * n + j];
#pragma unroll
    for (int k = 0; k < j; ++k) dj -= a[j * n + k] * a[j * n + k] * d[k];
    if (!(dj > floor_)) { dj = detail::dmax(fabs(dj), floor_); ++nreg; }
    d[j] = dj;
    dinv[j] = fast_rcp(dj);
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      double t = a[i * n + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= a[i * n + k] * a[j * n + k] * d[k];
      a[i * n + j] = t * dinv[j];
    }
  }
  return nreg;
}
template <int n>
__device__ inline void ldl_solve(const double* a, const double* dinv, double* b) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) b[i] -= a[i * n + k] * b[k];
  }
#pragma unroll
  for (int i = 0; i < n; ++i) b[i] *= dinv[i];
#pragma unroll
  for (int i = n - 1; i >= 0; --i) {
#pragma unroll
    for (int k = i + 1; k < n; ++k) b[i] -= a[k * n + i] * b[k];
  }
}

// every lane <- lane LANE of its own row of 16 lanes (DPP row_newbcast: stays in the vector pipe, ~10 cycles; v_readlane
// goes through the scalar register file and costs ~55 cycles before a vector instruction can use the value)
template <int LANE>
__device__ inline double wv_row_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v), rlo, rhi;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
               : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi), "n"(LANE));
  return __hiloint2double(rhi, rlo);
}
template <int NQ_>
struct RowBcast {       // q[i] = value of lane i (i < NQ_ <= 16) of the caller's row, for all i
  template <int I = 0>
  __device__ static inline void all(double v, double* q) {
    if constexpr (I < NQ_) { q[I] = wv_row_bcast<I>(v); all<I + 1>(v, q); }
  }
};

__device__ inline int wv_isum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Affine maps x -> A x + b (n x n) in registers, one per lane: composition and the two wave scans built on it.
// `wv_down(v, d)` = value of lane + d (own value beyond the wave), `wv_up` = lane - d.
__device__ inline double wv_down(double v, int d) { return __shfl_down(v, d, 64); }
__device__ inline double wv_up(double v, int d) { return __shfl_up(v, d, 64); }
template <int n>
__device__ inline void affine_after(double* A, double* b, const double* A2, const double* b2) {   // (A,b) <- (A,b) o (A2,b2)
  double R[n * n], r[n];
#pragma unroll
  for (int i = 0; i < n; ++i) {
    double v = b[i];
#pragma unroll
    for (int k = 0; k < n; ++k) v += A[i * n + k] * b2[k];
    r[i] = v;
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < n; ++k) w += A[i * n + k] * A2[k * n + j];
      R[i * n + j] = w;
    }
  }
#pragma unroll
  for (int i = 0; i < n * n; ++i) A[i] = R[i];
#pragma unroll
  for (int i = 0; i < n; ++i) b[i] = r[i];
}
// suffix scan: lane l <- T_l o T_{l+1} o .. o T_63        prefix scan: lane l <- T_l o T_{l-1} o .. o T_0
template <int n, bool SUFFIX>
__device__ inline void affine_scan(double* A, double* b) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    double A2[n * n], b2[n];
#pragma unroll
    for (int i = 0; i < n * n; ++i) A2[i] = SUFFIX ? wv_down(A[i], d) : wv_up(A[i], d);
#pragma unroll
    for (int i = 0; i < n; ++i) b2[i] = SUFFIX ? wv_down(b[i], d) : wv_up(b[i], d);
    if (SUFFIX ? (lane + d < 64) : (lane >= d)) affine_after<n>(A, b, A2, b2);
  }
}
// The prefix scan with DPP moves instead of ds_bpermute (one VALU move per dword and round): four shifts inside the rows
// of 16 lanes, then row_bcast:15 (rows 1, 3 take lane 15 / 47) and row_bcast:31 (lanes 32..63 take lane 31) -- the
// wave-scan idiom of GFX9.  Inline asm, executed by all lanes: a DPP builtin sunk into the divergent branch that consumes
// it would read 0 from the lanes EXEC has switched off (see dpp_row_shr4 below).
template <int STEP>
__device__ inline double dpp_scan_src(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v), rlo, rhi;
  if constexpr (STEP == 0)
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
  else if constexpr (STEP == 1)
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %3 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
  else if constexpr (STEP == 2)
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
  else if constexpr (STEP == 3)
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_mov_b32_dpp %1, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
  else if constexpr (STEP == 4)
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_bcast:15 row_mask:0xa bank_mask:0xf" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
  else
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_bcast:31 row_mask:0xc bank_mask:0xf" : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
  return __hiloint2double(rhi, rlo);
}
template <int n, int STEP>
__device__ inline void affine_prefix_round(double* A, double* b) {
  const int lane = threadIdx.x & 63, l16 = lane & 15;
  double A2[n * n], b2[n];
#pragma unroll
  for (int i = 0; i < n * n; ++i) A2[i] = dpp_scan_src<STEP>(A[i]);
#pragma unroll
  for (int i = 0; i < n; ++i) b2[i] = dpp_scan_src<STEP>(b[i]);
  const bool take = STEP < 4 ? (l16 >= (1 << STEP)) : (STEP == 4 ? ((lane >> 4) & 1) != 0 : lane >= 32);
  if (take) affine_after<n>(A, b, A2, b2);
}
template <int n>
__device__ inline void affine_prefix_scan_dpp(double* A, double* b) {       // lane l <- T_l o T_{l-1} o .. o T_0
  affine_prefix_round<n, 0>(A, b); affine_prefix_round<n, 1>(A, b); affine_prefix_round<n, 2>(A, b);
  affine_prefix_round<n, 3>(A, b); affine_prefix_round<n, 4>(A, b); affine_prefix_round<n, 5>(A, b);
}

// SCHEME 0: Hermite-Simpson (K = 2N+1 points, stage unknowns y = (dx_s, du_s, du_m, du_e), two eliminated controls);
// SCHEME 1: trapezoidal collocation (/root/reference/myriad/trajectory_optimizers/collocation/trapezoidal.py:80-163; K = N+1
// points, y = (dx_s, du_s, du_e), one eliminated control, no midpoint terms) -- the same phases, the same sweep, the
// algorithm of TrapCore (os_solver.h), against whose host build it is tested.
template <class Sys, int SCHEME = 0>
struct HsWave {
  using S = HsSolver<Sys>;
  using D = HsSol<Sys>;
  static constexpr bool TRAP = SCHEME == 1;
  static constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NY = TRAP ? NS + 2 * NU : D::NY, NQ = TRAP ? NU : D::NQ, NC = D::NC, NY1 = NY + 1;
  static constexpr int QE = NQ - NU;               // position of du_e among the eliminated controls q
  static constexpr int MLAM = TRAP ? 1 : 2;        // multiplier blocks (NS each) per interval
  static constexpr bool MLP = NodeTraits<Sys>::mlp;
  static_assert(!(MLP && TRAP), "network dynamics are built for the Hermite-Simpson transcription");
  // wavefronts per workgroup: network systems share their weights (40 KB of LDS) between WPB_MAX independent solves
  // (4 x 24 KB + 40 KB = 138 KB of LDS: one wavefront per SIMD, which is what 512 registers allow anyway)
  static constexpr int WPB_MAX = MLP ? 4 : 1;
  __device__ static inline void wsync() { wave_sync<MLP>(); }
  static constexpr int ND2 = MLP ? NodeMfma64::NPAIR : Sys::NNZ2;   // stored second-derivative data per point
  __host__ __device__ static constexpr int npoints(int N) { return TRAP ? N + 1 : 2 * N + 1; }
  // quadrature weight and time of point j (hermite_simpson.py:212-214 / trapezoidal.py:80-94)
  __host__ __device__ static inline double wq(int K, int j, double h) {
    if (TRAP) return (j == 0 || j == K - 1) ? 0.5 * h : h;
    return S::wsimp(K, j, h);
  }
  __host__ __device__ static inline double tq(int j, double h) { return TRAP ? h * j : 0.5 * h * j; }
  // per-point record, SoA over points: field f of point j at pt[f*K + j]
  // f and A, which only the elimination phase reads, come LAST: the Hessian records (written after that phase, dead before
  // the next linearisation) are overlaid on them -- and the gains K | kc on the adjoint maps Ld..li0 of the stage record,
  // read for the last time before the sweep writes gains.  273 -> 223 KB of scratch per resident wavefront: the
  // working set of a launch (1024 slots) drops below the 256 MB Infinity Cache.
  static constexpr int PF_B = 0, PF_GW = PF_B + NS * NU, PF_D2 = PF_GW + NW,
                       PF_SIG = PF_D2 + ND2, PF_G1 = PF_SIG + NW, PF_ZLU = PF_G1 + NW, PF_F = PF_ZLU + NW, PF_A = PF_F + NS,
                       PF_N = PF_A + NS * NS;
  // per-point Hessian record, AoS: H (NW x NW), g0 (NW), g1 (NW)
  static constexpr int HR_H = 0, HR_G0 = NW * NW, HR_G1 = HR_G0 + NW, HR_N = HR_G1 + NW;
  // per-stage record, AoS (the trapezoidal scheme has no midpoint map Gm and one multiplier block per interval)
  static constexpr int SG_GE = 0, SG_GM = SG_GE + NS * NY1, SG_LD = SG_GM + (TRAP ? 0 : NS * NY1), SG_LD0 = SG_LD + NS * NS,
                       SG_LI = SG_LD0 + NS, SG_LI0 = SG_LI + (TRAP ? 0 : NS * NS), SG_QM = SG_LI0 + (TRAP ? 0 : NS),
                       SG_QCM = SG_QM + (TRAP ? 0 : NY * NY),
#if defined(MYR_RICCATI_VALU) || defined(MYR_RICCATI_CHECK)
                       SG_N = SG_QCM + (TRAP ? 0 : NY * 2);
#else     // the matrix-core sweep forms Qm | qcm itself: the record ends before them (CARTPOLE: 167 -> 104 doubles per stage)
                       SG_N = (NU == 1 && NS <= 4) ? SG_QM : SG_QCM + (TRAP ? 0 : NY * 2);
#endif
  static constexpr int KST = NQ * NW + NQ * NC;   // K | kc per stage (global scratch)
#ifdef MYR_RICCATI_CHECK
  static constexpr bool OVERLAY_K = false;        // (the self-check addresses the gains as one flat array)
#else
  static constexpr bool OVERLAY_K = KST <= SG_QM - SG_LD;
#endif
  static constexpr int KSTR = OVERLAY_K ? SG_N : KST;   // stride of the gain records
  static constexpr int ZR = 2 * NY + 2;           // block of zeros (masked stage inputs of the Riccati lanes read it)
  static constexpr int PHI = NW * (NW + 1);       // closed-loop stage map Phi | phi per stage (LDS)

  // The sweeps' prefetch rings read MYR_RICCATI_PF stages below stage 0 (never used): PADF doubles in front of every slot keep
  // those reads inside the allocation whatever the system's sizes (VANDERPOL, N = 1: 105 doubles back from 84 in front).
  static constexpr int PADF = 8 * ((2 * HR_N > SG_N ? 2 * HR_N : SG_N) + 1);
  __host__ __device__ static long scratch_doubles(int N) {
    const long K = npoints(N), n = K * NW;
    const long fa = (long)(NS + NS * NS) * K, hrn = (long)HR_N * K;        // hr overlays the f | A fields at the end of pt
    return PADF + 3 * n + (long)PF_N * K + (hrn > fa ? hrn - fa : 0) + (long)SG_N * N + (OVERLAY_K ? 0 : (long)KST * N) + ZR + 2 /* write-only slot */ +
           (long)MLAM * N * NS /* lambda when the caller passes none */;
  }
  // LDS doubles: region R0 (adjoint M|v, later Phi|phi, later trial x|f), Pi, S, exchange
  __host__ __device__ static int r0_doubles(int N) {
#if defined(MYR_RECUR_SEQ) || defined(MYR_RECUR_SEQ_FWD)
    const int b = N * PHI;
#else
    const int b = NodeTraits<Sys>::mlp ? 0 : N * PHI;     // (network systems: the closed-loop maps are not staged, see PHI_IN_LDS)
#endif
    const int a = N * (NS * NS + NS), c = 2 * npoints(N) * NS;
    return a > b ? (a > c ? a : c) : (b > c ? b : c);
  }
  static constexpr int EXCH = NW * NW + NW * NC + NS * NY1 + NS * NC + NU * NC + 8;
  __host__ __device__ static int lds_solver_doubles(int N) { return r0_doubles(N) + N * NS + (N + 1) * NW + EXCH + 8; }
  __host__ __device__ static size_t lds_bytes(int N) { return (size_t)(lds_solver_doubles(N) + NodeTraits<Sys>::lds_doubles) * 8; }

  struct Ctx {
    int N, K, n, lane;
    double h, h6, h8;
    double *z, *zL, *zU, *dz, *lam, *pt, *hr, *st, *kg, *zr;
    const double *lb, *ub;
    SysParams<Sys> pp;
    bool term_pinned[NS];
    // LDS
    double *r0, *sPi, *sS, *sP, *sPc, *sGe, *sTnu, *sKu;
    double* wl;           // network weights in LDS (node_mfma.h), network systems only
    int coop = 0;         // > 1: that many wavefronts of the workgroup share THIS trajectory's network passes (see the kernel)
    double* cmd = nullptr;   // LDS mailbox of the cooperative mode
#ifdef MYR_PHASE_TIMING
    long long tph[16], t0;
#endif
  };
#ifdef MYR_PHASE_TIMING
#define MYR_PH(i) { const long long t1_ = clock64(); c.tph[i] += t1_ - c.t0; c.t0 = t1_; }
#else
#define MYR_PH(i)
#endif

  __device__ static inline long zi(const Ctx& c, int j, int comp) { return S::zi(c.K, j, comp); }

  // ---- network systems: matrix-core passes over all points (node_mfma.h) ---------------------------------------------
  // Cooperative mode (small batches: fewer trajectories than CUs): the solver runs in wavefront 0 of a workgroup, and for
  // every network pass it posts the arguments in LDS and meets the other wavefronts at a workgroup barrier; the tiles of 16
  // points are dealt round-robin, a second barrier ends the pass.  The helpers spend the rest of the solve at the barrier.
  struct CoopCmd { int mode; NodeMfma64::Args a; };
  template <int MODE>
  __device__ static inline void node_pass(Ctx& c, double alpha) {
    if constexpr (MLP) {
      NodeMfma64::Args a;
      a.z = (const nd_glb*)c.z; a.dz = (const nd_glb*)c.dz; a.lam = (const nd_glb*)c.lam; a.pt = (nd_glb*)c.pt;
      a.sF = (nd_lds*)(c.r0 + (long)c.K * NS);
      a.alpha = alpha; a.h6 = c.h6; a.h8 = c.h8; a.K = c.K; a.N = c.N;
      a.pf_f = PF_F; a.pf_a = PF_A; a.pf_b = PF_B; a.pf_d2 = PF_D2;
      if (c.coop > 1) {
        a.ts = c.coop;
        if (c.lane == 0) { CoopCmd* m = reinterpret_cast<CoopCmd*>(c.cmd); m->mode = MODE; m->a = a; }
        __syncthreads();
        NodeMfma64::pass<MODE>((const nd_lds*)c.wl, a, c.lane);          // (t0 = 0: wavefront 0's share)
        __syncthreads();
      } else
        NodeMfma64::pass<MODE>((const nd_lds*)c.wl, a, c.lane);
    } else { (void)c; (void)alpha; }
  }
  __device__ static void coop_helper(const double* wl, double* cmd, int wave, int lane) {
    if constexpr (MLP) {
      for (;;) {
        __syncthreads();
        const CoopCmd* m = reinterpret_cast<const CoopCmd*>(cmd);
        const int mode = m->mode;
        if (mode < 0) break;
        NodeMfma64::Args a = m->a;
        a.t0 = wave;
        if (mode == 0) NodeMfma64::pass<0>((const nd_lds*)wl, a, lane);
        else if (mode == 1) NodeMfma64::pass<1>((const nd_lds*)wl, a, lane);
        else NodeMfma64::pass<2>((const nd_lds*)wl, a, lane);
        __syncthreads();
      }
    } else { (void)wl; (void)cmd; (void)wave; (void)lane; }
  }
  static constexpr int COOP_CMD_DOUBLES = (sizeof(CoopCmd) + 7) / 8 + 1;

  // ---- phase 1: lanes over points -- linearisation, bound terms ----------------------------------------------
  struct P1 { double f, cmax, cmin, sm, lg; int nm; };   // lg = -sum log(slack): barrier term of the merit function / mu
  // The accepted step of the previous iteration is applied HERE, on the values this phase loads anyway (update() as a
  // phase of its own re-read six arrays and paid their latency sixteen rounds in a row): z += a_p dz, zL/zU += a_d d..,
  // same formulas, same order of operations as update().
  struct Step { bool on; double ap, ad, mu, ksig; };
  __device__ static void points_lin(Ctx& c, P1& o, const Step& st) {
    if constexpr (MLP) {       // the network pass reads z from memory: apply the step first
      if (st.on) { update(c, st.ap, st.ad, st.mu, st.ksig); wsync(); }
    }
    node_pass<1>(c, 0.0);
    double f = 0, cmax = 0, cmin = INFINITY, sm = 0, lg = 0; int nm = 0;
    const double iks = 1.0 / st.ksig;
    for (int j = c.lane; j < c.K; j += 64) {
      typename S::VarBlk V;
      double slk = 1.0; int sexp = 0;       // as in trial(): one log per point
      if (!MLP && st.on) {
        double dv[NW];                        // all loads of the point before its first store (see points_hess)
#pragma unroll
        for (int q = 0; q < NW; ++q) {
          const long i = zi(c, j, q);
          V.z[q] = c.z[i]; V.l[q] = c.lb[i]; V.u[q] = c.ub[i]; V.zl[q] = c.zL[i]; V.zu[q] = c.zU[i]; dv[q] = c.dz[i];
        }
#pragma unroll
        for (int q = 0; q < NW; ++q) {
          const long i = zi(c, j, q);
          const double l = V.l[q], u = V.u[q], zv = V.z[q], d = dv[q], zl = V.zl[q], zu = V.zu[q];
          const bool fr = l < u;
          const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
          const double zn = fr ? zv + st.ap * d : zv;
          const double sl = hl ? zv - l : 1.0, su = hu ? u - zv : 1.0;
          const double snl = hl ? zn - l : 1.0, snu = hu ? u - zn : 1.0;
          double vl = zl + st.ad * (-zl + (st.mu - zl * d) * detail::rcp_(sl));
          double vu = zu + st.ad * (-zu + (st.mu + zu * d) * detail::rcp_(su));
          const double ml = st.mu * detail::rcp_(snl), mu_ = st.mu * detail::rcp_(snu);
          vl = detail::dmax(detail::dmin(vl, st.ksig * ml), ml * iks);
          vu = detail::dmax(detail::dmin(vu, st.ksig * mu_), mu_ * iks);
          V.z[q] = zn; V.l[q] = l; V.u[q] = u; V.zl[q] = hl ? vl : 0.0; V.zu[q] = hu ? vu : 0.0;
          c.z[i] = V.z[q]; c.zL[i] = V.zl[q]; c.zU[i] = V.zu[q];
        }
      } else {
#pragma unroll
        for (int q = 0; q < NW; ++q) {
          const long i = zi(c, j, q);
          V.z[q] = c.z[i]; V.l[q] = c.lb[i]; V.u[q] = c.ub[i]; V.zl[q] = c.zL[i]; V.zu[q] = c.zU[i];
        }
      }
      HsPoint<Sys> P;
      set_time<Sys>(c.pp.get(), tq(j, c.h));
      double* pt = c.pt + j;
      const int K = c.K;
      if constexpr (MLP) {            // f, A, B come from the matrix-core pass; only the (closed-form) cost is per lane
#pragma unroll
        for (int q = 0; q < NS; ++q) P.x[q] = V.z[q];
#pragma unroll
        for (int q = 0; q < NU; ++q) P.u[q] = V.z[NS + q];
        Sys::cost_grad(P.x, P.u, c.pp.get(), &P.g, P.gw);
      } else {
        S::lin_point(V, c.pp.get(), P);
        if (TRAP && j == K - 1) fold_terminal<Sys>(P.x, P.u, c.pp.get(), wq(K, j, c.h), P.g, P.gw);   // trapezoidal.py:126-127
#pragma unroll
        for (int q = 0; q < NS; ++q) pt[(PF_F + q) * K] = P.f[q];
#pragma unroll
        for (int q = 0; q < NS * NS; ++q) pt[(PF_A + q) * K] = P.A[q];
#pragma unroll
        for (int q = 0; q < NS * NU; ++q) pt[(PF_B + q) * K] = P.B[q];
#pragma unroll
        for (int q = 0; q < Sys::NNZ2; ++q) pt[(PF_D2 + q) * K] = P.D2[q];
      }
#pragma unroll
      for (int q = 0; q < NW; ++q) pt[(PF_GW + q) * K] = P.gw[q];
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        typename S::BV b = S::bound_terms(V.z[q], V.l[q], V.u[q], V.zl[q], V.zu[q], cmax, cmin);
        pt[(PF_SIG + q) * K] = b.sigma; pt[(PF_G1 + q) * K] = b.g1; pt[(PF_ZLU + q) * K] = b.zlu;
        const bool fr = V.l[q] < V.u[q];
        const bool hl = fr && (V.l[q] > -INFINITY), hu = fr && (V.u[q] < INFINITY);
        sm += (hl ? V.zl[q] : 0.0) + (hu ? V.zu[q] : 0.0);
        nm += (hl ? 1 : 0) + (hu ? 1 : 0);
        const double sl = hl ? V.z[q] - V.l[q] : 1.0, su = hu ? V.u[q] - V.z[q] : 1.0;
        { int e_; slk *= frexp((sl > 0.0 ? sl : 1.0) * (su > 0.0 ? su : 1.0), &e_); sexp += e_; }
      }
      lg -= log(slk) + sexp * 0.6931471805599453;
      f += wq(K, j, c.h) * P.g;
    }
    o.f = wv_sum(f); o.cmax = wv_max(cmax); o.cmin = wv_min(cmin); o.sm = wv_sum(sm); o.nm = wv_isum(nm); o.lg = wv_sum(lg);
  }

  __device__ static inline void read_pt(const Ctx& c, int j, double* x, double* f, double* A, double* B) {
    const double* pt = c.pt + j;
    const int K = c.K;
#pragma unroll
    for (int q = 0; q < NS; ++q) { x[q] = c.z[(long)j * NS + q]; f[q] = pt[(PF_F + q) * K]; }
#pragma unroll
    for (int q = 0; q < NS * NS; ++q) A[q] = pt[(PF_A + q) * K];
#pragma unroll
    for (int q = 0; q < NS * NU; ++q) B[q] = pt[(PF_B + q) * K];
  }

  // ---- phase 2: lanes over intervals -- constraints, eliminations, adjoint maps -------------------------------
  // trapezoidal scheme (TrapCore::backward, os_solver.h): c_k = h/2 (f_k + f_{k+1}) - (x_{k+1} - x_k);
  // E = I - h/2 A_e, E dx_e = (I + h/2 A_s) dx_s + h/2 B_s du_s + h/2 B_e du_e + c_k; adjoint E^T lam_k = own_e + Pi_k,
  // Pi_{k-1} = (I + h/2 A_s)^T lam_k
  __device__ static void intervals_elim_trap(Ctx& c, double& c1o, double& cinfo) {
    using namespace detail;
    const int N = c.N, K = c.K;
    const double hh = 0.5 * c.h;
    double c1 = 0, cinf = 0;
    for (int k = c.lane; k < N; k += 64) {
      double xs[NS], fs[NS], As[NS * NS], Bs[NS * NU], xe[NS], fe[NS], Ae[NS * NS], Be[NS * NU];
      read_pt(c, k, xs, fs, As, Bs); read_pt(c, k + 1, xe, fe, Ae, Be);
      const double* pe = c.pt + (k + 1);
      const double we = wq(K, k + 1, c.h);
      double owne[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q)
        owne[q] = (k == N - 1 && c.term_pinned[q]) ? 0.0 : (we * pe[(PF_GW + q) * K] + pe[(PF_ZLU + q) * K]);
      double E[NS * NS], Ge[NS * NY1];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        const double cj = hh * (fs[r] + fe[r]) - (xe[r] - xs[r]);
        c1 += fabs(cj); cinf = dmax(cinf, fabs(cj));
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          E[r * NS + q] = ((r == q) ? 1.0 : 0.0) - hh * Ae[r * NS + q];
          Ge[r * NY1 + q] = ((r == q) ? 1.0 : 0.0) + hh * As[r * NS + q];
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) { Ge[r * NY1 + NS + a] = hh * Bs[r * NU + a]; Ge[r * NY1 + NW + a] = hh * Be[r * NU + a]; }
        Ge[r * NY1 + NY] = cj;
      }
      lu_factor<NS>(E);
      lu_solve<NS, NY1>(E, Ge);
      double* st = c.st + (long)k * SG_N;
#pragma unroll
      for (int q = 0; q < NS * NY1; ++q) st[SG_GE + q] = Ge[q];
      // lam_k = Ld Pi_k + ld0 with Ld = E^-T, ld0 = E^-T own_e
      double Ld[NS * NS], ld0[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double y[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) y[q] = (q == i) ? 1.0 : 0.0;
        lu_solve_t<NS>(E, y);
#pragma unroll
        for (int q = 0; q < NS; ++q) Ld[q * NS + i] = y[q];
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) v += Ld[r * NS + q] * owne[q];
        ld0[r] = v;
      }
#pragma unroll
      for (int q = 0; q < NS * NS; ++q) st[SG_LD + q] = Ld[q];
#pragma unroll
      for (int q = 0; q < NS; ++q) st[SG_LD0 + q] = ld0[q];
      double* M = c.r0 + (long)k * (NS * NS + NS);
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int q = 0; q <= NS; ++q) {
          double v = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) v += (((r == t) ? 1.0 : 0.0) + hh * As[t * NS + r]) * (q < NS ? Ld[t * NS + q] : ld0[t]);
          if (q < NS) M[r * NS + q] = v; else M[NS * NS + r] = v;
        }
      }
    }
    c1o = wv_sum(c1); cinfo = wv_max(cinf);
  }

  __device__ static void intervals_elim(Ctx& c, double& c1o, double& cinfo) {
    if constexpr (TRAP) { intervals_elim_trap(c, c1o, cinfo); return; }
    using namespace detail;
    const int N = c.N, K = c.K;
    const double h6 = c.h6, h8 = c.h8;
    double c1 = 0, cinf = 0;
    for (int k = c.lane; k < N; k += 64) {
      const int js = 2 * k, jm = 2 * k + 1, je = 2 * k + 2;
      double xs[NS], fs[NS], As[NS * NS], Bs[NS * NU], xm[NS], fm[NS], Am[NS * NS], Bm[NS * NU], xe[NS], fe[NS], Ae[NS * NS], Be[NS * NU];
      read_pt(c, js, xs, fs, As, Bs); read_pt(c, jm, xm, fm, Am, Bm); read_pt(c, je, xe, fe, Ae, Be);
      double dk[NS], ik[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        dk[q] = (xe[q] - xs[q]) - h6 * (fs[q] + 4.0 * fm[q] + fe[q]);
        ik[q] = xm[q] - 0.5 * (xs[q] + xe[q]) - h8 * (fs[q] - fe[q]);
        c1 += fabs(dk[q]) + fabs(ik[q]);
        cinf = dmax(cinf, dmax(fabs(dk[q]), fabs(ik[q])));
      }
      double Cm[NS * NS], Ne[NS * NS], Nsm[NS * NS], E[NS * NS];
#pragma unroll
      for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          const double id = (r == q) ? 1.0 : 0.0;
          Cm[r * NS + q] = 4.0 * h6 * Am[r * NS + q];
          Ne[r * NS + q] = 0.5 * id - h8 * Ae[r * NS + q];
          Nsm[r * NS + q] = 0.5 * id + h8 * As[r * NS + q];
        }
#pragma unroll
      for (int r = 0; r < NS; ++r)
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = ((r == q) ? 1.0 : 0.0) - h6 * Ae[r * NS + q];
#pragma unroll
          for (int t = 0; t < NS; ++t) s -= Cm[r * NS + t] * Ne[t * NS + q];
          E[r * NS + q] = s;
        }
      lu_factor<NS>(E);
      // (loads of the point records before the first store to the stage record: see points_hess)
      const double* pm = c.pt + jm; const double* pe = c.pt + je;
      const double wm = wq(K, jm, c.h), we = wq(K, je, c.h);
      double rm[NS], owne[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        rm[q] = wm * pm[(PF_GW + q) * K] + pm[(PF_ZLU + q) * K];
        owne[q] = (k == N - 1 && c.term_pinned[q]) ? 0.0 : (we * pe[(PF_GW + q) * K] + pe[(PF_ZLU + q) * K]);
      }
      double* st = c.st + (long)k * SG_N;
      // Ge | ge
      double Ge[NS * NY1];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = ((r == q) ? 1.0 : 0.0) + h6 * As[r * NS + q];
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Cm[r * NS + t] * Nsm[t * NS + q];
          Ge[r * NY1 + q] = s;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          double cbs = 0.0, cbe = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) { cbs += Cm[r * NS + t] * Bs[t * NU + a]; cbe += Cm[r * NS + t] * Be[t * NU + a]; }
          Ge[r * NY1 + NS + a] = h6 * Bs[r * NU + a] + h8 * cbs;
          Ge[r * NY1 + NS + NU + a] = 4.0 * h6 * Bm[r * NU + a];
          Ge[r * NY1 + NS + 2 * NU + a] = h6 * Be[r * NU + a] - h8 * cbe;
        }
        double s = -dk[r];
#pragma unroll
        for (int t = 0; t < NS; ++t) s -= Cm[r * NS + t] * ik[t];
        Ge[r * NY1 + NY] = s;
      }
      lu_solve<NS, NY1>(E, Ge);
#pragma unroll
      for (int q = 0; q < NS * NY1; ++q) st[SG_GE + q] = Ge[q];
      // Gm | gm
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double row[NY1];
#pragma unroll
        for (int q = 0; q <= NY; ++q) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Ne[r * NS + t] * Ge[t * NY1 + q];
          row[q] = s;
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) row[q] += Nsm[r * NS + q];
#pragma unroll
        for (int a = 0; a < NU; ++a) { row[NS + a] += h8 * Bs[r * NU + a]; row[NS + 2 * NU + a] -= h8 * Be[r * NU + a]; }
        row[NY] -= ik[r];
#pragma unroll
        for (int q = 0; q <= NY; ++q) st[SG_GM + r * NY1 + q] = row[q];
      }
      // adjoint maps: lam_d = Ld Pi + ld0, lam_i = Li Pi + li0, Pi_prev = M Pi + v
      double Ld[NS * NS], ld0[NS], Li[NS * NS], li0[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double y[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) y[q] = (q == i) ? 1.0 : 0.0;
        lu_solve_t<NS>(E, y);            // y = E^-T e_i  -> column i of E^-T
#pragma unroll
        for (int q = 0; q < NS; ++q) Ld[q * NS + i] = -y[q];
      }
      {
        double t0[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = owne[q];
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Ne[t * NS + q] * rm[t];
          t0[q] = s;
        }
#pragma unroll
        for (int r = 0; r < NS; ++r) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < NS; ++q) s += Ld[r * NS + q] * t0[q];
          ld0[r] = s;
        }
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Cm[t * NS + r] * Ld[t * NS + q];
          Li[r * NS + q] = s;
        }
        double s = -rm[r];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += Cm[t * NS + r] * ld0[t];
        li0[r] = s;
      }
#pragma unroll
      for (int q = 0; q < NS * NS; ++q) { st[SG_LD + q] = Ld[q]; st[SG_LI + q] = Li[q]; }
#pragma unroll
      for (int q = 0; q < NS; ++q) { st[SG_LD0 + q] = ld0[q]; st[SG_LI0 + q] = li0[q]; }
      // Pi_prev = (-I - h6 As^T) lam_d + (-I/2 - h8 As^T) lam_i
      double* M = c.r0 + (long)k * (NS * NS + NS);
#pragma unroll
      for (int r = 0; r < NS; ++r) {
#pragma unroll
        for (int q = 0; q <= NS; ++q) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) {
            const double sd = ((r == t) ? -1.0 : 0.0) - h6 * As[t * NS + r];
            const double si = ((r == t) ? -0.5 : 0.0) - h8 * As[t * NS + r];
            s += sd * (q < NS ? Ld[t * NS + q] : ld0[t]) + si * (q < NS ? Li[t * NS + q] : li0[t]);
          }
          if (q < NS) M[r * NS + q] = s; else M[NS * NS + r] = s;
        }
      }
    }
    c1o = wv_sum(c1); cinfo = wv_max(cinf);
  }

  // adjoint recurrence Pi_{k-1} = M_k Pi_k + v_k, k = N-1 .. 0: lane r < NS owns row r of M|v (LDS, prefetched one
  // stage ahead); Pi travels between lanes with v_readlane, so a stage is NS FMAs + NS readlanes and no barrier.
#ifndef MYR_RECUR_SEQ
  // Wave-scan form: the recurrence is affine, so the N dependent stages become a scan of map compositions, 64 stages at a
  // time (6 DPP rounds of an NS x NS product per block).  Lane j takes stage base + 63 - j, which turns the suffix scan
  // over the stages into the prefix scan over the lanes that the DPP idiom provides; lane j ends with Pi of its stage.
  __device__ static void adjoint_recur(Ctx& c, const double* nuT) {
    constexpr int MV = NS * NS + NS;
    const int lane = c.lane, N = c.N;
    double piS[NS];                          // Pi of the stage above the block (uniform)
#pragma unroll
    for (int q = 0; q < NS; ++q) piS[q] = c.term_pinned[q] ? nuT[q] : 0.0;
    for (int base = ((N - 1) / 64) * 64; base >= 0; base -= 64) {
      const int k = base + 63 - lane;
      const bool on = k < N;
      double A[NS * NS], b[NS];
      const double* M = c.r0 + (long)(on ? k : 0) * MV;
#pragma unroll
      for (int q = 0; q < NS * NS; ++q) A[q] = on ? M[q] : (((q / NS) == (q % NS)) ? 1.0 : 0.0);
#pragma unroll
      for (int q = 0; q < NS; ++q) b[q] = on ? M[NS * NS + q] : 0.0;
      affine_prefix_scan_dpp<NS>(A, b);
      double lo[NS];                         // Pi_{k-1}
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double v = b[r];
#pragma unroll
        for (int q = 0; q < NS; ++q) v += A[r * NS + q] * piS[q];
        lo[r] = v;
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const double t = wv_up(lo[q], 1);       // Pi_k = (Pi_{k'-1} of the lane below, whose stage is k' = k + 1)
        const double own = (lane == 0) ? piS[q] : t;
        if (on) c.sPi[k * NS + q] = own;
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) piS[q] = __shfl(lo[q], 63, 64);
    }
  }
#else
  __device__ static void adjoint_recur(Ctx& c, const double* nuT) {
    // every row of 16 lanes repeats the computation of lanes 0..NS-1 (the broadcast below is per row); row 0 stores
    const int lane = c.lane, l16 = lane & 15, r = l16 < NS ? l16 : 0;
    constexpr int MV = NS * NS + NS, UB = 4;       // UB stages per batch of LDS reads: their latency is paid once per batch
    double piq[NS], own = 0.0;
#pragma unroll
    for (int q = 0; q < NS; ++q) { piq[q] = c.term_pinned[q] ? nuT[q] : 0.0; own = (q == l16) ? piq[q] : own; }
    for (int kb = c.N - 1; kb >= 0; kb -= UB) {
      double row[UB][NS + 1];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = kb - u > 0 ? kb - u : 0;      // (stages below 0: a valid row, not used)
        const double* M = c.r0 + (long)k * MV;
#pragma unroll
        for (int q = 0; q < NS; ++q) row[u][q] = M[r * NS + q];
        row[u][NS] = M[NS * NS + r];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = kb - u;
        if (k < 0) break;
        if (lane < NS) c.sPi[k * NS + lane] = own;
        double v = row[u][NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) v += row[u][q] * piq[q];
        RowBcast<NS>::all(v, piq);        // lanes 0..NS-1 hold the rows: broadcast within each row of 16 lanes
        own = v;
      }
    }
  }

#endif

  // ---- phase 3: lanes over intervals -- multipliers ------------------------------------------------------------
  __device__ static void intervals_lambda(Ctx& c, double& lam_inf, double& sum_mult) {
    double li = 0, sm = 0;
    for (int k = c.lane; k < c.N; k += 64) {
      double st[SG_QM];                        // the adjoint maps of this interval, loaded before the first store (see points_hess)
      {
        const double* sg = c.st + (long)k * SG_N;
#pragma unroll
        for (int q = SG_LD; q < SG_QM; ++q) st[q] = sg[q];
      }
      double pi[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) pi[q] = c.sPi[k * NS + q];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double d = st[SG_LD0 + r], i2 = TRAP ? 0.0 : st[SG_LI0 + r];
#pragma unroll
        for (int q = 0; q < NS; ++q) { d += st[SG_LD + r * NS + q] * pi[q]; if (!TRAP) i2 += st[SG_LI + r * NS + q] * pi[q]; }
        c.lam[(long)k * NS + r] = d;
        if (!TRAP) c.lam[(long)c.N * NS + (long)k * NS + r] = i2;
        li = detail::dmax(li, detail::dmax(fabs(d), fabs(i2)));
        sm += fabs(d) + fabs(i2);
      }
    }
    lam_inf = wv_max(li); sum_mult = wv_sum(sm);
  }

  // ---- phase 4: lanes over points -- Lagrangian Hessian, gradient columns, control-row stationarity ------------
  __device__ static void points_hess(Ctx& c, double& stat) {
    node_pass<2>(c, 0.0);
    const int N = c.N, K = c.K;
    const double h6 = c.h6, h8 = c.h8;
    double st_ = 0;
    for (int j = c.lane; j < K; j += 64) {
      const double* pt = c.pt + j;
      double a[NS];
      if (TRAP) {           // a_j = h/2 (lam_{j-1} + lam_j): TrapCore's mue = mu_c + h/2 lam
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = 0.0;
          if (j >= 1) s += 0.5 * c.h * c.lam[(long)(j - 1) * NS + q];
          if (j < N) s += 0.5 * c.h * c.lam[(long)j * NS + q];
          a[q] = s;
        }
      } else if (j & 1) {
        const int k = (j - 1) >> 1;
#pragma unroll
        for (int q = 0; q < NS; ++q) a[q] = -4.0 * h6 * c.lam[(long)k * NS + q];
      } else {
        const int kL = (j >> 1) - 1, kR = j >> 1;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = 0.0;
          if (kL >= 0) s += -h6 * c.lam[(long)kL * NS + q] + h8 * c.lam[(long)N * NS + (long)kL * NS + q];
          if (kR < N) s += -h6 * c.lam[(long)kR * NS + q] - h8 * c.lam[(long)N * NS + (long)kR * NS + q];
          a[q] = s;
        }
      }
      const double wj = wq(K, j, c.h);
      // every load of this point BEFORE the first store: the stores below may alias what is loaded as far as the compiler
      // can tell (they do alias the f | A fields of pt), so a load placed between them waits for a full memory round trip
      double gw[NW], D2[ND2], W[NW * NW], sig[NW], g1v[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) { gw[q] = pt[(PF_GW + q) * K]; sig[q] = pt[(PF_SIG + q) * K]; g1v[q] = pt[(PF_G1 + q) * K]; }
#pragma unroll
      for (int q = 0; q < ND2; ++q) D2[q] = pt[(PF_D2 + q) * K];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double r = wj * gw[NS + u] + pt[(PF_ZLU + NS + u) * K];
#pragma unroll
        for (int t = 0; t < NS; ++t) r += pt[(PF_B + t * NU + u) * K] * a[t];
        st_ = detail::dmax(st_, fabs(r));
      }
      {
        double xj[NS], uj[NU];
#pragma unroll
        for (int q = 0; q < NS; ++q) xj[q] = c.z[zi(c, j, q)];
#pragma unroll
        for (int q = 0; q < NU; ++q) uj[q] = c.z[zi(c, j, NS + q)];
        if constexpr (MLP) Sys::hessian_packed(xj, uj, D2, wj, W);   // D2: the network's contraction, from the matrix-core pass
        else Sys::hessian(xj, uj, c.pp.get(), D2, a, wj, W);
      }
      double* hr = c.hr + (long)j * HR_N;
      const bool last = (j == K - 1);
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        const bool zr = last && r < NS && c.term_pinned[r];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
          const bool zq = last && q < NS && c.term_pinned[q];
          hr[HR_H + r * NW + q] = (zr || zq) ? 0.0 : (W[r * NW + q] + ((r == q) ? sig[r] : 0.0));
        }
        hr[HR_G0 + r] = zr ? 0.0 : wj * gw[r];
        hr[HR_G1 + r] = zr ? 0.0 : g1v[r];
      }
    }
    stat = wv_max(st_);
  }

  // ---- phase 5: lanes over intervals -- midpoint Schur terms Qm = Gm^^T (H_m + delta I) Gm^, qcm ----------------
  __device__ static void intervals_qm(Ctx& c, double delta) {
    if constexpr (TRAP) { (void)c; (void)delta; return; } else {
    for (int k = c.lane; k < c.N; k += 64) {
      double* st = c.st + (long)k * SG_N;
      const double* hr = c.hr + (long)(2 * k + 1) * HR_N;
      double Gm[NS * NY1], H[NW * NW], T1[NW * NY1];
#pragma unroll
      for (int q = 0; q < NS * NY1; ++q) Gm[q] = st[SG_GM + q];
#pragma unroll
      for (int q = 0; q < NW * NW; ++q) H[q] = hr[HR_H + q];
#pragma unroll
      for (int q = 0; q < NW; ++q) H[q * NW + q] += delta;
#pragma unroll
      for (int r = 0; r < NW; ++r)
#pragma unroll
        for (int q = 0; q <= NY; ++q) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) s += H[r * NW + t] * Gm[t * NY1 + q];
#pragma unroll
          for (int a = 0; a < NU; ++a) if (q == NS + NU + a) s += H[r * NW + NS + a];
          T1[r * NY1 + q] = s;
        }
#pragma unroll
      for (int r = 0; r < NY; ++r) {
#pragma unroll
        for (int q = 0; q < NY; ++q) {
          double s = 0.0;
#pragma unroll
          for (int t = 0; t < NS; ++t) s += Gm[t * NY1 + r] * T1[t * NY1 + q];
#pragma unroll
          for (int a = 0; a < NU; ++a) if (r == NS + NU + a) s += T1[(NS + a) * NY1 + q];
          st[SG_QM + r * NY + q] = s;
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int t = 0; t < NS; ++t) {
          s0 += Gm[t * NY1 + r] * (T1[t * NY1 + NY] + hr[HR_G0 + t]);
          s1 += Gm[t * NY1 + r] * hr[HR_G1 + t];
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) if (r == NS + NU + a) { s0 += T1[(NS + a) * NY1 + NY] + hr[HR_G0 + NS + a]; s1 += hr[HR_G1 + NS + a]; }
        st[SG_QCM + r * 2 + 0] = s0; st[SG_QCM + r * 2 + 1] = s1;
      }
    }
    }
  }

  // ---- phase 6: Riccati sweep, sequential over stages, ONE COLUMN PER LANE ---------------------------------------
  // The stage update  [Q | qc] = [Qm | qcm] + Ge^^T (P' [Ge^ | ge^] + [0 | pc'])  followed by the elimination of
  // q = (du_m, du_e) acts column by column, so lane j < NY carries column j of Q, lane NY + cc carries column cc of
  // the right-hand sides (gradient, mu, nu_1..nu_NS), and between stages lane j < NW keeps column j of P and lane
  // NY + cc column cc of pc in registers.  Every lane runs the SAME instruction stream (no element-dependent
  // branches); the only shared data per stage are P' (NW x NW, through LDS), Ge|ge (LDS) and a handful of entries of
  // the q columns, which are read with v_readlane.  One barrier per stage.
  // Returns the number of regularised pivots (wave-uniform); aborts at the first one when `abort_on_reg`.
  __device__ static inline double rdlane(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
  }

  // first point: dx_0 = 0; add its control terms, eliminate du_0 (every lane redundantly; tiny).  P, pc, Tnu come from
  // the sweep through LDS (sP, sPc, sTnu).
  __device__ static int riccati_first_point(Ctx& c, const HsSolveOpts& o, double delta, int nreg) {
    using namespace detail;
    const int lane = c.lane;
    {
      const double* hr = c.hr;   // point 0
      double Puu[NU * NU], ku[NU * NC], pun[NU * NS];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int b = 0; b < NU; ++b) Puu[a * NU + b] = c.sP[(NS + a) * NW + NS + b] + hr[HR_H + (NS + a) * NW + NS + b] + ((a == b) ? delta : 0.0);
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)
          ku[a * NC + cc] = c.sPc[(NS + a) * NC + cc] + (cc == 0 ? hr[HR_G0 + NS + a] : (cc == 1 ? hr[HR_G1 + NS + a] : 0.0));
#pragma unroll
        for (int i = 0; i < NS; ++i) pun[a * NS + i] = ku[a * NC + 2 + i];
      }
      nreg += chol_reg<NU>(Puu, o.reg_floor);
      chol_solve<NU, NC>(Puu, ku);
      wsync();
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
          for (int cc = 0; cc < NC; ++cc) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < NU; ++a) s += pun[a * NS + i] * ku[a * NC + cc];
            c.sTnu[i * NC + cc] -= s;
          }
#pragma unroll
        for (int i = 0; i < NU * NC; ++i) c.sKu[i] = ku[i];
      }
      wsync();
    }
    return nreg;
  }

  __device__ static int riccati(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg) {
    using namespace detail;
    static_assert(NY + NC <= 64, "one column per lane");
    const int lane = c.lane, N = c.N;
    const bool isP = lane < NW;                 // column `lane` of P / of H_e
    const bool isY = lane < NY;                 // column `lane` of Q
    const int cc = lane - NY;                   // column cc of the right-hand sides when 0 <= cc < NC
    const bool isC = cc >= 0 && cc < NC;
    const bool isVal = isP || isC;
    // per-lane addressing of the stage inputs: base pointer + stride per stage; lanes without an input (the nu
    // columns have no qcm, lanes past the columns have nothing) read a block of zeros, so every load is unconditional
    const bool m_on = !TRAP && (isY || (isC && cc < 2));   // Qm column (unit stride; Qm is symmetric) | qcm column (stride 2); none for the trapezoidal scheme
    const double* m_ptr = m_on ? c.st + (isY ? SG_QM + lane * NY : SG_QCM + (cc == 1 ? 1 : 0)) : c.zr;
    const int m_str = isY ? 1 : 2;
    const long m_step = m_on ? SG_N : 0;
    const bool h_on = isP || (isC && cc < 2);   // H_e column | g0 | g1 of the end point (point 2k+2 for stage k)
    const double* h_ptr = h_on ? c.hr + (isP ? HR_H + lane * NW : (cc == 1 ? HR_G1 : HR_G0)) : c.zr;
    // the end point of stage k is point 2k + 2 (Hermite-Simpson: knots at the even points) or k + 1 (trapezoidal: K = N + 1 points)
    constexpr long H_PTS = TRAP ? 1 : 2;
    const long h_step = h_on ? H_PTS * HR_N : 0;
    const int v_col = isY ? lane : NY;          // own column of Ge^ (Q lanes), ge^ (gradient column), none otherwise
    const double v_on = (isY || cc == 0) ? 1.0 : 0.0;
    const double c_on = isC ? 1.0 : 0.0;
    constexpr int NGE = NS * NY1, NGL = (NGE + 63) / 64;

    // delta on the own diagonal entry; the pinned terminal diagonal carries rho instead (rho - delta + delta below)
    double val[NW], dvec[NW], tnuB[NS], tnuA = 0.0;
#pragma unroll
    for (int r = 0; r < NW; ++r) {
      const bool pin = r < NS && c.term_pinned[r < NS ? r : 0];
      dvec[r] = (isP && r == lane) ? delta : 0.0;
      val[r] = (isP && r == lane && pin) ? o.rho_term - delta : ((isC && cc == 2 + r && pin) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) tnuB[i] = 0.0;
    int nreg = 0;
    double m_pre[NY], h_pre[NW], g_pre[NGL];
    {
      const double* mp = m_ptr + (long)(N - 1) * m_step;
      const double* hp = h_ptr + (long)(N - 1) * h_step + (h_on ? H_PTS * HR_N : 0);
      const double* st = c.st + (long)(N - 1) * SG_N;
#pragma unroll
      for (int r = 0; r < NY; ++r) m_pre[r] = mp[r * m_str];
#pragma unroll
      for (int r = 0; r < NW; ++r) h_pre[r] = hp[r];
#pragma unroll
      for (int t = 0; t < NGL; ++t) { const int e = lane + 64 * t; g_pre[t] = st[SG_GE + (e < NGE ? e : NGE - 1)]; }
    }
    for (int k = N - 1; k >= 0; --k) {
      double m[NY];
#pragma unroll
      for (int r = 0; r < NY; ++r) m[r] = m_pre[r];
      // (a) P' = P + H_e + delta I, pc' = pc + gbar_e; share P' and Ge|ge
#pragma unroll
      for (int r = 0; r < NW; ++r) val[r] += h_pre[r] + dvec[r];
      if (isP) {
#pragma unroll
        for (int r = 0; r < NW; ++r) c.sP[lane * NW + r] = val[r];
      }
#pragma unroll
      for (int t = 0; t < NGL; ++t) { const int e = lane + 64 * t; if (e < NGE) c.sGe[e] = g_pre[t]; }
      if (k > 0) {       // prefetch the next stage while this one is processed
        const double* mp = m_ptr + (long)(k - 1) * m_step;
        const double* hp = h_ptr + (long)(k - 1) * h_step + (h_on ? H_PTS * HR_N : 0);
        const double* st = c.st + (long)(k - 1) * SG_N;
#pragma unroll
        for (int r = 0; r < NY; ++r) m_pre[r] = mp[r * m_str];
#pragma unroll
        for (int r = 0; r < NW; ++r) h_pre[r] = hp[r];
#pragma unroll
        for (int t = 0; t < NGL; ++t) { const int e = lane + 64 * t; g_pre[t] = st[SG_GE + (e < NGE ? e : NGE - 1)]; }
      }
      wsync();
      // (b) tv = P' v + (pc' on the right-hand-side lanes),  v = own column of [Ge^ | ge^]
      double v[NW], tv[NW];
#pragma unroll
      for (int t = 0; t < NS; ++t) v[t] = v_on * c.sGe[t * NY1 + v_col];
#pragma unroll
      for (int a = 0; a < NU; ++a) v[NS + a] = (lane == NY - NU + a) ? 1.0 : 0.0;     // selector of du_e, the last NU entries of y
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        double s = c_on * val[r];
#pragma unroll
        for (int q = 0; q < NW; ++q) s += c.sP[r * NW + q] * v[q];
        tv[r] = s;
      }
      // (c) own column of [Q | qc] = [Qm | qcm] + Ge^^T tv
      double col[NY];
#pragma unroll
      for (int r = 0; r < NY; ++r) {
        double s = m[r];
#pragma unroll
        for (int t = 0; t < NS; ++t) s += c.sGe[t * NY1 + r] * tv[t];
        if (r >= NY - NU) s += tv[NS + (r - (NY - NU))];
        col[r] = s;
      }
      // terminal-multiplier bookkeeping, part 1 (lanes NY+2+i): Tnu[i][0] += ge^T pc'[:, nu_i]
      {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NS; ++t) s += c.sGe[t * NY1 + NY] * val[t];
        tnuA += s;
      }
      // (d) L D L^T of Qqq (identical on every lane), own column of the gains K | kc
      double Lq[NQ * NQ], dinv[NQ];
#pragma unroll
      for (int r = 0; r < NQ; ++r)
#pragma unroll
        for (int q = 0; q <= r; ++q) Lq[r * NQ + q] = rdlane(col[NW + r], NW + q);
      nreg += ldl_reg<NQ>(Lq, dinv, o.reg_floor);
      if (nreg > 0 && abort_on_reg) return nreg;
      double kk[NQ];
#pragma unroll
      for (int r = 0; r < NQ; ++r) kk[r] = col[NW + r];
      ldl_solve<NQ>(Lq, dinv, kk);
      {
        double* Kst = c.kg + (long)k * KSTR;
        if (isP) {
#pragma unroll
          for (int r = 0; r < NQ; ++r) Kst[r * NW + lane] = kk[r];
        } else if (isC) {
#pragma unroll
          for (int r = 0; r < NQ; ++r) Kst[NQ * NW + r * NC + cc] = kk[r];
        }
      }
      // (e) value function: own column of P = Qss - Qsq K,  pc = qc_s - Qsq kc   (lanes without a column: unused)
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        double s = col[r];
#pragma unroll
        for (int t = 0; t < NQ; ++t) s -= rdlane(col[r], NW + t) * kk[t];
        val[r] = s;
      }
      // terminal-multiplier bookkeeping, part 2 (lane NY+cc holds Tnu[:, cc]): Tnu[i][cc] -= qc_q[:, nu_i]^T kc[:, cc]
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < NQ; ++t) s += rdlane(col[NW + t], NY + 2 + i) * kk[t];
        tnuB[i] -= s;
      }
      wsync();
    }
    // hand P, pc, Tnu to the first-point step through LDS
    if (isP) {
#pragma unroll
      for (int r = 0; r < NW; ++r) c.sP[r * NW + lane] = val[r];
    }
    if (isC) {
#pragma unroll
      for (int r = 0; r < NW; ++r) c.sPc[r * NC + cc] = val[r];
#pragma unroll
      for (int i = 0; i < NS; ++i) c.sTnu[i * NC + cc] = tnuB[i];
    }
    wsync();
    if (isC && cc >= 2) c.sTnu[(cc - 2) * NC + 0] += tnuA;
    wsync();
    return riccati_first_point(c, o, delta, nreg);
  }

  // ---- phase 6, matrix-core form (NU == 1, NS <= 4): the SAME stage algebra as riccati() on v_mfma_f64_16x16x4_f64 -------
  // The stage update is three small dense products,
  //     R~      = P' [Ge^ | ge^] + [0 | pc']                          (NW x (NY+NC))
  //     [Q|qc]  = [Qm | qcm] + Ge^^T R~                               (NY x (NY+NC))
  //     [P|pc]  = [Qss | qc_s] - Qsq (Qqq^-1 [Qqs | qc_q])            (NW x (NW+NC))   (+ the dual bookkeeping rows)
  // and one 16x16 matrix-core tile holds all of a stage: the SAME slot placement is used for rows and for columns,
  //     0..3 dx_s | 4,5 du_s (twice) | 6 rhs "1" (where ge enters) | 7 rhs mu | 8,9 du_e (twice) | 10,11,14,15 rhs nu_1..4 |
  //     12,13 du_m (twice),
  // chosen for the register layout of the instruction (A[i][k] and B[k][j] one value per lane at lane 16k+i / 16k+j; C/D
  // element (i,j) at lane 16 (i%4) + j, register i/4 -- probed on the hardware, tools/dev/mfma/probe_f64.hip):
  //   * rows 0..3 of a result (register 0) ARE the B operand of the next product, and, P and Q being symmetric, also its A
  //     operand: the three products chain without any data movement between lanes;
  //   * the rows of the two eliminated controls du_m (12,13) and du_e (8,9) share lanes (registers 3 and 2 of lane groups
  //     0 and 1), so every column's gain is a per-lane 2x2 solve; keeping du_s, du_m, du_e TWICE gives lane groups 0 and 1
  //     each their own copy, which is exactly where the rank-2 update wants its two K-slots;
  //   * the dual bookkeeping Tnu (the rows ge^T pc' and -qc_q^T kc of riccati()) falls out of the same instructions as
  //     extra result rows (6 and 10,11,14,15) that are otherwise unused.
  // Per stage: 3 MFMA + ~70 VALU + 6 v_readlane, no LDS, no barrier (riccati(): ~430 instructions, 2 barriers).
#ifndef MYR_RICCATI_PF
#define MYR_RICCATI_PF 4
#endif
#ifndef MYR_RICCATI_INLINE
#define MYR_RICCATI_INLINE
#endif
  static_assert(MYR_RICCATI_PF >= 2 && MYR_RICCATI_PF <= 8, "the prefetch ring needs two slots; PADF covers eight");
  static constexpr bool MFMA_RICCATI = (NU == 1 && NS <= 4);
  typedef double mfma_d4 __attribute__((ext_vector_type(4)));
  // lane l <- lane l-4 within its row of 16 lanes (0 where l%16 < 4).  Inline asm on purpose: the compiler sinks the
  // DPP builtin into the divergent branch of the select that consumes it, and a DPP read of a lane that EXEC has
  // switched off returns 0 -- the shifted value must be produced with every lane enabled.
  __device__ static inline double dpp_row_shr4(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v), rlo, rhi;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %1, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
    return __hiloint2double(rhi, rlo);
  }
  // every lane <- lane N of its own row of 16 (DPP row_newbcast: stays in the vector pipe, ~10 cycles; v_readlane goes
  // through the scalar file and costs ~55 cycles before a vector instruction can use the value)
  template <int LANE>
  __device__ static inline double dpp_row_bcast(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v), rlo, rhi;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_mov_b32_dpp %1, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf"
                 : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi), "n"(LANE));
    return __hiloint2double(rhi, rlo);
  }
  __device__ static inline double dpp_row_shr8(double v) {      // lane l <- lane l-8 within its row of 16
    int lo = __double2loint(v), hi = __double2hiint(v), rlo, rhi;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
                 "v_mov_b32_dpp %1, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                 : "=&v"(rlo), "=&v"(rhi) : "v"(lo), "v"(hi));
    return __hiloint2double(rhi, rlo);
  }
  __device__ static MYR_RICCATI_INLINE int riccati_mfma(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg) {
    static_assert(!TRAP, "Hermite-Simpson form");
    using namespace detail;
    const int lane = c.lane, N = c.N;
    const int g = lane >> 4, j = lane & 15;
    // what column / row slot j stands for
    const int scol = j < 4 ? (j < NS ? j : -1) : (j < 6 ? NS : -1);                       // index into s = (dx, du)
    const int ycol = scol >= 0 ? scol : ((j == 12 || j == 13) ? NS + 1 : ((j == 8 || j == 9) ? NS + 2 : -1));   // into y
    const int cc = j == 6 ? 0 : (j == 7 ? 1 : (j == 10 ? 2 : (j == 11 ? 3 : (j == 14 ? 4 : (j == 15 ? 5 : -1)))));
    const int rcc = (cc >= 0 && cc < NC) ? cc : -1;                                        // right-hand-side column
    const bool rowx = g < NS;                 // register-0 row g is a state row
    // ---- per-lane addresses of the stage inputs (stage N-1), zero block for the slots that hold no input ----
    const double* he = c.hr + (long)(2 * (N - 1) + 2) * HR_N;   // end point of the stage
    const double* hm = he - HR_N;                                 // its midpoint
    const double* st = c.st + (long)(N - 1) * SG_N;
    auto hsel = [&](const double* rec, int row, bool on) -> const double* {   // element (row, column slot j) of [H | g0 | g1]
      if (!on) return c.zr;
      if (scol >= 0) return rec + HR_H + scol * NW + row;
      if (rcc == 0) return rec + HR_G0 + row;
      if (rcc == 1) return rec + HR_G1 + row;
      return c.zr;
    };
    auto gsel = [&](int off) -> const double* {                  // element (row g, column slot j) of [G | g], G = Ge or Gm
      if (!rowx) return c.zr;
      if (ycol >= 0) return st + off + g * NY1 + ycol;
      if (rcc == 0) return st + off + g * NY1 + NY;
      return c.zr;
    };
    // six input streams per stage: H_e rows 0..3 | H_e row du | Ge^ | H_m rows 0..3 | H_m row du | Gm^
    const double* ptr[6] = {hsel(he, g, rowx), hsel(he, NS, g < 2), gsel(SG_GE), hsel(hm, g, rowx), hsel(hm, NS, g < 2), gsel(SG_GM)};
    long stp[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) stp[q] = (ptr[q] == c.zr) ? 0 : ((q == 2 || q == 5) ? (long)SG_N : 2L * HR_N);
    // ---- state: X = [P | pc] in result layout (rows 0..3 register 0, row du twice in register 1 of groups 0, 1) ----
    const bool pinr = rowx && c.term_pinned[rowx ? g : 0];
    double X0 = (pinr && scol == g) ? o.rho_term - delta : ((pinr && rcc == 2 + g) ? 1.0 : 0.0), X1 = 0.0;
    const double dv0 = (rowx && scol == g) ? delta : 0.0, dv1 = (g < 2 && scol == NS) ? delta : 0.0;
    // Lane selections are per-lane 0/1 factors folded into multiply-adds (one fp64 instruction instead of two 32-bit
    // selects plus an add).  A factor 0 meets only finite values: the unused rows / columns of the tile hold finite
    // combinations of the inputs (if an input is not finite the solve is reported NAN anyway).
    const double f_a1 = j < 6 ? 1.0 : 0.0;                                  // A operand of the R~ products: P' / H_m columns
    const double f_keep = rcc >= 0 ? 1.0 : 0.0;                             // C operand: the right-hand-side columns pass
    const double f_she = (j == 8 || j == 9) ? 1.0 : 0.0, f_shm = (j == 12 || j == 13) ? 1.0 : 0.0;   // selector columns
    const double f_x1 = g < 2 ? 1.0 : 0.0, f_t1 = g == 2 ? 1.0 : 0.0, f_t23 = g >= 2 ? 1.0 : 0.0;    // rows of registers 1..3
    const bool a3_on = g < 2 && (j < 6 || j == 10 || j == 11 || j == 14 || j == 15);
    const double f_a3m = (a3_on && g == 0) ? -1.0 : 0.0, f_a3e = (a3_on && g == 1) ? -1.0 : 0.0;
    // where this lane's gain goes in the per-stage record K | kc (group 0 only, one copy of the du_s column); the other
    // lanes store unconditionally too, into the 2 spare doubles behind the block of zeros (never read)
    const int k_off = (g == 0 && scol >= 0 && j != 5) ? scol : ((g == 0 && rcc >= 0) ? NQ * NW + rcc : -1);
    const int k_str = k_off < 0 ? 1 : ((scol >= 0) ? NW : NC);
    double* k_ptr = k_off >= 0 ? c.kg + (long)(N - 1) * KSTR + k_off : c.zr + ZR;
    const long k_step = k_off >= 0 ? KSTR : 0;
    double reg_floor = o.reg_floor;
    asm volatile("" : "+v"(reg_floor));       // own register: otherwise every use reloads the spilled 16-SGPR argument block
    int nreg = 0;
    // Stage inputs are prefetched PF stages ahead into registers (6 doubles per stage).  The loop is unrolled by PF so that the
    // ring of prefetch registers is addressed statically.  Stages below 0 are read too (valid scratch in front of the
    // records) and never used.
    constexpr int PF = MYR_RICCATI_PF;
    double in[PF][6];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int q = 0; q < 6; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
    }
    // midpoint part of a stage, Qm^ = Gm^^T (H_m + delta I) [Gm^ | gm^] + Gm^^T [g0 g1]: the same two products as the
    // end-point part with H_m in the place of P'; it depends on loaded data only, so stage k-1's is issued while stage
    // k waits for its pivots (this is what intervals_qm() computes, lanes over intervals, for riccati())
    auto mid_part = [&](double n0, double n1, double Gm) -> mfma_d4 {
      n0 += dv0; n1 += dv1;
      const double s0 = dpp_row_shr8(n0), s1 = dpp_row_shr8(n1);
      mfma_d4 C;
      C[0] = fma(s0, f_shm, n0 * f_keep);
      C[1] = fma(s1, f_shm, n1 * f_keep);
      C[2] = 0.0; C[3] = 0.0;
      const mfma_d4 R = __builtin_amdgcn_mfma_f64_16x16x4f64(n0 * f_a1, Gm, C, 0, 0, 0);
      mfma_d4 C2;
      C2[0] = 0.0; C2[1] = 0.0; C2[2] = 0.0; C2[3] = R[1];          // selector row: the du_m rows take R's row du
      return __builtin_amdgcn_mfma_f64_16x16x4f64(Gm, R[0], C2, 0, 0, 0);
    };
    mfma_d4 Qm = mid_part(in[0][3], in[0][4], in[0][5]);
    mfma_d4 D3 = {X0, X1, 0.0, 0.0};          // previous stage's result: rows 0..5 [P | pc], rows 6, 10, 11, 14, 15 Tnu
    for (int kb = N - 1; kb >= 0; kb -= PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int k = kb - u;
        if (k < 0) break;
#ifdef MYR_RICC_PROBE
        unsigned long long tp[8]; tp[0] = __builtin_amdgcn_s_memtime();
#define MYR_TP(i) tp[i] = __builtin_amdgcn_s_memtime();
#else
#define MYR_TP(i)
#endif
        // (a) P' = P + H_e + delta I, pc' = pc + gbar_e
        X0 = D3[0] + (in[u][0] + dv0); X1 = fma(D3[1], f_x1, in[u][1] + dv1);
        const double G = in[u][2];
        // inputs of the next stage's midpoint part (slot u+1 of the ring, already loaded)
        const double nn0 = in[(u + 1) % PF][3], nn1 = in[(u + 1) % PF][4], nGm = in[(u + 1) % PF][5];
        // refill this slot with stage k - PF
#pragma unroll
        for (int q = 0; q < 6; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
        // (b) R~ = P' [Ge^ | ge^] + [0 | pc']; the selector row of Ge^ (du_e) is the shifted column du of P'
        const double sh0 = dpp_row_shr4(X0), sh1 = dpp_row_shr4(X1);
        mfma_d4 C1;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        C1[2] = 0.0; C1[3] = 0.0;
        MYR_TP(1)
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
        // (c) [Q | qc] = Qm^ + Ge^^T R~  (selector row: the du_e rows take R~'s row du); rows 6, 10.. carry Tnu
        mfma_d4 C2;
        C2[0] = Qm[0]; C2[1] = fma(D3[1], f_t1, Qm[1]); C2[2] = fma(D3[2], f_t23, Qm[2]) + D1[1]; C2[3] = fma(D3[3], f_t23, Qm[3]);
        MYR_TP(2)
        const mfma_d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(G, D1[0], C2, 0, 0, 0);
        // (d) this column's gains [K | kc] = Qqq^-1 [Qqs | qc_q].  Pivots of the L D L^T of Qqq as in ldl_reg: d0 = q00,
        // d1 = q11 - q10^2 / q00 = det / q00, both required > reg_floor.  When they are (always, except inside the inertia
        // correction's probing), the 2x2 solve is Cramer's rule with ONE reciprocal, 1 / det, whose dependent chain
        // (product, fma, rcp + Newton, product) is a third of the factor-and-substitute one; the numerators do not depend
        // on it.
        const double q00 = rdlane(D2[3], 12), q10 = rdlane(D2[2], 12), q11 = rdlane(D2[2], 8);
        MYR_TP(3)
#ifdef MYR_RICC_PROBE
        { double t_ = q00 + q10 + q11; asm volatile("" : "+v"(t_)); tp[7] = __builtin_amdgcn_s_memtime(); }
#endif
        const double det = fma(q00, q11, -(q10 * q10));
        const double rdet = fast_rcp(det);
#ifdef MYR_RICC_PROBE
        unsigned long long tq; { double t_ = rdet; asm volatile("" : "+v"(t_)); tq = __builtin_amdgcn_s_memtime(); }
#endif
        const double b0 = D2[3], b1 = D2[2];
        double kk0 = fma(q11, b0, -(q10 * b1)) * rdet;
        double kk1 = fma(q00, b1, -(q10 * b0)) * rdet;
        if (!(q00 > reg_floor) || !(det > reg_floor * q00)) {          // wave-uniform, rare
          const double u00 = q00, u10 = q10, u11 = q11;
          double d0 = u00;
          if (!(d0 > reg_floor)) { d0 = dmax(fabs(d0), reg_floor); ++nreg; }
          const double i0 = fast_rcp(d0);
          const double l10 = u10 * i0;
          double d1 = u11 - l10 * l10 * d0;
          if (!(d1 > reg_floor)) { d1 = dmax(fabs(d1), reg_floor); ++nreg; }
          if (nreg > 0 && abort_on_reg) return nreg;
          const double i1 = fast_rcp(d1);
          kk0 = b0; kk1 = b1;
          kk1 -= l10 * kk0;
          kk0 *= i0; kk1 *= i1;
          kk0 -= l10 * kk1;
        }
        MYR_TP(4)
        k_ptr[0] = kk0; k_ptr[k_str] = kk1;                            // (lanes without a gain write a scratch slot)
        k_ptr -= k_step;
        // (e) [P | pc] = [Qss | qc_s] - Qsq [K | kc]; rows 10, 11, 14, 15: Tnu -= qc_q[:, nu]^T kc
        const double A3 = fma(D2[3], f_a3m, D2[2] * f_a3e);
        const double B3 = g == 0 ? kk0 : (g == 1 ? kk1 : 0.0);     // (a select: groups 2, 3 may hold non-finite junk)
        MYR_TP(5)
        D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, B3, D2, 0, 0, 0);
        // midpoint part of stage k-1, first product: independent of the recursion, both products run behind D3
        // while the next stage's operands are prepared
        double m0 = nn0 + dv0, m1 = nn1 + dv1;
        const double ms0 = dpp_row_shr8(m0), ms1 = dpp_row_shr8(m1);
        mfma_d4 Cm;
        Cm[0] = fma(ms0, f_shm, m0 * f_keep);
        Cm[1] = fma(ms1, f_shm, m1 * f_keep);
        Cm[2] = 0.0; Cm[3] = 0.0;
        const mfma_d4 Rm = __builtin_amdgcn_mfma_f64_16x16x4f64(m0 * f_a1, nGm, Cm, 0, 0, 0);
        // midpoint part of stage k-1, second product (selector row: the du_m rows take Rm's row du)
        mfma_d4 Cq;
        Cq[0] = 0.0; Cq[1] = 0.0; Cq[2] = 0.0; Cq[3] = Rm[1];
        Qm = __builtin_amdgcn_mfma_f64_16x16x4f64(nGm, Rm[0], Cq, 0, 0, 0);
#ifdef MYR_RICC_PROBE
        asm volatile("" : "+v"(D3));
        tp[6] = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && lane == 0 && k >= N / 2 - 2 && k <= N / 2 + 1 && delta == 0.0)
          printf("stage %d: top->D1 %llu  D1->D2 %llu  D2+mid->ldl %llu  [wait D2+readlane %llu  det+rcp %llu  gains %llu]  gains->D3 %llu  D3 issue %llu | total %llu\n", k,
                 tp[1] - tp[0], tp[2] - tp[1], tp[3] - tp[2], tp[7] - tp[3], tq - tp[7], tp[4] - tq, tp[5] - tp[4], tp[6] - tp[5], tp[6] - tp[0]);
#endif
      }
    }
    X0 = D3[0]; X1 = D3[1];                                          // (X1 is read below in group 0 only)
    const double T1 = D3[1], T2 = D3[2], T3 = D3[3];                 // read in groups 2, 3 only
    // hand P, pc, Tnu to the first-point step through LDS (layouts of riccati())
    if (scol >= 0 && j != 5) {
      if (rowx) c.sP[g * NW + scol] = X0;
      if (g == 0) c.sP[NS * NW + scol] = X1;
    }
    if (rcc >= 0) {
      if (rowx) c.sPc[g * NC + rcc] = X0;
      if (g == 0) c.sPc[NS * NC + rcc] = X1;
      if (g >= 2 && g - 2 < NS) c.sTnu[(g - 2) * NC + rcc] = T2;
      if (g >= 2 && g < NS) c.sTnu[g * NC + rcc] = T3;
    }
    wsync();
    if (g == 2 && rcc >= 2) c.sTnu[(rcc - 2) * NC + 0] += T1;       // row 6: ge^T pc'[:, nu_i], summed over the stages
    wsync();
    return riccati_first_point(c, o, delta, nreg);
  }

  // The same sweep for the trapezoidal scheme: y = (dx_s, du_s, du_e), ONE eliminated control per stage, no midpoint part --
  // three MFMA per stage (R~, [Q|qc], the rank-1 update), the single pivot Q[du_e][du_e] read from lane 8 of register 2.
  // Slots as above (12, 13 unused); the end point of stage k is point k+1.
  __device__ static MYR_RICCATI_INLINE int riccati_mfma_trap(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg) {
    using namespace detail;
    static_assert(NQ == 1, "one control");
    const int lane = c.lane, N = c.N;
    const int g = lane >> 4, j = lane & 15;
    const int scol = j < 4 ? (j < NS ? j : -1) : (j < 6 ? NS : -1);
    const int ycol = scol >= 0 ? scol : ((j == 8 || j == 9) ? NW : -1);
    const int cc = j == 6 ? 0 : (j == 7 ? 1 : (j == 10 ? 2 : (j == 11 ? 3 : (j == 14 ? 4 : (j == 15 ? 5 : -1)))));
    const int rcc = (cc >= 0 && cc < NC) ? cc : -1;
    const bool rowx = g < NS;
    const double* he = c.hr + (long)N * HR_N;
    const double* st = c.st + (long)(N - 1) * SG_N;
    auto hsel = [&](int row, bool on) -> const double* {
      if (!on) return c.zr;
      if (scol >= 0) return he + HR_H + scol * NW + row;
      if (rcc == 0) return he + HR_G0 + row;
      if (rcc == 1) return he + HR_G1 + row;
      return c.zr;
    };
    const double* ptr[3] = {hsel(g, rowx), hsel(NS, g < 2),
                            !rowx ? c.zr : (ycol >= 0 ? st + SG_GE + g * NY1 + ycol : (rcc == 0 ? st + SG_GE + g * NY1 + NY : c.zr))};
    long stp[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) stp[q] = (ptr[q] == c.zr) ? 0 : (q == 2 ? (long)SG_N : (long)HR_N);
    const bool pinr = rowx && c.term_pinned[rowx ? g : 0];
    const double X0i = (pinr && scol == g) ? o.rho_term - delta : ((pinr && rcc == 2 + g) ? 1.0 : 0.0);
    const double dv0 = (rowx && scol == g) ? delta : 0.0, dv1 = (g < 2 && scol == NS) ? delta : 0.0;
    const double f_a1 = j < 6 ? 1.0 : 0.0, f_keep = rcc >= 0 ? 1.0 : 0.0, f_she = (j == 8 || j == 9) ? 1.0 : 0.0;
    const double f_x1 = g < 2 ? 1.0 : 0.0, f_t1 = g == 2 ? 1.0 : 0.0, f_t23 = g >= 2 ? 1.0 : 0.0;
    const double f_a3 = (g == 0 && (j < 6 || j == 10 || j == 11 || j == 14 || j == 15)) ? -1.0 : 0.0;
    const int k_off = (g == 0 && scol >= 0 && j != 5) ? scol : ((g == 0 && rcc >= 0) ? NQ * NW + rcc : -1);
    double* k_ptr = k_off >= 0 ? c.kg + (long)(N - 1) * KSTR + k_off : c.zr + ZR;
    const long k_step = k_off >= 0 ? KSTR : 0;
    double reg_floor = o.reg_floor;
    asm volatile("" : "+v"(reg_floor));
    int nreg = 0;
    constexpr int PF = MYR_RICCATI_PF;
    double in[PF][3];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
    }
    mfma_d4 D3 = {X0i, 0.0, 0.0, 0.0};
    for (int kb = N - 1; kb >= 0; kb -= PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int k = kb - u;
        if (k < 0) break;
        const double X0 = D3[0] + (in[u][0] + dv0), X1 = fma(D3[1], f_x1, in[u][1] + dv1);
        const double G = in[u][2];
#pragma unroll
        for (int q = 0; q < 3; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
        const double sh0 = dpp_row_shr4(X0), sh1 = dpp_row_shr4(X1);
        mfma_d4 C1;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        C1[2] = 0.0; C1[3] = 0.0;
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
        mfma_d4 C2;
        C2[0] = 0.0; C2[1] = D3[1] * f_t1; C2[2] = fma(D3[2], f_t23, D1[1]); C2[3] = D3[3] * f_t23;
        const mfma_d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(G, D1[0], C2, 0, 0, 0);
        const double q11 = rdlane(D2[2], 8);
        double d = q11;
        if (!(d > reg_floor)) {                                        // wave-uniform, rare (same pivot rule as chol_reg)
          d = dmax(fabs(d), reg_floor); ++nreg;
          if (abort_on_reg) return nreg;
        }
        const double kk = D2[2] * fast_rcp(d);
        k_ptr[0] = kk;
        k_ptr -= k_step;
        const double A3 = D2[2] * f_a3;
        const double B3 = g == 0 ? kk : 0.0;
        D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, B3, D2, 0, 0, 0);
      }
    }
    const double X0 = D3[0], X1 = D3[1], T1 = D3[1], T2 = D3[2], T3 = D3[3];
    if (scol >= 0 && j != 5) {
      if (rowx) c.sP[g * NW + scol] = X0;
      if (g == 0) c.sP[NS * NW + scol] = X1;
    }
    if (rcc >= 0) {
      if (rowx) c.sPc[g * NC + rcc] = X0;
      if (g == 0) c.sPc[NS * NC + rcc] = X1;
      if (g >= 2 && g - 2 < NS) c.sTnu[(g - 2) * NC + rcc] = T2;
      if (g >= 2 && g < NS) c.sTnu[g * NC + rcc] = T3;
    }
    wsync();
    if (g == 2 && rcc >= 2) c.sTnu[(rcc - 2) * NC + 0] += T1;
    wsync();
    return riccati_first_point(c, o, delta, nreg);
  }

  // ---- phase 8a: lanes over intervals -- closed-loop stage maps  s_{k+1} = Phi_k s_k + phi_k, s = (dx, du) of a knot
  // (the gains applied to the elimination rows, for the multipliers theta = (1, mu, nu)); rows -> LDS region R0
  // closed-loop map of stage k as an affine map in registers: A (NW x NW, row-major), b (NW)
  __device__ static inline void stage_phi(const Ctx& c, const double* th, int k, double* A, double* b) {
    const int N = c.N;
    const double* Kst = c.kg + (long)k * KSTR;
    const double* st = c.st + (long)k * SG_N;
    double Kk[NQ * NW], kq[NQ];
#pragma unroll
    for (int q = 0; q < NQ * NW; ++q) Kk[q] = Kst[q];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      double v = 0.0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v += Kst[NQ * NW + t * NC + cc] * th[cc];
      kq[t] = v;
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      double g[NY1];
#pragma unroll
      for (int q = 0; q <= NY; ++q) g[q] = st[SG_GE + i * NY1 + q];
      const bool pin = (k == N - 1) && c.term_pinned[i];
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        double v = g[q];
#pragma unroll
        for (int t = 0; t < NQ; ++t) v -= g[NW + t] * Kk[t * NW + q];
        A[i * NW + q] = pin ? 0.0 : v;
      }
      double v = g[NY];
#pragma unroll
      for (int t = 0; t < NQ; ++t) v -= g[NW + t] * kq[t];
      b[i] = pin ? 0.0 : v;
    }
#pragma unroll
    for (int a = 0; a < NU; ++a) {
#pragma unroll
      for (int q = 0; q < NW; ++q) A[(NS + a) * NW + q] = -Kk[(QE + a) * NW + q];
      b[NS + a] = -kq[QE + a];
    }
  }
  // Staged through LDS for the closed-form systems (computing the maps inside the scan phase costs them more in registers
  // than the LDS round trip: 195.6 k against 200.0 k solves/s on the bench); network systems compute them in the scan phase
  // and give the 24 KB back (PHI_IN_LDS: four wavefronts per workgroup instead of three).
#if defined(MYR_RECUR_SEQ) || defined(MYR_RECUR_SEQ_FWD)
  static constexpr bool PHI_IN_LDS = true;       // (the sequential form of the recursion reads the maps row by row from LDS)
#else
  static constexpr bool PHI_IN_LDS = !MLP;
#endif
  __device__ static void intervals_phi(Ctx& c, const double* th) {
    const int N = c.N;
    for (int k = c.lane; k < N; k += 64) {
      double A[NW * NW], b[NW];
      stage_phi(c, th, k, A, b);
      double* P = c.r0 + (long)k * PHI;
#pragma unroll
      for (int i = 0; i < NW; ++i) {
#pragma unroll
        for (int q = 0; q < NW; ++q) P[i * (NW + 1) + q] = A[i * NW + q];
        P[i * (NW + 1) + NW] = b[i];
      }
    }
  }

  // ---- phase 8b: forward recursion (sequential): lane r < NW owns row r of Phi|phi (prefetched one stage ahead),
  // s travels between lanes with v_readlane; s_k -> LDS for phase 9.  No barrier inside the loop.
#if !defined(MYR_RECUR_SEQ) && !defined(MYR_RECUR_SEQ_FWD)
  // Wave-scan form (see adjoint_recur): prefix scan of the closed-loop maps, lane k ends with s_{k+1}.
  __device__ static void forward_recur(Ctx& c, const double* th) {
    const int lane = c.lane, N = c.N;
    double s0[NW];                           // state in front of the block (uniform)
#pragma unroll
    for (int q = 0; q < NS; ++q) s0[q] = 0.0;
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      double v = 0.0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v -= c.sKu[a * NC + cc] * th[cc];
      s0[NS + a] = v;
    }
    if (lane < NW) {
      double v = 0.0;
#pragma unroll
      for (int q = 0; q < NW; ++q) v = (q == lane) ? s0[q] : v;
      c.sS[lane] = v;
    }
    for (int base = 0; base < N; base += 64) {
      const int k = base + lane;
      const bool on = k < N;
      double A[NW * NW], b[NW];
      if constexpr (PHI_IN_LDS) {
        const double* P = c.r0 + (long)(on ? k : 0) * PHI;
#pragma unroll
        for (int r = 0; r < NW; ++r) {
#pragma unroll
          for (int q = 0; q < NW; ++q) A[r * NW + q] = on ? P[r * (NW + 1) + q] : ((r == q) ? 1.0 : 0.0);
          b[r] = on ? P[r * (NW + 1) + NW] : 0.0;
        }
      } else {
#pragma unroll
        for (int q = 0; q < NW * NW; ++q) A[q] = ((q / NW) == (q % NW)) ? 1.0 : 0.0;      // identity beyond the last stage
#pragma unroll
        for (int q = 0; q < NW; ++q) b[q] = 0.0;
        if (on) stage_phi(c, th, k, A, b);
      }
      affine_prefix_scan_dpp<NW>(A, b);
      double sn[NW];
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        double v = b[r];
#pragma unroll
        for (int q = 0; q < NW; ++q) v += A[r * NW + q] * s0[q];
        sn[r] = v;
      }
      if (on) {
#pragma unroll
        for (int q = 0; q < NW; ++q) c.sS[(k + 1) * NW + q] = sn[q];
      }
#pragma unroll
      for (int q = 0; q < NW; ++q) s0[q] = __shfl(sn[q], 63, 64);
    }
  }
#else
  __device__ static void forward_recur(Ctx& c, const double* th) {
    // every row of 16 lanes repeats the computation of lanes 0..NW-1 (the broadcast below is per row); row 0 stores
    const int lane = c.lane, N = c.N, l16 = lane & 15, r = l16 < NW ? l16 : 0;
    constexpr int UB = 4;                          // stages per batch of LDS reads
    double v = 0.0;
    if (l16 >= NS && l16 < NW) {
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v -= c.sKu[(l16 - NS) * NC + cc] * th[cc];
    }
    double sq[NW];
    RowBcast<NW>::all(v, sq);
    if (lane < NW) c.sS[lane] = v;
    for (int kb = 0; kb < N; kb += UB) {
      double row[UB][NW + 1];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = kb + u < N ? kb + u : N - 1;
        const double* P = c.r0 + (long)k * PHI + r * (NW + 1);
#pragma unroll
        for (int q = 0; q <= NW; ++q) row[u][q] = P[q];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = kb + u;
        if (k >= N) break;
        v = row[u][NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) v += row[u][q] * sq[q];
        RowBcast<NW>::all(v, sq);
        if (lane < NW) c.sS[(k + 1) * NW + lane] = v;
      }
    }
  }

#endif

  // ---- phase 9: lanes over intervals -- step for midpoint / end point variables ---------------------------------
  __device__ static void intervals_dz(Ctx& c, const double* th) {
    const int N = c.N;
    if constexpr (TRAP) {       // every point is a knot: dz of point j is the recursion's s_j = (dx_j, du_j)
      (void)th;
      for (int e = c.lane; e < (N + 1) * NW; e += 64) {
        const int j = e / NW, q = e - j * NW;
        c.dz[zi(c, j, q)] = (j == 0 && q < NS) ? 0.0 : c.sS[e];
      }
      return;
    }
    if (c.lane < NW) c.dz[zi(c, 0, c.lane)] = c.lane < NS ? 0.0 : c.sS[c.lane];
    for (int k = c.lane; k < N; k += 64) {
      const double* st = c.st + (long)k * SG_N;
      const double* Kst = c.kg + (long)k * KSTR;
      double y[NY];
#pragma unroll
      for (int q = 0; q < NW; ++q) y[q] = c.sS[(long)k * NW + q];
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) v -= Kst[t * NW + q] * y[q];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) v -= Kst[NQ * NW + t * NC + cc] * th[cc];
        y[NW + t] = v;
      }
      double vmr[NS];                          // (all loads before the first store, see points_hess)
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double vm = st[SG_GM + r * NY1 + NY];
#pragma unroll
        for (int q = 0; q < NY; ++q) vm += st[SG_GM + r * NY1 + q] * y[q];
        vmr[r] = vm;
      }
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        c.dz[zi(c, 2 * k + 1, r)] = vmr[r];
        c.dz[zi(c, 2 * k + 2, r)] = c.sS[(long)(k + 1) * NW + r];     // = Ge y + ge (0 on a pinned terminal state)
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        c.dz[zi(c, 2 * k + 1, NS + a)] = y[NW + a];
        c.dz[zi(c, 2 * k + 2, NS + a)] = y[NW + NU + a];
      }
    }
  }

  // ---- phase 10: lanes over points -- step limits and merit slope ------------------------------------------------
  __device__ static void points_limits(Ctx& c, const HsSolveOpts& o, double mu, typename S::FwdOut& fo) {
    const double tau = detail::dmax(o.tau_min, 1.0 - mu);
    typename S::FwdOut l; l.alpha_p = 1.0; l.alpha_d = 1.0; l.gphi = 0.0;
    for (int j = c.lane; j < c.K; j += 64) {
      const double wj = wq(c.K, j, c.h);
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const long i = zi(c, j, q);
        S::step_limits(c.z[i], c.lb[i], c.ub[i], c.zL[i], c.zU[i], c.dz[i], mu, wj * c.pt[(PF_GW + q) * c.K + j], tau, l);
      }
    }
    fo.alpha_p = wv_min(l.alpha_p); fo.alpha_d = wv_min(l.alpha_d); fo.gphi = wv_sum(l.gphi);
  }

  // ---- merit trial at z + alpha dz --------------------------------------------------------------------------------
  __device__ static bool trial(Ctx& c, double alpha, double mu, double& f, double& bar, double& c1) {
    const int N = c.N, K = c.K;
    double* sX = c.r0; double* sF = c.r0 + (long)K * NS;
    node_pass<0>(c, alpha);                 // network systems: f of every trial point by the matrix-core pass -> sF
    double fa = 0, ba = 0; int bad = 0;
    for (int j = c.lane; j < K; j += 64) {
      double x[NS], u[NU], ff[NS];
      // one log per point instead of 2 NW (fp64 log is a long software sequence and dominated the trial): the slack
      // pairs (z-l)(u-z) are multiplied as mantissas, their binary exponents summed, so no product can under- or overflow
      double slk = 1.0; int sexp = 0;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const long i = zi(c, j, q);
        const double v = c.z[i] + alpha * c.dz[i];
        const double l = c.lb[i], ub = c.ub[i];
        const bool fr = l < ub;
        const bool hl = fr && (l > -INFINITY), hu = fr && (ub < INFINITY);
        const double sl = hl ? v - l : 1.0, su = hu ? ub - v : 1.0;
        bad += (sl > 0.0 ? 0 : 1) + (su > 0.0 ? 0 : 1);
        { int e_; slk *= frexp((sl > 0.0 ? sl : 1.0) * (su > 0.0 ? su : 1.0), &e_); sexp += e_; }
        if (q < NS) x[q] = v; else u[q - NS] = v;
      }
      ba -= log(slk) + sexp * 0.6931471805599453;
      if constexpr (!MLP) Sys::f(x, u, c.pp.get(), ff);
      set_time<Sys>(c.pp.get(), tq(j, c.h));
      double gj = Sys::g(x, u, c.pp.get());
      if (TRAP && j == K - 1) fold_terminal<Sys>(x, u, c.pp.get(), wq(K, j, c.h), gj, nullptr);
      fa += wq(K, j, c.h) * gj;
#pragma unroll
      for (int q = 0; q < NS; ++q) { sX[j * NS + q] = x[q]; if constexpr (!MLP) sF[j * NS + q] = ff[q]; }
    }
    wsync();
    double ca = 0;
    for (int k = c.lane; k < N; k += 64) {
      if constexpr (TRAP) {
#pragma unroll
        for (int q = 0; q < NS; ++q)
          ca += fabs(0.5 * c.h * (sF[k * NS + q] + sF[(k + 1) * NS + q]) - (sX[(k + 1) * NS + q] - sX[k * NS + q]));
        continue;
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const double xs = sX[(2 * k) * NS + q], xm = sX[(2 * k + 1) * NS + q], xe = sX[(2 * k + 2) * NS + q];
        const double fs = sF[(2 * k) * NS + q], fm = sF[(2 * k + 1) * NS + q], fe = sF[(2 * k + 2) * NS + q];
        ca += fabs((xe - xs) - c.h6 * (fs + 4.0 * fm + fe));
        ca += fabs(xm - 0.5 * (xs + xe) - c.h8 * (fs - fe));
      }
    }
    wsync();
    f = wv_sum(fa); bar = mu * wv_sum(ba); c1 = wv_sum(ca);
    bad = wv_isum(bad);
    if (bad != 0) return false;
    if (!detail::finite_(f)) return false;
    if (!detail::finite_(c1)) return false;
    return detail::finite_(bar);
  }

  __device__ static void update(Ctx& c, double ap, double ad, double mu, double ksig) {
    const double iks = 1.0 / ksig;
    for (int i = c.lane; i < c.n; i += 64) {
      const double l = c.lb[i], u = c.ub[i], zv = c.z[i], d = c.dz[i], zl = c.zL[i], zu = c.zU[i];
      const bool fr = l < u;
      const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
      const double zn = fr ? zv + ap * d : zv;
      const double sl = hl ? zv - l : 1.0, su = hu ? u - zv : 1.0;
      const double snl = hl ? zn - l : 1.0, snu = hu ? u - zn : 1.0;
      double vl = zl + ad * (-zl + (mu - zl * d) * detail::rcp_(sl));
      double vu = zu + ad * (-zu + (mu + zu * d) * detail::rcp_(su));
      const double ml = mu * detail::rcp_(snl), mu_ = mu * detail::rcp_(snu);  // the safeguard band is [m / ksig, m ksig]
      vl = detail::dmax(detail::dmin(vl, ksig * ml), ml * iks);
      vu = detail::dmax(detail::dmin(vu, ksig * mu_), mu_ * iks);
      c.z[i] = zn; c.zL[i] = hl ? vl : 0.0; c.zU[i] = hu ? vu : 0.0;
    }
  }

  __device__ static void init(Ctx& c) {
    const double k1 = 1e-2, k2 = 1e-2;
    for (int i = c.lane; i < c.n; i += 64) {
      const double l = c.lb[i], u = c.ub[i], v0 = c.z[i];
      const bool fr = l < u;
      const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
      const double width = (hl && hu) ? (u - l) : INFINITY;
      const double pl = detail::dmin(k1 * detail::dmax(1.0, fabs(l)), k2 * width);
      const double pu = detail::dmin(k1 * detail::dmax(1.0, fabs(u)), k2 * width);
      double v = v0;
      v = hl ? detail::dmax(v, l + pl) : v;
      v = hu ? detail::dmin(v, u - pu) : v;
      v = fr ? v : l;
      c.z[i] = v; c.zL[i] = hl ? 1.0 : 0.0; c.zU[i] = hu ? 1.0 : 0.0;
    }
    if (c.lane < ZR) c.zr[c.lane] = 0.0;
  }

  // ---- the solve (control flow identical to HsSolver<Sys>::solve) ------------------------------------------------
  __device__ static void solve(Ctx& c, const HsSolveOpts& o, HsSolveResult& res) {
    using namespace detail;
    init(c);
    wsync();
#pragma unroll
    for (int q = 0; q < NS; ++q) { const long i = zi(c, c.K - 1, q); c.term_pinned[q] = !(c.lb[i] < c.ub[i]); }
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
    Step pending{false, 0.0, 0.0, 0.0, o.kappa_sigma};
    for (int it = 0; it <= o.max_iter; ++it) {
      P1 p1;
      points_lin(c, p1, pending);
      pending.on = false;
      wsync();
      MYR_PH(0)
      double c1, cinf, lam_inf, sum_mult, stat_raw;
      intervals_elim(c, c1, cinf);
      wsync();
      MYR_PH(1)
      adjoint_recur(c, nuT);
      wsync();
      MYR_PH(2)
      intervals_lambda(c, lam_inf, sum_mult);
      wsync();
      MYR_PH(3)
      points_hess(c, stat_raw);
      wsync();
      MYR_PH(4)
      // inertia correction with retries (only the delta-dependent phases are redone)
      double delta = lm;
      if (o.delta_warm && delta_last > o.delta_warm_min) delta = dmax(delta, delta_last / DELTA_WARM_DIV);
      int nreg = 0;
      for (int tr_ = 0; tr_ < 12; ++tr_) {
        bool abort_on_reg = (tr_ < 11) && !(delta > 1e8);
#if defined(MYR_RICCATI_VALU) || defined(MYR_RICCATI_CHECK)
        intervals_qm(c, delta);
#else
        if constexpr (!MFMA_RICCATI && !TRAP) intervals_qm(c, delta);   // the matrix-core sweep forms the midpoint terms itself
#endif
        wsync();
        MYR_PH(5)
#ifdef MYR_RICCATI_CHECK   // dev self-check: both forms on the same inputs, differences of every output printed
        if constexpr (MFMA_RICCATI) {
          const int nv = riccati(c, o, delta, abort_on_reg);
          wsync();
          const int nk = c.N * KST, nl = NW * NW + NW * NC + NS * NY1 + NS * NC + NU * NC;
          for (int i = c.lane; i < nk; i += 64) c.r0[i] = c.kg[i];
          for (int i = c.lane; i < nl; i += 64) c.r0[nk + i] = c.sP[i];
          wsync();
          for (int i = c.lane; i < nk; i += 64) c.kg[i] = -7.0;
          wsync();
          const int nm = riccati_mfma(c, o, delta, false);
          wsync();
          double dk = 0, dP = 0, dPc = 0, dT = 0, dKu = 0, mk = 0;
          int worst = -1;
          for (int i = c.lane; i < nk; i += 64) { const double d = fabs(c.r0[i] - c.kg[i]); if (d > dk) { dk = d; worst = i; } mk = dmax(mk, fabs(c.r0[i])); }
          for (int i = c.lane; i < nl; i += 64) {
            const double d = fabs(c.r0[nk + i] - c.sP[i]);
            if (i < NW * NW) dP = dmax(dP, d);
            else if (i < NW * NW + NW * NC) dPc = dmax(dPc, d);
            else if (i < NW * NW + NW * NC + NS * NY1) ;
            else if (i < NW * NW + NW * NC + NS * NY1 + NS * NC) dT = dmax(dT, d);
            else dKu = dmax(dKu, d);
          }
          const double dkm = wv_max(dk);
          const int wl = (dk == dkm) ? worst : -1;
          const int wmax = -wv_isum(0) + (int)wv_max((double)wl);
          dP = wv_max(dP); dPc = wv_max(dPc); dT = wv_max(dT); dKu = wv_max(dKu); mk = wv_max(mk);
          if (c.lane == 0 && blockIdx.x == 0)
            printf("ricc check it %d delta %.3g: nreg valu %d mfma %d | max|dK| %.3e (|K| %.3e, worst elem %d = stage %d field %d) dP %.3e dPc %.3e dTnu %.3e dKu %.3e\n",
                   it, delta, nv, nm, dkm, mk, wmax, wmax / KST, wmax % KST, dP, dPc, dT, dKu);
          nreg = nm;
        } else nreg = riccati(c, o, delta, abort_on_reg);
#else
#ifndef MYR_RICCATI_VALU
        if constexpr (MFMA_RICCATI && TRAP) nreg = riccati_mfma_trap(c, o, delta, abort_on_reg);
        else if constexpr (MFMA_RICCATI) nreg = riccati_mfma(c, o, delta, abort_on_reg);
        else
#endif
          nreg = riccati(c, o, delta, abort_on_reg);
#endif
        wsync();
        MYR_PH(6)
        if (nreg == 0) break;
        if (!abort_on_reg) break;
        if (delta == 0.0) delta = (delta_last > 0.0) ? dmax(1e-8, delta_last / 3.0) : 1e-4;
        else delta *= (delta_last > 0.0) ? 8.0 : 100.0;
      }
      delta_last = (delta > lm) ? delta : 0.0;
      const int nm = MLAM * c.N * NS + p1.nm;
      const double sd = nm > 0 ? dmax(1.0, (sum_mult + p1.sm) / nm / 100.0) : 1.0;
      const double stat = stat_raw / sd, comp = p1.cmax / sd;
      res.cost = p1.f; res.feas = cinf; res.stat = stat; res.compl_ = comp;
      if (!(finite_(p1.f) && finite_(cinf) && finite_(stat_raw))) { res.status = 2; res.iters = it; return; }
      if (cinf <= o.tol_feas && stat <= o.tol_stat && comp <= o.tol_compl) { res.status = 0; res.iters = it; return; }
      if (it == o.max_iter) break;
      for (int guard = 0; guard < 8; ++guard) {
        const double cerr = (p1.cmin <= p1.cmax) ? dmax(fabs(p1.cmax - mu), fabs(p1.cmin - mu)) : 0.0;
        const double emu = dmax(dmax(stat, cinf), cerr / sd);
        if (emu <= o.kappa_eps * mu && mu > mu_min) {
          const double nmu = dmax(mu_min, dmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
          mu = nmu;
        } else break;
      }
      // terminal multipliers (every lane, identical)
      typename S::SweepOut so;
#pragma unroll
      for (int i = 0; i < NS * NC; ++i) so.Tnu[i] = c.sTnu[i];
#pragma unroll
      for (int i = 0; i < NS; ++i) so.term_pinned[i] = c.term_pinned[i];
      double nu[NS];
      S::solve_nu(so, mu, nu);
      double th[NC];
      th[0] = 1.0; th[1] = mu;
#pragma unroll
      for (int i = 0; i < NS; ++i) th[2 + i] = nu[i];
      MYR_PH(7)
      if constexpr (PHI_IN_LDS) {
        intervals_phi(c, th);
        wsync();
      }
      MYR_PH(13)
      forward_recur(c, th);
      wsync();
      MYR_PH(8)
      intervals_dz(c, th);
      wsync();
      MYR_PH(9)
      typename S::FwdOut fo;
      points_limits(c, o, mu, fo);
      MYR_PH(10)
      if (!(finite_(fo.gphi) && finite_(fo.alpha_p))) { res.status = 2; res.iters = it; return; }
      if (c1 > 0.0) {
        const double need = fo.gphi / (0.9 * c1);
        if (pen < need) pen = need + 1.0;
        if (PEN_RELAX > 0) {         // penalty relaxation (see hs_solver.h)
          const double want = 2.0 * dmax(need, 0.0) + 1.0;
          pen_over = (pen > PEN_RELAX_RATIO * want) ? pen_over + 1 : 0;
          if (pen_over >= PEN_RELAX && pen_cuts < PEN_RELAX_MAX) { pen = want; pen_over = 0; ++pen_cuts; }
        }
      }
      const double Dphi = fo.gphi - pen * c1;
      // merit value at the current point: objective, barrier sum and l1 defect norm are by-products of the
      // linearisation phases (same formulas as trial()), so no trial at alpha = 0 is spent on it
      const double f0 = p1.f, bar0 = mu * p1.lg, c10 = c1;
      const double phi0 = f0 + bar0 + pen * c10;
      // non-monotone Armijo reference (see hs_solver.h)
      if (mu != hist_mu || pen != hist_pen) { nhist = 0; hpos = 0; hist_mu = mu; hist_pen = pen; }
      double phiref = phi0;
      for (int j = 0; j < nhist; ++j) phiref = dmax(phiref, hist[j]);
      if (o.nonmono > 0) { hist[hpos % o.nonmono] = phi0; ++hpos; if (nhist < o.nonmono) ++nhist; }
      double a = fo.alpha_p;
      bool ok = false;
      for (int ls = 0; ls < 40; ++ls) {
        double ft, bt, ct;
        if (trial(c, a, mu, ft, bt, ct)) {
          const double phit = ft + bt + pen * ct;
          if (phit <= phiref + 1e-8 * a * Dphi + 1e-13 * fabs(phi0)) { ok = true; break; }
        }
        a *= 0.5;
      }
      if (!ok) {
        if (++stall > 5) { res.status = 3; res.iters = it; return; }
      } else stall = 0;
      MYR_PH(11)
      pending.on = true; pending.ap = a; pending.ad = o.dual_follow ? fo.alpha_d * (a / fo.alpha_p) : fo.alpha_d; pending.mu = mu;
      MYR_PH(12)
#pragma unroll
      for (int i = 0; i < NS; ++i) nuT[i] += a * (nu[i] - nuT[i]);
      if (o.lm_init > 0.0) {     // step-quality feedback -> Levenberg-Marquardt damping (see hs_solver.h)
        const double ratio = o.lm_abs ? a : a / fo.alpha_p;   // step actually taken, relative to the full Newton step
        if (ratio <= 0.25) lm = dmin(1e2, dmax(o.lm_init, 4.0 * lm));
        else if (ratio >= 0.99) { lm *= 0.25; if (lm < 0.1 * o.lm_init) lm = 0.0; }
      }
      if (o.recenter > 0) {
        small_steps = (a < o.recenter_alpha) ? small_steps + 1 : 0;
        if (small_steps >= o.recenter && mu < o.mu_init) { mu = dmin(o.mu_init, 10.0 * mu); small_steps = 0; }
      }
      wsync();
    }
    res.status = 1; res.iters = o.max_iter;
  }
};

#ifndef MYR_WAVE_MIN_WAVES
#define MYR_WAVE_MIN_WAVES 1   // waves per SIMD the register allocation must allow (occupancy)
#endif
// Persistent: grid = the wavefronts the device keeps resident (blockDim.x / 64 per workgroup: 1, or up to WPB_MAX for network
// systems, whose wavefronts share the weights in LDS; dynamic LDS = waves x lds_solver_doubles + weights); every wavefront
// pulls trajectories from `ticket` until the batch is done and owns ONE scratch block (slot blockIdx.x * waves + wave) that
// it re-uses for all of them.
template <class Sys, int SCHEME = 0>
__global__ __launch_bounds__((64 * HsWave<Sys, SCHEME>::WPB_MAX), MYR_WAVE_MIN_WAVES)
void hs_solve_wave_kernel(int B, int* ticket, HsSolveOpts o, VarScale vs, double* __restrict__ z, const double* __restrict__ lb,
                          const double* __restrict__ ub, double* lam, double* scratch, long scratch_stride,
                          const double* __restrict__ params, int params_stride, double* cost, int32_t* status,
                          int32_t* iters, double* kkt, int coop, unsigned long long poison) {
  using W = HsWave<Sys, SCHEME>;
  extern __shared__ __attribute__((aligned(16))) char smem_wave[];
  typename W::Ctx c;
  // coop (network systems, small batches): the workgroup's wavefronts share ONE trajectory -- wavefront 0 solves, the others
  // help with the network passes (node_pass); otherwise every wavefront of the workgroup is an independent solve
  // (wavefront index wave-uniform by construction: branches on it are scalar branches, not EXEC-masked regions -- hs_solver_fused.h, sweep())
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wave = coop ? 0 : wv, waves = coop ? 1 : (int)(blockDim.x >> 6);
  c.N = o.N; c.K = W::npoints(o.N); c.n = c.K * W::NW; c.lane = threadIdx.x & 63;
  c.h = o.h; c.h6 = o.h / 6.0; c.h8 = o.h / 8.0;
  double* s = scratch + ((long)blockIdx.x * waves + wave) * scratch_stride + W::PADF;
  c.zL = s; s += c.n; c.zU = s; s += c.n; c.dz = s; s += c.n;
  c.pt = s; s += (long)W::PF_N * c.K;
  c.hr = c.pt + (long)W::PF_F * c.K;                            // overlaid on the f | A fields (see PF_*)
  if (W::HR_N > W::NS + W::NS * W::NS) s += (long)(W::HR_N - W::NS - W::NS * W::NS) * c.K;
  c.st = s; s += (long)W::SG_N * c.N;
  if (W::OVERLAY_K) c.kg = c.st + W::SG_LD;                     // overlaid on the adjoint maps of the stage records
  else { c.kg = s; s += (long)W::KST * c.N; }
  c.zr = s; s += W::ZR + 2;
  double* const lam_own = s;
  double* l = reinterpret_cast<double*>(smem_wave) + (long)wave * W::lds_solver_doubles(c.N);
  c.r0 = l; l += W::r0_doubles(c.N);
  c.sPi = l; l += c.N * W::NS;
  c.sS = l; l += (c.N + 1) * W::NW;
  c.sP = l; l += W::NW * W::NW;
  c.sPc = l; l += W::NW * W::NC;
  c.sGe = l; l += W::NS * W::NY1;
  c.sTnu = l; l += W::NS * W::NC;
  c.sKu = l; l += W::NU * W::NC;
  c.wl = reinterpret_cast<double*>(smem_wave) + (long)waves * W::lds_solver_doubles(c.N);
  if (coop) { c.coop = blockDim.x >> 6; c.cmd = c.wl + NodeTraits<Sys>::lds_doubles; }
  if constexpr (W::MLP) {        // weights shared by the batch: wavefront 0 loads them once, the ONE workgroup barrier of the kernel
    if (params_stride == 0) {
      SysParams<Sys> pw;
      pw.load(params, 0, 0);
      if (wv == 0) NodeMfma64::load_weights(pw.get(), c.wl, c.lane);
      __syncthreads();
    }
    if (coop && wv != 0) {       // helper wavefronts of the cooperative mode
      W::coop_helper(c.wl, c.cmd, wv, c.lane);
      return;
    }
  }
  for (;;) {
    int t = 0;
    if (c.lane == 0) t = atomicAdd(ticket, 1);
    const long b = __builtin_amdgcn_readfirstlane(t);
    if (b >= B) break;
    c.z = z + b * (long)c.n; c.lb = lb + b * (long)c.n; c.ub = ub + b * (long)c.n;
    c.lam = lam ? lam + b * (long)(W::MLAM * c.N * W::NS) : lam_own;
    if (poison) {      // MYRIAD_POISON (tests/test_gpu_poison.py): what this wavefront inherits from its previous trajectory -- its LDS, its scratch slot
      const unsigned long long salt = (unsigned long long)b * 1315423911ULL + blockIdx.x * 8 + wave;
      double* l0 = reinterpret_cast<double*>(smem_wave) + (long)wave * W::lds_solver_doubles(c.N);
      for (int i = c.lane; i < W::lds_solver_doubles(c.N); i += 64) l0[i] = poison_value(poison, (unsigned long long)i, salt);
      double* s0 = scratch + ((long)blockIdx.x * waves + wave) * scratch_stride;
      for (long i = c.lane; i < scratch_stride; i += 64) s0[i] = poison_value(poison, (unsigned long long)i + (1ULL << 32), salt);
      W::wsync();
    }
    c.pp.load(params, b, params_stride);
    c.pp.set_scale(vs.s);
    if constexpr (W::MLP) {      // per-trajectory weights (launched with one wavefront per workgroup): loaded per trajectory
      if (params_stride != 0) {
        NodeMfma64::load_weights(c.pp.get(), c.wl, c.lane);
        W::wsync();
      }
    }
    HsSolveResult r;
#ifdef MYR_PHASE_TIMING
    for (int i = 0; i < 16; ++i) c.tph[i] = 0;
    c.t0 = clock64();
#endif
    W::solve(c, o, r);
#ifdef MYR_PHASE_TIMING
    if (c.lane == 0 && b < 4) {
      printf("traj %ld it %d: lin %lld elim %lld adj %lld lam %lld hess %lld qm %lld ricc %lld nu %lld fwd %lld dz %lld lim %lld ls %lld upd %lld\n",
             b, r.iters, c.tph[0], c.tph[1], c.tph[2], c.tph[3], c.tph[4], c.tph[5], c.tph[6], c.tph[7], c.tph[8], c.tph[9],
             c.tph[10], c.tph[11], c.tph[12]);
    }
#endif
    if (c.lane == 0) {
      if (cost) cost[b] = r.cost;
      if (status) status[b] = r.status;
      if (iters) iters[b] = r.iters;
      if (kkt) { kkt[3 * b] = r.feas; kkt[3 * b + 1] = r.stat; kkt[3 * b + 2] = r.compl_; }
    }
    W::wsync();      // the slot's scratch and LDS are handed to the next trajectory
  }
  if constexpr (W::MLP) {
    if (coop) {          // release the helper wavefronts
      if (c.lane == 0) reinterpret_cast<typename W::CoopCmd*>(c.cmd)->mode = -1;
      __syncthreads();
    }
  }
}

}  // namespace myriad
