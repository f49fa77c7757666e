// hs_eval.h -- Hermite-Simpson transcription evaluation kernel (K1 in DESIGN.md).
//
// Replaces, for a whole batch, the four jitted callbacks the reference hands to its NLP solver
// (/root/reference/myriad/nlp_solvers/__init__.py:32-40) for the transcription of
// /root/reference/myriad/trajectory_optimizers/collocation/hermite_simpson.py:
//   constraints (:325-335; hs_defect :110-128, hs_interpolation :153-170)
//   jacrev(constraints)  -> stage blocks, never the dense 800x1005 matrix
//   objective (:243-257; hs_cost :194-214) and grad(objective)
//
// Mapping (gfx950): one trajectory per workgroup of WPT wavefronts (WPT=1: one trajectory per wavefront).
//   phase 1  lanes over the K=2N+1 collocation points: coalesced loads of (x_j,u_j) from the
//            instance-major decision vector, closed-form f/A/B (systems_gen.h), record -> LDS
//   phase 2  lanes over OUTPUT elements: every c / J-block / grad element is one LDS gather + one FMA
//            through a 100-entry stencil table, stored with fully coalesced 16-byte stores.
// HBM traffic = algorithmic bytes: z is read once, every output is written once.
#pragma once
#include <hip/hip_runtime.h>
#include "systems_gen.h"
#include "node_mfma.h"

namespace myriad {

enum { EVAL_HS = 0, EVAL_TRAP = 1 };

template <class Sys, int SCHEME = EVAL_HS>
struct HsLayout {
  static constexpr int NS = Sys::NS, NU = Sys::NU, NW = Sys::NW;
  // per-point LDS record: x | f | A | B ; padded to an odd number of doubles (bank-conflict-free b64 access)
  static constexpr int OFF_X = 0, OFF_F = NS, OFF_A = 2 * NS, OFF_B = 2 * NS + NS * NS;
  static constexpr int REC_RAW = 2 * NS + NS * NS + NS * NU;
  static constexpr int REC = (REC_RAW % 2 == 0) ? REC_RAW + 1 : REC_RAW;
  // per-interval Jacobian stencil: Dxs,Dxm,Dxe,Dus,Dum,Due,Ixs,Ixe,Ius,Iue
  // trapezoidal: Cxs = h/2 A_s + I, Cxe = h/2 A_e - I (ns x ns), Cus = h/2 B_s, Cue = h/2 B_e (ns x nu)
  static constexpr int JPI = SCHEME == EVAL_HS ? 5 * NS * NS + 5 * NS * NU : 2 * NS * NS + 2 * NS * NU;
  static constexpr int PPI = SCHEME == EVAL_HS ? 2 : 1;      // points advanced per interval
  static constexpr int CPI = SCHEME == EVAL_HS ? 2 : 1;      // constraint blocks (of ns rows) per interval
  static constexpr int NGRAD_PER_PT = Sys::COST_DEP_X ? NW : NU;
};

struct HsStencil {   // out = coef * rec[2k*REC + off] + ident
  double coef;
  double ident;
  int off;
  int pad;
};

// Fill the per-interval stencil table (JPI entries) -- same for every interval and instance.
template <class Sys, int SCHEME>
__device__ inline void hs_build_stencil(HsStencil* tab, double h, int tid, int nthreads) {
  using L = HsLayout<Sys, SCHEME>;
  constexpr int NS = L::NS, NU = L::NU, REC = L::REC;
  const double h6 = h / 6.0, h8 = h / 8.0;
  if (SCHEME == EVAL_TRAP) {
    const double hh = 0.5 * h;
    for (int r = tid; r < L::JPI; r += nthreads) {
      int q = r, blk, row, col; bool isx;
      if (q < 2 * NS * NS) { blk = q / (NS * NS); q %= NS * NS; row = q / NS; col = q % NS; isx = true; }
      else { q -= 2 * NS * NS; blk = 2 + q / (NS * NU); q %= NS * NU; row = q / NU; col = q % NU; isx = false; }
      HsStencil s;
      s.coef = hh;
      s.ident = (blk == 0 && row == col) ? 1.0 : ((blk == 1 && row == col) ? -1.0 : 0.0);   // trapezoidal.py:161-163
      const int pt = (blk == 0 || blk == 2) ? 0 : 1;
      s.off = pt * REC + (isx ? (L::OFF_A + row * NS + col) : (L::OFF_B + row * NU + col));
      s.pad = 0;
      tab[r] = s;
    }
    return;
  }
  for (int r = tid; r < L::JPI; r += nthreads) {
    // decode block
    int q = r;
    int blk, row, col;
    bool isx;
    if (q < 3 * NS * NS) { blk = q / (NS * NS); q %= NS * NS; row = q / NS; col = q % NS; isx = true; }
    else if ((q -= 3 * NS * NS) < 3 * NS * NU) { blk = 3 + q / (NS * NU); q %= NS * NU; row = q / NU; col = q % NU; isx = false; }
    else if ((q -= 3 * NS * NU) < 2 * NS * NS) { blk = 6 + q / (NS * NS); q %= NS * NS; row = q / NS; col = q % NS; isx = true; }
    else { q -= 2 * NS * NS; blk = 8 + q / (NS * NU); q %= NS * NU; row = q / NU; col = q % NU; isx = false; }
    // point offset within the interval (0 = start knot, 1 = midpoint, 2 = end knot), coefficient, identity part
    int pt; double coef, ident = 0.0;
    switch (blk) {
      case 0: pt = 0; coef = -h6;       ident = (row == col) ? -1.0 : 0.0; break;   // Dxs = -I - h/6 A_s
      case 1: pt = 1; coef = -4.0 * h6; break;                                      // Dxm = -4h/6 A_m
      case 2: pt = 2; coef = -h6;       ident = (row == col) ? 1.0 : 0.0; break;    // Dxe =  I - h/6 A_e
      case 3: pt = 0; coef = -h6; break;                                            // Dus
      case 4: pt = 1; coef = -4.0 * h6; break;                                      // Dum
      case 5: pt = 2; coef = -h6; break;                                            // Due
      case 6: pt = 0; coef = -h8;       ident = (row == col) ? -0.5 : 0.0; break;   // Ixs = -I/2 - h/8 A_s
      case 7: pt = 2; coef = h8;        ident = (row == col) ? -0.5 : 0.0; break;   // Ixe = -I/2 + h/8 A_e
      case 8: pt = 0; coef = -h8; break;                                            // Ius
      default: pt = 2; coef = h8; break;                                            // Iue
    }
    HsStencil s;
    s.coef = coef; s.ident = ident;
    s.off = pt * REC + (isx ? (L::OFF_A + row * NS + col) : (L::OFF_B + row * NU + col));
    s.pad = 0;
    tab[r] = s;
  }
}

// grid.x = B (one trajectory per workgroup), block = 64*WPT threads.
// dynamic LDS: K*REC doubles + JPI stencil entries + reduction scratch.
template <class Sys, int WPT, bool NTS = false, int SCHEME = EVAL_HS>
__global__ __launch_bounds__(64 * WPT)
void hs_eval_kernel(int N, double h, const double* __restrict__ z, const double* __restrict__ params,
                    int params_stride, double* __restrict__ fout, double* __restrict__ gout,
                    double* __restrict__ cout, double* __restrict__ jout) {
  using L = HsLayout<Sys, SCHEME>;
  constexpr int NS = L::NS, NU = L::NU, NW = L::NW, REC = L::REC, JPI = L::JPI, PPI = L::PPI;
  constexpr int NT = 64 * WPT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int K = PPI * N + 1;
  const int n = K * NW;
  double* rec = reinterpret_cast<double*>(smem_raw);                       // K*REC
  HsStencil* tab = reinterpret_cast<HsStencil*>(rec + ((K * REC + 1) & ~1));  // JPI entries, 16-B aligned
  double* red = reinterpret_cast<double*>(tab + JPI);                      // WPT partial sums
  // network systems (Hermite-Simpson): f, A, B of all points by the matrix-core pass of node_mfma.h, 16 points per tile, the
  // tiles dealt over the workgroup's wavefronts; the weights (40 KB) sit in LDS behind the reduction scratch
  constexpr bool MLP = NodeTraits<Sys>::mlp && SCHEME == EVAL_HS;
  double* wl = red + ((WPT + 1) & ~1);

  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  const double* zb = z + b * (long)n;
  SysParams<Sys> pp;
  pp.load(params, b, params_stride);
  const double* p = pp.get();

  hs_build_stencil<Sys, SCHEME>(tab, h, tid, NT);
  if constexpr (MLP) {
    NodeMfma64::load_weights(p, wl, tid, NT);
    __syncthreads();
    NodeMfma64::Args a;
    a.z = (const nd_glb*)zb; a.dz = nullptr; a.lam = nullptr; a.pt = nullptr; a.sF = nullptr;
    a.alpha = 0.0; a.h6 = 0.0; a.h8 = 0.0; a.K = K; a.N = N; a.pf_f = a.pf_a = a.pf_b = a.pf_d2 = 0;
    a.t0 = tid >> 6; a.ts = WPT;
    a.rec = (nd_lds*)rec; a.use_rec = 1; a.rec_stride = REC; a.rec_f = L::OFF_F; a.rec_a = L::OFF_A; a.rec_b = L::OFF_B;
    NodeMfma64::pass<1>((const nd_lds*)wl, a, tid & 63);
  }

  // ---- phase 1: per-point dynamics, Jacobians, cost ----
  const double h6 = h / 6.0;
  double facc = 0.0;
  for (int j = tid; j < K; j += NT) {
    double x[NS], u[NU], f[NS], A[NS * NS], Bm[NS * NU], g, gw[NW];
#pragma unroll
    for (int i = 0; i < NS; ++i) x[i] = zb[j * NS + i];
#pragma unroll
    for (int i = 0; i < NU; ++i) u[i] = zb[K * NS + j * NU + i];
    set_time<Sys>(p, (SCHEME == EVAL_HS ? 0.5 * h : h) * j);      // t_j of linspace(0, T, K) (hermite_simpson.py:252, trapezoidal.py:124)
    double* r = rec + j * REC;
    if constexpr (MLP) {          // dynamics and Jacobians are already in the record; only the (closed-form) cost is per lane
      Sys::cost_grad(x, u, p, &g, gw);
#pragma unroll
      for (int i = 0; i < NS; ++i) r[L::OFF_X + i] = x[i];
      (void)f; (void)A; (void)Bm;
    } else {
      Sys::lin(x, u, p, f, A, Bm, &g, gw);
#pragma unroll
      for (int i = 0; i < NS; ++i) { r[L::OFF_X + i] = x[i]; r[L::OFF_F + i] = f[i]; }
#pragma unroll
      for (int i = 0; i < NS * NS; ++i) r[L::OFF_A + i] = A[i];
#pragma unroll
      for (int i = 0; i < NS * NU; ++i) r[L::OFF_B + i] = Bm[i];
    }
    // Simpson weight of point j in  sum_k h/6 (g_s + 4 g_m + g_e)   (hermite_simpson.py:212-214)
    const double w = SCHEME == EVAL_HS ? ((j & 1) ? 4.0 * h6 : ((j == 0 || j == K - 1) ? h6 : 2.0 * h6))
                                       : ((j == 0 || j == K - 1) ? 0.5 * h : h);      // trapezoidal.py:80-94
    if (SCHEME == EVAL_TRAP && j == K - 1) fold_terminal<Sys>(x, u, p, w, g, gw);   // trapezoidal.py:126-127 (not in the HS objective)
    facc += w * g;
    if (gout) {
      double* gb = gout + b * (long)(K * L::NGRAD_PER_PT);
      if (Sys::COST_DEP_X) {
#pragma unroll
        for (int i = 0; i < NS; ++i) gb[j * NS + i] = w * gw[i];
#pragma unroll
        for (int i = 0; i < NU; ++i) gb[K * NS + j * NU + i] = w * gw[NS + i];
      } else {
#pragma unroll
        for (int i = 0; i < NU; ++i) gb[j * NU + i] = w * gw[NS + i];
      }
    }
  }
  // objective: wave shuffle reduction, then across waves through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) facc += __shfl_down(facc, o, 64);
  if ((tid & 63) == 0) red[tid >> 6] = facc;
  __syncthreads();
  if (tid == 0 && fout) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < WPT; ++w) s += red[w];
    fout[b] = s;
  }

  // ---- phase 2a: constraints  c = [defects (interval-major, state-minor) ; interpolation residuals] ----
  if (cout) {
    double* cb = cout + b * (long)(L::CPI * N * NS);
    const double h8 = h / 8.0;
    const int half = N * NS;
    for (int e = tid; e < L::CPI * half; e += NT) {
      const int isint = e >= half;
      const int q = isint ? e - half : e;
      const int k = q / NS, i = q - k * NS;
      const double* rs = rec + (PPI * k) * REC;
      double val;
      if (SCHEME == EVAL_HS) {
        const double xs = rs[L::OFF_X + i], xm = rs[REC + L::OFF_X + i], xe = rs[2 * REC + L::OFF_X + i];
        const double fs = rs[L::OFF_F + i], fm = rs[REC + L::OFF_F + i], fe = rs[2 * REC + L::OFF_F + i];
        const double d = (xe - xs) - h6 * (fs + 4.0 * fm + fe);               // hermite_simpson.py:124-128
        const double it = xm - 0.5 * (xs + xe) - h8 * (fs - fe);              // hermite_simpson.py:167-170
        val = isint ? it : d;
      } else {
        const double xs = rs[L::OFF_X + i], xe = rs[REC + L::OFF_X + i];
        const double fs = rs[L::OFF_F + i], fe = rs[REC + L::OFF_F + i];
        val = 0.5 * h * (fs + fe) - (xe - xs);                                // trapezoidal.py:161-163
      }
      if (NTS) __builtin_nontemporal_store(val, &cb[e]); else cb[e] = val;
    }
  }

  // ---- phase 2b: Jacobian stage blocks; two consecutive elements per lane -> one 16-byte store ----
  if (jout) {
    if constexpr ((JPI & 1) == 0) {
      double2* jb = reinterpret_cast<double2*>(jout + b * (long)N * JPI);
      const int npairs = (N * JPI) >> 1;
      for (int e2 = tid; e2 < npairs; e2 += NT) {
        const int e = e2 << 1;
        const int k = e / JPI, r = e - k * JPI;                               // r even, r+1 < JPI
        const double* rk = rec + (PPI * k) * REC;
        const HsStencil s0 = tab[r], s1 = tab[r + 1];
        double2 v;
        v.x = fma(s0.coef, rk[s0.off], s0.ident);
        v.y = fma(s1.coef, rk[s1.off], s1.ident);
        if (NTS) {
          typedef double d2v __attribute__((ext_vector_type(2)));
          d2v vv; vv.x = v.x; vv.y = v.y;
          __builtin_nontemporal_store(vv, reinterpret_cast<d2v*>(&jb[e2]));
        } else jb[e2] = v;
      }
    } else {
      double* jb = jout + b * (long)N * JPI;
      for (int e = tid; e < N * JPI; e += NT) {
        const int k = e / JPI, r = e - k * JPI;
        const HsStencil s0 = tab[r];
        jb[e] = fma(s0.coef, rec[(PPI * k) * REC + s0.off], s0.ident);
      }
    }
  }
}

template <class Sys, int SCHEME = EVAL_HS>
inline size_t hs_eval_lds_bytes(int N, int wpt) {
  using L = HsLayout<Sys, SCHEME>;
  const int K = L::PPI * N + 1;
  const size_t weights = (NodeTraits<Sys>::mlp && SCHEME == EVAL_HS) ? (size_t)NodeTraits<Sys>::lds_doubles * 8 + 16 : 0;
  return (size_t)((K * L::REC + 1) & ~1) * 8 + (size_t)L::JPI * sizeof(HsStencil) + (size_t)wpt * 8 + 16 + weights;
}

}  // namespace myriad
