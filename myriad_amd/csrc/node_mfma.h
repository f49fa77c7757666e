// node_mfma.h -- the neural-ODE dynamics of BASELINE config 5 (x' = MLP([x; u]), 2 x 64 sigmoid layers) on fp64 matrix cores.
//
// What the reference does per collocation point through jax autodiff (/root/reference/myriad/systems/neural_ode/
// node_system.py:35-38, network of /root/reference/myriad/neural_ode/create_node.py:110-117) -- value, input Jacobian and the
// multiplier-contracted second derivative of the network -- is evaluated here for SIXTEEN points at a time by one
// wavefront with v_mfma_f64_16x16x4_f64, in the transposed ("features along rows") form
//     A1 = W1^T X + b1,  H1 = s(A1);   A2 = W2^T H1 + b2,  H2 = s(A2);   F = W3^T H2 + b3          (columns = points)
// because of the register layout of that instruction (probed: tools/dev/mfma/probe_f64.hip): a 16 x 16 result tile keeps
// element (row n, column i) in lane 16 (n % 4) + i, register n / 4, and the B operand of the next product wants element
// (k, i) in lane 16 (k % 4) + i for k-step k / 4 -- register s of a result tile IS k-step s of the next layer's B operand.
// A whole layer chains into the next one without moving a single value between lanes; the elementwise work (sigmoid, its
// derivatives, the tangent seeds) runs on all 64 lanes on 16 values each.  The weights are the A operands: they sit in LDS
// (40 KB, loaded once per workgroup, re-used for every trajectory of the persistent kernel) in layouts whose 64-lane
// reads are conflict-free, one ds_read_b64 per 1024-FMA instruction.  (The per-lane form this replaces pulled 4 804
// weights through scalar loads for every point and ran at <1 % of the fp64 rate.)
//
//   MODE 0  F                                   (merit-function trials)               88 MFMA per 16 points
//   MODE 1  F, dF/dx, dF/du  (forward tangents) (linearisation)                       488
//   MODE 2  sum_r a_r d2 F_r / d(x,u)2          (Lagrangian Hessian, 15 entries)      476
// Round 5: the fused solver's passes share what they compute instead of recomputing it (hb / tb of the arguments):
//   MODE 0  + stores the hidden activations h1, h2 of the trial point (the accepted trial point IS the next iterate)
//   MODE 3  MODE 1 from the stored activations (no layer 1 / 2 products, no sigmoids when they are valid); the tangents
//           m_c = W2^T (s'(A1) * W1[c, :]) are stored for MODE 4; the last layer on the vector pipe                320
//   MODE 4  MODE 2 from the stored activations and tangents: g1 = W2 e2 is the one 64 x 64 product left            68
#pragma once
#include <hip/hip_runtime.h>

namespace myriad {

typedef __attribute__((address_space(3))) double nd_lds;
typedef __attribute__((address_space(1))) double nd_glb;
typedef double nd4 __attribute__((ext_vector_type(4)));

#ifdef MYR_PHASE_TIMING
__device__ long long node_seg_[5][8];      // phase-timing builds: cycles of workgroup 0, wavefront 0 inside a pass, per MODE and segment (tools/dev/exp/exp104.sh)
#define MYR_NSEG(k) { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long t1_ = clock64(); node_seg_[MODE][k] += t1_ - tseg_; tseg_ = t1_; } }
#else
#define MYR_NSEG(k)
#endif
struct NodeMfma64 {     // NS = 4, NU = 1, hidden (64, 64)
  static constexpr int NS = 4, NU = 1, NW = 5, H = 64, LD2 = 65;
  // LDS block (doubles): W1 [8][64] (rows 5..7 zero) | W2 [64][65] (padded rows: column reads are conflict-free too) |
  // W3 [64][4] | b1 | b2 | b3
  static constexpr int L_W1 = 0, L_W2 = L_W1 + 8 * H, L_W3 = L_W2 + H * LD2, L_B1 = L_W3 + H * NS, L_B2 = L_B1 + H, L_B3 = L_B2 + H,
                       L_N = (L_B3 + NS + 7) / 8 * 8;
  // parameter vector in global memory (node_system.h): w1 [5][64] | b1 | w2 [64][64] | b2 | w3 [64][4] | b3
  static constexpr int O_W1 = 0, O_B1 = O_W1 + NW * H, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + H * NS;
  static constexpr int NPAIR = NW * (NW + 1) / 2;       // entries of the symmetric 5 x 5 contraction

  // (by `nt` threads, thread `t`: one wavefront in the solver, the whole workgroup in the evaluation kernel)
  __device__ static inline void load_weights(const double* p, double* wl, int t, int nt = 64) {
    for (int e = t; e < 8 * H; e += nt) wl[L_W1 + e] = e < NW * H ? p[O_W1 + e] : 0.0;
    for (int e = t; e < H * H; e += nt) wl[L_W2 + (e >> 6) * LD2 + (e & 63)] = p[O_W2 + e];
    for (int e = t; e < H * NS; e += nt) wl[L_W3 + e] = p[O_W3 + e];
    if (t < H) { wl[L_B1 + t] = p[O_B1 + t]; wl[L_B2 + t] = p[O_B2 + t]; }
    if (t < NS) wl[L_B3 + t] = p[O_B3 + t];
  }

  // the activation / tangent stores are streams (written once, read once or twice by the same lane, ~0.9 MB per trajectory and iteration):
  // non-temporal, so that they do not evict the solver's records from the L2 (-DMYR_NODE_NT=0: plain accesses)
#ifndef MYR_NODE_NT
#define MYR_NODE_NT 1
#endif
  __device__ static inline double ntl(const nd_glb* p) { return MYR_NODE_NT ? __builtin_nontemporal_load(p) : *p; }
  __device__ static inline void nts(double v, nd_glb* p) { if (MYR_NODE_NT) __builtin_nontemporal_store(v, p); else *p = v; }
  __device__ static inline nd4 mm(double a, double b, nd4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  // 1 / (1 + exp(-a)) with the reciprocal from v_rcp_f64 + two Newton steps (5 instructions instead of the ~35 of the IEEE
  // division sequence: a tile of 16 points takes 32 sigmoids per lane); <= 1 ulp
  __device__ static inline double sigm(double a) {
    const double d = 1.0 + exp(-a);
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
  }

  // The 64 x 64 products below are written k-step OUTER, output tile INNER: every k-step issues four INDEPENDENT matrix
  // instructions (one per output tile of 16 rows).  On MI355X a v_mfma_f64_16x16x4_f64 occupies the matrix pipe 64 cycles and a
  // dependent one can issue ~130 cycles after its producer -- the original nest (tile outer: sixteen dependent instructions in
  // a row per tile) left the pipe idle half of the time and more, with four accumulators in flight it is fed back to back.
  // out = W2^T in (+ b2): row m = 16 mt + i of the A operand, k = n = 16 t + 4 s + g
  // (the "memory" barrier in front of every product keeps the compiler from holding the 64 weight values of one product in
  // registers for the next ones -- six products share W2 in MODE 2, and 128 registers of hoisted weights pushed that pass into
  // 4.4 KB of scratch per lane; an LDS read per matrix instruction is what the layout was made for)
  // (Round 6: the weights of k-step ks + 1 are read BEFORE the four matrix instructions of k-step ks are issued, into registers of their own, and scheduling
  // barriers keep it that way.  The compiler's own order -- one ds_read2_b64 into one register quad, s_waitcnt lgkmcnt(0), two matrix instructions, 32 times --
  // exposed the LDS latency in front of every pair: a 64-instruction product took 9.6 k cycles for 4.1 k of matrix pipe, tools/dev/exp/exp104.sh.  Same
  // operations in the same order: same bits.  -DMYR_NODE_PIPE=0: the old form.)
#ifndef MYR_NODE_PIPE
#define MYR_NODE_PIPE 1
#endif
  template <bool BIAS>
  __device__ static inline void gemm_t(const nd_lds* wl, int g, int i, const nd4* in, nd4* out) {
    asm volatile("" ::: "memory");
    nd4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[mt][s] = BIAS ? wl[L_B2 + 16 * mt + 4 * s + g] : 0.0;
#if MYR_NODE_PIPE
    double w[2][4];
    {
      const nd_lds* row = wl + L_W2 + g * LD2 + i;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) w[0][mt] = row[16 * mt];
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 1 < 16) {
        const nd_lds* row = wl + L_W2 + (16 * ((ks + 1) >> 2) + 4 * ((ks + 1) & 3) + g) * LD2 + i;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) w[(ks + 1) & 1][mt] = row[16 * mt];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mm(w[ks & 1][mt], in[ks >> 2][ks & 3], acc[mt]);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const nd_lds* row = wl + L_W2 + (16 * t + 4 * s + g) * LD2 + i;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = mm(row[16 * mt], in[t][s], acc[mt]);
      }
#endif
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) out[mt] = acc[mt];
  }
  // out = W2 in: row n = 16 nt + i, k = m = 16 t + 4 s + g
  __device__ static inline void gemm_n(const nd_lds* wl, int g, int i, const nd4* in, nd4* out) {
    asm volatile("" ::: "memory");
    nd4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = nd4{0.0, 0.0, 0.0, 0.0};
#if MYR_NODE_PIPE
    double w[2][4];
    {
      const nd_lds* col = wl + L_W2 + i * LD2 + g;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) w[0][nt] = col[16 * nt * LD2];
    }
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 1 < 16) {
        const nd_lds* col = wl + L_W2 + i * LD2 + 16 * ((ks + 1) >> 2) + 4 * ((ks + 1) & 3) + g;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) w[(ks + 1) & 1][nt] = col[16 * nt * LD2];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = mm(w[ks & 1][nt], in[ks >> 2][ks & 3], acc[nt]);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const nd_lds* col = wl + L_W2 + i * LD2 + 16 * t + 4 * s + g;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = mm(col[16 * nt * LD2], in[t][s], acc[nt]);
      }
#endif
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) out[nt] = acc[nt];
  }
  // rows 0..3 of W3^T in (+ b3): result for output r of point i in lane 16 r + i.  One output tile only: the sixteen k-steps go
  // to four accumulators (one per s) that are added at the end.
  template <bool BIAS>
  __device__ static inline double layer3(const nd_lds* wl, int g, int i, const nd4* in) {
    nd4 acc[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = nd4{(BIAS && s == 0) ? wl[L_B3 + g] : 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const double a = wl[L_W3 + (16 * t + 4 * s + g) * NS + (i & 3)];
        acc[s] = mm(i < NS ? a : 0.0, in[t][s], acc[s]);
      }
    return (acc[0][0] + acc[1][0]) + (acc[2][0] + acc[3][0]);
  }

  // The same on the vector pipe (round 5): the last layer has four output rows -- a quarter of the matrix instruction's sixteen -- so
  // 64 multiply-adds per lane and one scattered sum over the lane groups (9 instructions) cost a third of the sixteen matrix
  // instructions (on this part the fp64 matrix rate IS the fp64 vector rate).  Output r of point i in lane 16 r + i.
  __device__ static inline double layer3_valu(const nd_lds* wl, int g, const nd4* in) {
    double f[NS];
#pragma unroll
    for (int r = 0; r < NS; ++r) f[r] = 0.0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const nd_lds* w = wl + L_W3 + (16 * mt + 4 * s + g) * NS;
#pragma unroll
        for (int r = 0; r < NS; ++r) f[r] = fma(w[r], in[mt][s], f[r]);
      }
    return group_scatter_sum(f[0], f[1], f[2], f[3]) + wl[L_B3 + g];
  }

  // ZT: address space of the iterate, the step and the multipliers (global in HsWave, LDS in the fused kernel)
  template <class ZT>
  struct ArgsT {
    const ZT* z; const ZT* dz; const ZT* lam;               // decision vector, step (MODE 0), multipliers (MODE 2)
    nd_glb* pt;                                             // per-point records, field f of point j at pt[f * K + j]
    nd_lds* sF;                                             // MODE 0: f of point j at sF[j * NS + r]
    double alpha, h6, h8;
    int K, N, pf_f, pf_a, pf_b, pf_d2;
    int t0 = 0, ts = 1;                                     // this wavefront's tiles of 16 points: t0, t0 + ts, ...
    nd_lds* rec = nullptr;                                  // MODE 1, evaluation kernel: F / A / B go to the LDS record of point j
    int use_rec = 0;                                        //   (an explicit flag: the record may sit at LDS offset 0)
    int rec_stride = 0, rec_f = 0, rec_a = 0, rec_b = 0;    //   rec[j * rec_stride + rec_f + r], + rec_a + r * NS + c, + rec_b + r * NU
    nd_glb* hb = nullptr;                                   // hidden activations of tile t: hb[(t * 32 + f) * 64 + lane], f = 4 mt + s (h1), 16 + 4 mt + s (h2)
    nd_glb* mb = nullptr;                                   // MODE 3 -> MODE 4: tangents m_c of tile t: mb[(t * 80 + c * 16 + 4 mt + s) * 64 + lane]
    int h_valid = 0;                                        // MODE 3: hb holds the activations of this iterate (the last trial was accepted)
    const ZT* avec = nullptr;                               // MODE 4: the contraction vector of point j given directly, avec[j * NS + r] (shooting: stage weights
                                                            //   of the step's costate, os_solver.h: step_lin / rk4_lin) instead of the Hermite-Simpson combination of `lam`
  };
  static constexpr int HB_TILE = 32 * 64, MB_TILE = NW * 16 * 64;
  __host__ __device__ static constexpr int ntiles(int K) { return (K + 15) / 16; }

  // sums over the four lane groups (lanes 16 g + i, g = 0..3), scattered: lane group g ends with the total of v[g].
  // v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-wavefronts / odd and even rows between two registers: 9 vector
  // instructions for four values where the shuffle form takes 8 LDS-crossbar moves and 8 additions.
  __device__ static inline double swap32_sum(double a, double b) {      // lanes < 32: a + a(lane ^ 32); lanes >= 32: b + b(lane ^ 32)
    auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  }
  __device__ static inline double swap16_sum(double a, double b) {      // even rows: a + a(lane ^ 16); odd rows: b + b(lane ^ 16)
    auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
  }
  __device__ static inline double group_scatter_sum(double v0, double v1, double v2, double v3) {
    return swap16_sum(swap32_sum(v0, v2), swap32_sum(v1, v3));
  }
  using Args = ArgsT<nd_glb>;

  template <int MODE, class ZT = nd_glb>
  __device__ __attribute__((noinline)) static void pass(const nd_lds* wl, ArgsT<ZT> a, int lane) {
    const int g = lane >> 4, i = lane & 15, K = a.K;
#ifdef MYR_PHASE_TIMING
    long long tseg_ = clock64();
#endif
    for (int j0 = 16 * a.t0; j0 < K; j0 += 16 * a.ts) {
      const bool valid = j0 + i < K;
      const int j = valid ? j0 + i : K - 1;
      // inputs in B-operand layout: state component g of point i, and the control in lane group 0 of the second k-step
      double xg = a.z[(long)j * NS + g], ug = g == 0 ? a.z[(long)K * NS + j] : 0.0;
      if (MODE == 0) {
        xg = fma(a.alpha, a.dz[(long)j * NS + g], xg);       // (explicitly fused: a helper workgroup gets fma(alpha, dz, z) from the owner and passes alpha = 0 -- same bits)
        if (g == 0) ug = fma(a.alpha, a.dz[(long)K * NS + j], ug);
      }
      // (s'(A) = h (1 - h) and s''(A) = s'(A) (1 - 2 h) are formed where they are used: keeping them as arrays next to h1, h2
      // cost MODE 2 64 more live registers than it had)
      MYR_NSEG(0)      // inputs
      nd4 h1[4], h2[4];
      nd_glb* const hbt = (MODE == 0 || MODE >= 3) && a.hb ? a.hb + (long)(j0 >> 4) * HB_TILE + lane : nullptr;
      const bool stored = (MODE == 4) || (MODE == 3 && a.h_valid);      // (wave-uniform)
      if (stored) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) { h1[mt][s] = ntl(&hbt[(4 * mt + s) * 64]); h2[mt][s] = ntl(&hbt[(16 + 4 * mt + s) * 64]); }
      } else {
        {
          nd4 a1[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) a1[mt][s] = wl[L_B1 + 16 * mt + 4 * s + g];
#if MYR_NODE_PIPE
          double wa[4], wb[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) { wa[mt] = wl[L_W1 + g * H + 16 * mt + i]; wb[mt] = wl[L_W1 + (4 + g) * H + 16 * mt + i]; }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a1[mt] = mm(wa[mt], xg, a1[mt]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a1[mt] = mm(wb[mt], ug, a1[mt]);
#else
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a1[mt] = mm(wl[L_W1 + g * H + 16 * mt + i], xg, a1[mt]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) a1[mt] = mm(wl[L_W1 + (4 + g) * H + 16 * mt + i], ug, a1[mt]);
#endif
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) h1[mt][s] = sigm(a1[mt][s]);
        }
        MYR_NSEG(1)    // layer 1 + sigmoids
        gemm_t<true>(wl, g, i, h1, h2);
        MYR_NSEG(2)    // layer 2 product
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) h2[mt][s] = sigm(h2[mt][s]);
        MYR_NSEG(3)    // sigmoids of layer 2
        if (hbt) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) { nts(h1[mt][s], &hbt[(4 * mt + s) * 64]); nts(h2[mt][s], &hbt[(16 + 4 * mt + s) * 64]); }
        }
      }
      MYR_NSEG(4)      // activations stored / loaded
      if (MODE == 0 || MODE == 1 || MODE == 3) {
        const double F = (MODE == 1) ? layer3<true>(wl, g, i, h2) : layer3_valu(wl, g, h2);
        if (MODE == 0) { if (valid) a.sF[j * NS + g] = F; }
        else if (valid) { if (a.use_rec) a.rec[j * a.rec_stride + a.rec_f + g] = F; else a.pt[(long)(a.pf_f + g) * K + j] = F; }
      }
      MYR_NSEG(5)      // last layer + F stored
      if (MODE == 3) {
        // All five forward tangents through W2 in ONE k-loop: a k-step's weight reads feed ten independent matrix instructions
        // (5 tangents x 2 output tiles; the output tiles in two halves), the seeds s'(A1) W1[c, :] are formed on the fly.  The tangents
        // m_c = W2^T (s'(A1) * W1[c, :]) go to `mb` for MODE 4; the last layer -- four output rows, a quarter of a matrix instruction
        // -- runs on the vector pipe: J[r][c] = sum_n W3[n][r] s'(A2_n) m_c[n], 16 units per lane, then over the lane groups.
        double J[NS][NW];
#pragma unroll
        for (int r = 0; r < NS; ++r)
#pragma unroll
          for (int c = 0; c < NW; ++c) J[r][c] = 0.0;
        nd_glb* const mbt = a.mb + (long)(j0 >> 4) * MB_TILE + lane;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          nd4 m[NW][2];
#pragma unroll
          for (int c = 0; c < NW; ++c) { m[c][0] = nd4{0.0, 0.0, 0.0, 0.0}; m[c][1] = nd4{0.0, 0.0, 0.0, 0.0}; }
#if MYR_NODE_PIPE
          // (the k-step's seven LDS values -- two weights of W2, five of W1 -- read one k-step ahead, as in gemm_t)
          double wq[2][2 + NW];
          {
            const nd_lds* row = wl + L_W2 + g * LD2 + 32 * half + i;
            wq[0][0] = row[0]; wq[0][1] = row[16];
#pragma unroll
            for (int c = 0; c < NW; ++c) wq[0][2 + c] = wl[L_W1 + c * H + g];
          }
#pragma unroll
          for (int ks = 0; ks < 16; ++ks) {
            if (ks + 1 < 16) {
              const int k1 = 16 * ((ks + 1) >> 2) + 4 * ((ks + 1) & 3) + g;
              const nd_lds* row = wl + L_W2 + k1 * LD2 + 32 * half + i;
              wq[(ks + 1) & 1][0] = row[0]; wq[(ks + 1) & 1][1] = row[16];
#pragma unroll
              for (int c = 0; c < NW; ++c) wq[(ks + 1) & 1][2 + c] = wl[L_W1 + c * H + k1];
            }
            __builtin_amdgcn_sched_barrier(0);
            const double hv1 = h1[ks >> 2][ks & 3];
            const double sp = hv1 * (1.0 - hv1);
#pragma unroll
            for (int c = 0; c < NW; ++c) {
              const double d = sp * wq[ks & 1][2 + c];
              m[c][0] = mm(wq[ks & 1][0], d, m[c][0]);
              m[c][1] = mm(wq[ks & 1][1], d, m[c][1]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#else
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int k = 16 * t + 4 * s + g;
              const nd_lds* row = wl + L_W2 + k * LD2 + 32 * half + i;
              const double w0 = row[0], w1 = row[16];
              const double sp = h1[t][s] * (1.0 - h1[t][s]);
#pragma unroll
              for (int c = 0; c < NW; ++c) {
                const double d = sp * wl[L_W1 + c * H + k];
                m[c][0] = mm(w0, d, m[c][0]);
                m[c][1] = mm(w1, d, m[c][1]);
              }
            }
#endif
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int mt = 2 * half + q;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const int n = 16 * mt + 4 * s + g;
              const double hv = h2[mt][s], sp2 = hv * (1.0 - hv);
#pragma unroll
              for (int c = 0; c < NW; ++c) nts(m[c][q][s], &mbt[(c * 16 + 4 * mt + s) * 64]);
#pragma unroll
              for (int r = 0; r < NS; ++r) {
                const double qr = wl[L_W3 + n * NS + r] * sp2;
#pragma unroll
                for (int c = 0; c < NW; ++c) J[r][c] = fma(qr, m[c][q][s], J[r][c]);
              }
            }
          }
        }
#pragma unroll
        for (int c = 0; c < NW; ++c) {
          const double dF = group_scatter_sum(J[0][c], J[1][c], J[2][c], J[3][c]);      // d F_g / d w_c at point i
          if (valid) {
            if (a.use_rec) a.rec[j * a.rec_stride + (c < NS ? a.rec_a + g * NS + c : a.rec_b + g * NU)] = dF;
            else a.pt[(long)(c < NS ? a.pf_a + g * NS + c : a.pf_b + g * NU) * K + j] = dF;
          }
        }
      }
      if (MODE == 4) {
        double ar;
        if (a.avec) ar = a.avec[(long)j * NS + g];
        else if (j & 1) ar = -4.0 * a.h6 * a.lam[(long)((j - 1) >> 1) * NS + g];
        else {
          const int kL = (j >> 1) - 1, kR = j >> 1;
          ar = 0.0;
          if (kL >= 0) ar += -a.h6 * a.lam[(long)kL * NS + g] + a.h8 * a.lam[(long)a.N * NS + (long)kL * NS + g];
          if (kR < a.N) ar += -a.h6 * a.lam[(long)kR * NS + g] - a.h8 * a.lam[(long)a.N * NS + (long)kR * NS + g];
        }
        const nd_glb* const mbt = a.mb + (long)(j0 >> 4) * MB_TILE + lane;
        // e2 = (W3 a) s'(A2); c2 = e2 (1 - 2 h2); g1 = W2 e2; c1 = g1 s''(A1)
        nd4 e2[4], c1[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) e2[nt] = mm(wl[L_W3 + (16 * nt + i) * NS + g], ar, nd4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int s = 0; s < 4; ++s) e2[nt][s] *= h2[nt][s] * (1.0 - h2[nt][s]);
        gemm_n(wl, g, i, e2, c1);
        double acc[NPAIR];
#pragma unroll
        for (int e = 0; e < NPAIR; ++e) acc[e] = 0.0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double cv1 = c1[mt][s] * (h1[mt][s] * (1.0 - h1[mt][s]) * (1.0 - 2.0 * h1[mt][s]));
            const double cv2 = e2[mt][s] * (1.0 - 2.0 * h2[mt][s]);
            double w1v[NW], mv[NW];
#pragma unroll
            for (int c = 0; c < NW; ++c) { w1v[c] = wl[L_W1 + c * H + 16 * mt + 4 * s + g]; mv[c] = ntl(&mbt[(c * 16 + 4 * mt + s) * 64]); }
            int e = 0;
#pragma unroll
            for (int pa = 0; pa < NW; ++pa) {
              const double wa = cv1 * w1v[pa], ma = cv2 * mv[pa];
#pragma unroll
              for (int pb = pa; pb < NW; ++pb, ++e) acc[e] = fma(ma, mv[pb], fma(wa, w1v[pb], acc[e]));
            }
          }
#pragma unroll
        for (int e0 = 0; e0 < NPAIR; e0 += 4) {
          const double v = group_scatter_sum(acc[e0], e0 + 1 < NPAIR ? acc[e0 + 1] : 0.0, e0 + 2 < NPAIR ? acc[e0 + 2] : 0.0,
                                             e0 + 3 < NPAIR ? acc[e0 + 3] : 0.0);
          if (valid && e0 + g < NPAIR) a.pt[(long)(a.pf_d2 + e0 + g) * K + j] = v;
        }
      }
      MYR_NSEG(6)      // MODE 3: tangents + Jacobian; MODE 4: contraction
      if (MODE == 1) {
        // forward tangents: d A1 / d w_c = W1[c, :] (constant), so d H1 = s'(A1) * W1[c, :], then the two upper layers
#pragma unroll
        for (int c = 0; c < NW; ++c) {
          nd4 d1[4], d2[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) d1[mt][s] = h1[mt][s] * (1.0 - h1[mt][s]) * wl[L_W1 + c * H + 16 * mt + 4 * s + g];
          gemm_t<false>(wl, g, i, d1, d2);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) d2[mt][s] *= h2[mt][s] * (1.0 - h2[mt][s]);
          const double dF = layer3<false>(wl, g, i, d2);      // d F_g / d w_c at point i
          if (valid) {
            if (a.use_rec) a.rec[j * a.rec_stride + (c < NS ? a.rec_a + g * NS + c : a.rec_b + g * NU)] = dF;
            else a.pt[(long)(c < NS ? a.pf_a + g * NS + c : a.pf_b + g * NU) * K + j] = dF;
          }
        }
      }
      if (MODE == 2) {
        // multiplier combination of this point (what points_hess() forms per lane): component g in lane group g
        double ar;
        if (j & 1) ar = -4.0 * a.h6 * a.lam[(long)((j - 1) >> 1) * NS + g];
        else {
          const int kL = (j >> 1) - 1, kR = j >> 1;
          ar = 0.0;
          if (kL >= 0) ar += -a.h6 * a.lam[(long)kL * NS + g] + a.h8 * a.lam[(long)a.N * NS + (long)kL * NS + g];
          if (kR < a.N) ar += -a.h6 * a.lam[(long)kR * NS + g] - a.h8 * a.lam[(long)a.N * NS + (long)kR * NS + g];
        }
        // g2 = W3 a (one k-step: k = output r = lane group), e2 = g2 s'(A2), c2 = g2 s''(A2)
        nd4 e2[4], c2[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) e2[nt] = mm(wl[L_W3 + (16 * nt + i) * NS + g], ar, nd4{0.0, 0.0, 0.0, 0.0});
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int s = 0; s < 4; ++s) { const double e = e2[nt][s] * (h2[nt][s] * (1.0 - h2[nt][s])); e2[nt][s] = e; c2[nt][s] = e * (1.0 - 2.0 * h2[nt][s]); }
        // g1 = W2 e2, c1 = g1 s''(A1)
        nd4 c1[4];
        gemm_n(wl, g, i, e2, c1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) c1[mt][s] *= h1[mt][s] * (1.0 - h1[mt][s]) * (1.0 - 2.0 * h1[mt][s]);
        // m_c = W2^T (s'(A1) * W1[c, :]): the five tangents of A2
        nd4 m[NW][4];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
          nd4 d1[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) d1[mt][s] = h1[mt][s] * (1.0 - h1[mt][s]) * wl[L_W1 + c * H + 16 * mt + 4 * s + g];
          gemm_t<false>(wl, g, i, d1, m[c]);
        }
        // W[a][b] = sum_n c1_n W1[a][n] W1[b][n] + sum_n c2_n m_a[n] m_b[n]: 16 hidden units per lane, then over the 4 groups
        double acc[NPAIR];
#pragma unroll
        for (int e = 0; e < NPAIR; ++e) acc[e] = 0.0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            double w1v[NW];
#pragma unroll
            for (int c = 0; c < NW; ++c) w1v[c] = wl[L_W1 + c * H + 16 * mt + 4 * s + g];
            int e = 0;
#pragma unroll
            for (int p = 0; p < NW; ++p)
#pragma unroll
              for (int q = p; q < NW; ++q, ++e)
                acc[e] += c1[mt][s] * w1v[p] * w1v[q] + c2[mt][s] * m[p][mt][s] * m[q][mt][s];
          }
#pragma unroll
        for (int e = 0; e < NPAIR; ++e) {
          double v = acc[e];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if (valid && (e & 3) == g) a.pt[(long)(a.pf_d2 + e) * K + j] = v;
        }
      }
    }
  }
};

// network dynamics (node_system.h) that the matrix-core passes above implement
template <class True, int H1, int H2, int ID_> struct SysNODE;
template <class Sys> struct NodeTraits { static constexpr bool mlp = false; static constexpr int lds_doubles = 0; };
template <class True, int ID_> struct NodeTraits<SysNODE<True, 64, 64, ID_>> {
  static constexpr bool mlp = (True::NS == 4 && True::NU == 1);
  static constexpr int lds_doubles = mlp ? NodeMfma64::L_N : 0;
};

}  // namespace myriad
