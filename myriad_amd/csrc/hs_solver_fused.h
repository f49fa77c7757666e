// hs_solver_fused.h -- the wavefront-per-trajectory Hermite-Simpson interior-point SQP of hs_solver_wave.h with the phases
// of an iteration FUSED so that the per-point / per-stage records stay on the CU.
//
// Round 2's kernel (HsWave) ran an iteration as thirteen phases that handed 223 KB of records per trajectory to each other
// through global memory: 80 GB per 4096-trajectory launch, 3.9 TB/s, 500 x the algorithmic I/O -- and it was that traffic, not
// the occupancy, that bound it (profiles/r03/exp01_noinline_two_waves.md: twice the resident wavefronts, no gain).  Here:
//   * the iterate (z, zL, zU) lives in LDS for the whole solve (24 KB for CARTPOLE N = 100); bounds that are the same for
//     every interior point (the reference's transcriptions never produce anything else, hermite_simpson.py:55-81) are
//     detected once and served from a 30-double LDS table;
//   * BACKWARD phase (lanes over intervals, one pass): apply the accepted step -> linearise the interval's knot and
//     midpoint -> take the end knot's linearisation from the neighbouring lane (registers, not memory) -> eliminate ->
//     adjoint recursion as a wave scan on the rows still in registers -> multipliers.  Only the elimination maps
//     Ge | Gm (64 doubles per stage) are written; f, A, B, the adjoint maps and the M | v rows never leave the lane;
//   * HESSIAN phase (lanes over points): second derivatives are RECOMPUTED from the iterate (one more sin / cos per point
//     is cheaper than a 54-field record written and read back), records are symmetric-packed (25 instead of 35 doubles);
//   * the Riccati sweep on the fp64 matrix cores is HsWave's (same tile, same chaining), reading the packed records;
//   * FORWARD phase (lanes over intervals, one pass): closed-loop maps -> wave scan -> step of the interval's midpoint and
//     end knot -> fraction-to-the-boundary limits and merit slope of those points, on the values just computed.
// Per iteration and trajectory: ~40 k doubles of global traffic instead of ~100 k, and six dependent memory round trips
// instead of thirty.  The algorithm, its constants and its control flow are those of HsWave::solve / HsSolver::solve; the
// kernels are tested against each other (tests/test_gpu_solve.py).
// Reference: the problem is the one /root/reference/myriad/trajectory_optimizers/collocation/hermite_simpson.py builds
// (constraints :325-335, objective :243-257, bounds :55-81); the solve replaces myriad/nlp_solvers/__init__.py:57-58.
#pragma once
#include <hip/hip_runtime.h>
#include "hs_solver_wave.h"

namespace myriad {

// A condition that is the same in every lane by the algorithm but not by the compiler's analysis, as a wave-uniform value: the branch on
// it is a scalar branch (s_cbranch_scc), not an EXEC-masked region.  Used by the sweeps of the W > 1 kernels (inlined into a kernel whose
// wavefronts take different paths; measured there: B = 256 / 512 kernel 2.95 / 3.10 ms against 3.01 / 3.15 ms); the sweep W = 1 calls as a
// function keeps the per-lane form (all-or-none masks in uniform control flow, gated by the same tests: the scalar form costs it 1.6 %,
// 14.55 against 14.33 ms -- profiles/r04/README.md).  -DMYR_SWEEP_UNIFORM=0 / 1 forces one form everywhere.
#ifndef MYR_SWEEP_UNIFORM
#define MYR_SWEEP_UNIFORM -1
#endif
template <bool ON>
__device__ inline bool uniform_if(bool c) {
  if constexpr (ON) return __builtin_amdgcn_readfirstlane((int)c) != 0;
  else return c;
}

// lane l <- lane l - 1 (lane 0: unspecified, the callers overwrite it)
__device__ inline double lane_up1(double v) { return __shfl_up(v, 1, 64); }

// W wavefronts share one trajectory (W = 1: the throughput form, four trajectories per CU; W = 2: the small-batch form, each
// wavefront takes every other block of 64 stages / points and they meet at workgroup barriers -- the carries of the two wave
// scans and the neighbour records cross through LDS; the Riccati sweep runs in wavefront 0).
// SCHEME 0: Hermite-Simpson (K = 2N+1 points, stage unknowns y = (dx_s, du_s, du_m, du_e), two eliminated controls);
// SCHEME 1: trapezoidal collocation (/root/reference/myriad/trajectory_optimizers/collocation/trapezoidal.py:80-163; K = N+1 points,
// y = (dx_s, du_s, du_e), one eliminated control, no midpoint) -- the same passes, the algorithm of TrapCore (os_solver.h).
// Two-phase launch of the one-wavefront kernel (myriad_hip.hip: launch_hs_fused_w).  A batch of B > (resident wavefronts) trajectories runs as whole
// solves, several per slot, and the launch ends with the slots that drew the long solves last: 98 iterations' worth of time for the headline batch where
// a divisible load would take 77.5.  Phase 1 gives EVERY trajectory its first k1 iterations (equal work per slot) and parks the unfinished ones -- the
// solver's LDS and the loop's scalars, 40 KB -- in global memory; the residuals at that point predict what is left (rank correlation 0.95), so phase 2
// resumes them longest-first (`perm`) and the ragged end shrinks to a few iterations.  The arithmetic of a trajectory is the same, bit for bit.
struct ParkArgs {
  int mode;              // 0: whole solves; 1: stop at the top of iteration k1 and park; 2: ticket t resumes trajectory perm[t]
  int k1;
  const int* perm;       // mode 2: trajectories in resume order; count[0] of them
  const int* count;
  double* state;         // [B][stride]: the scalars of the loop, then the solver's LDS
  long stride;
};
constexpr int MYR_STATUS_PARKED_ = 5;     // internal: never leaves myr_solve

// Helper workgroups for the network passes (round 5).  Two thirds of an iteration of the network kernel are matrix-core passes over 13 independent tiles of
// 16 points, and a workgroup owns a CU: a batch smaller than the device (config 5's share of an 8-GPU node is 128 trajectories) leaves CUs idle for the
// whole launch, and every larger batch leaves them idle at its end, while the last -- longest -- solves run alone (one solve in a thousand takes 75
// iterations against a median of 24).  A workgroup without a trajectory therefore ATTACHES itself to a running one (an "owner") on its own XCD, at most
// `maxh` per owner, fewest-helpers-first.  The owner looks at its helper count once per iteration; for every pass it posts a command (sequence number,
// mode, helper count) and the few KB a helper needs (the evaluation point, or the multipliers) in global memory, all 4 (nh + 1) wavefronts deal the
// tiles round robin, the helpers write their results into the OWNER's records and raise a counter, the owner waits for it.  A tile is the same
// instructions on the same inputs whoever runs it, and a change of the helper count only makes the next linearisation recompute the stored
// activations (bit-identical to the stored ones): results do not depend on who helped when (tests/test_gpu_node.py).  Small batches (B <= grid) give
// trajectory b to workgroup b, so that owners and helpers spread evenly over the XCDs; larger ones draw tickets as before and help once the tickets
// are gone.  All workgroups of the launch are resident (one per CU, grid <= #CU), so the waits cannot deadlock; every wait is bounded all the same and
// raises `abort` instead of hanging the device.
struct NodeBoard {
  unsigned long long cmd;       // (sequence number << 32) | (generation << 16) | (helpers taking part << 12) | (activations valid << 8) | mode; mode 15: the solve is over
  unsigned int done;            // passes finished by helpers, cumulative: the owner waits for the sum of the helper counts it posted
  unsigned int xcc;             // 1 + the owner's XCD
  unsigned int att;             // bit 31: a solve is running here; bits 16..30: its generation (one per trajectory of this owner); low 16 bits: helpers attached to it
  unsigned int pad_[27];        // 128 B apart
};
struct CoopArgs {
  NodeBoard* boards;            // [workgroups that can own a trajectory]
  double* pub;                  // [same][pub_stride]: x (n = K NW) | multipliers (MLAM N NS) | f of the trial point (K NS)
  long pub_stride;
  int maxh;                     // helpers per owner at most (0: none; the kernel then is the round-4 one)
  int* abort;                   // set when a wait ran into its bound; abort[1]: trajectories finished
};
constexpr unsigned int MYR_COOP_RUNNING = 0x80000000u;
constexpr int MYR_COOP_EXIT = 15;
// The generation closes an ABA window of the attach (round 6): a helper reads att = RUNNING | k, then the command sequence, then raises the count by a
// compare-and-swap.  Without it an owner that finished its trajectory and started the next one in between (att back to RUNNING | 0 ... k) let the swap
// succeed against the NEW trajectory; the helper then took the OLD trajectory's EXIT for its own and left, still counted in, and the owner waited into its
// bound.  With the generation in the word the swap fails across trajectories; with it in every command a helper of an earlier trajectory that has not
// seen its EXIT yet (the command word is overwritten by the next trajectory's first command) leaves instead of serving under a helper index that a new
// helper holds.  The owner orders its three stores -- att <- 0, EXIT, att <- RUNNING | generation -- by waiting for each to leave the CU.
constexpr unsigned int MYR_COOP_GEN_MASK = 0x7fffu;
// Visibility between an owner and its helpers WITHOUT agent-scope fences: those write the whole L2 back (buffer_wbl2) -- measured with 256 workgroups on
// the device, B = 128 with one helper each: 14.0 ms against 9.4 ms without helpers (tools/dev/exp/exp54.sh).  Owner and helpers sit on the SAME XCD (a
// helper reads its XCD from HW_REG_XCC_ID and only attaches to owners whose board names the same one),
// i.e. behind the same L2: the producer only has to wait until its stores have left the CU (the vector L1 is write-through: s_waitcnt vmcnt(0)), the
// consumer only has to drop its L1 (buffer_inv sc1: no write-back); the flag words are relaxed atomics, executed at the L2.
__device__ inline void coop_release() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
__device__ inline void coop_acquire() { asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory"); }
#ifdef MYR_COOP_DEBUG
#define COOP_DBG(bit, ...) do { if ((MYR_COOP_DEBUG) & (bit)) printf(__VA_ARGS__); } while (0)
#else
#define COOP_DBG(bit, ...)
#endif
#ifdef MYR_COOP_TRACE      // developer aid: state words in pinned host memory, dumped by a host thread when a launch takes too long (myriad_hip.hip)
__device__ int* g_coop_trace;
#define COOP_TR(slot, val) do { if (g_coop_trace) __hip_atomic_store(g_coop_trace + blockIdx.x * 16 + (slot), (int)(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
#else
#define COOP_TR(slot, val)
#endif
__device__ inline unsigned int coop_xcc_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u; }      // hwreg(HW_REG_XCC_ID)
constexpr unsigned MYR_COOP_SPIN_MAX = 1u << 23;

#ifdef MYR_PHASE_TIMING
__device__ long long node_tph_[8];      // cycles of workgroup 0's network passes (MODE 0 / 1 / 2), wavefront 0
__device__ long long bw_tph_[4];        // the backward pass of workgroup 0 in parts: linearisation | stage maps | adjoint scan + multipliers (the pass runs on a COPY of the context)
#define MYR_BW_PH(i) { if (blockIdx.x == 0 && c.tid == 0) { const long long t1_ = clock64(); bw_tph_[i] += t1_ - bw_t0_; bw_t0_ = t1_; } }
#else
#define MYR_BW_PH(i)
#endif

template <class Sys, int W = 1, int SCHEME = 0>
struct HsFused {
  static constexpr int NT = 64 * W;
  static constexpr bool TRAP = SCHEME == 1;
  using W0 = HsWave<Sys, 0>;
  using S = HsSolver<Sys>;
  using D = HsSol<Sys>;
  static constexpr int NS = D::NS, NU = D::NU, NW = D::NW, NY = TRAP ? NS + 2 * NU : D::NY, NQ = TRAP ? NU : D::NQ, NC = D::NC, NY1 = NY + 1;
  static constexpr int QE = NQ - NU;
  // Two-level sweep (round 6; -DMYR_TWO_LEVEL=0: round 5's form): the W wavefronts a trajectory owns each condense a CHUNK of N / W stages in parallel
  // (riccati_chunk / riccati_chunk_trap), a small interface recursion joins the chunks (tl_join), see there.  Both collocation schemes on the hand-placed
  // tile (one control, NS <= 4).
#ifndef MYR_EARLY_EXIT
#define MYR_EARLY_EXIT 1       // (round 6) convergence tests and barrier update in front of the sweep in every kernel form (HsFused::solve): the converged iteration returns
                               // without its sweep -- bit-identical results, 12.61 -> 12.45 ms per headline solve (tools/dev/exp/exp87.sh); 0: round 5's order
#endif
#ifndef MYR_SWEEP_CARRY
#define MYR_SWEEP_CARRY 2      // (round 6) fewer operand moves in a sweep stage: 1 = C operands whose upper half is zero inherit it from the previous result instead of a zero
                               // fill (bit-identical); 2 = also no C tuple for the midpoint product, its control rows are added where Q is consumed (last-bit differences, same
                               // iteration counts): 788 -> 703 instructions per four stages, with the early exit 12.61 -> 12.25 ms per headline solve; 0: round 5's code
#endif
#ifndef MYR_TWO_LEVEL
#define MYR_TWO_LEVEL 1
#endif
#ifndef MYR_TL_SPEC
#define MYR_TL_SPEC 0           // 1: closed-form systems get a four-wavefront form for batches of at most one trajectory per CU -- two chunks x two rungs of the inertia
                                // ladder at a time (HsFused::TLS).  Built and measured (tools/dev/exp/exp90.sh): the iterates of the two-wavefront form, bit for bit, and
                                // B = 256 100.5 -> 103.7 k, B = 128 51.9 -> 53.3 k solves/s -- 3 % for 160 KB of code per system: off.
#endif
#ifndef MYR_FWD_SEQ_NW
#define MYR_FWD_SEQ_NW 7        // stages of at least this many knot variables run the forward phase's recursion sequentially (HsFused::FSEQ); 0: every system scans
#endif
#ifndef MYR_ZLU_GLOBAL_WIDE
#define MYR_ZLU_GLOBAL_WIDE 1       // wide closed-form systems keep the bound multipliers in global scratch where that doubles the workgroups per CU (HsFused::ZLU_GLOBAL)
#endif
#ifndef MYR_TL_FLOOR
#define MYR_TL_FLOOR 1e-10      // smallest pivot accepted in an interface's C = I + L^T M L (its eigenvalues lie in (0, ~1] when the reduced Hessian is positive definite)
#endif
  static constexpr bool TL = (MYR_TWO_LEVEL != 0) && W > 1 && (D::NU == 1 && D::NS <= 4);
  static constexpr int MLAM = TRAP ? 1 : 2;        // multiplier blocks (NS each) per interval
  // Network dynamics (config 5, node_system.h): f, A, B of ALL points come from the matrix-core pass of node_mfma.h (MODE 1, every
  // wavefront of the workgroup takes every W-th tile of 16 points) into a global record the backward pass reads instead of calling
  // Sys::lin; MODE 2 delivers the multiplier-contracted second derivatives for the hessian pass, MODE 0 the values for the trials
  // (into LDS).  The weights (40 KB) sit in LDS behind the iterate.  Round 2 ran these systems on HsWave with the three helper
  // wavefronts of a workgroup idle outside the passes; here every parallel pass is shared by the W = 4 wavefronts.
  static constexpr bool MLP = NodeTraits<Sys>::mlp;
  static_assert(!(MLP && TRAP), "network dynamics are built for the Hermite-Simpson transcription");
  static constexpr int PT_F = 0, PT_A = PT_F + NS, PT_B = PT_A + NS * NS, PT_D2 = PT_B + NS * NU, PT_N = PT_D2 + NW * (NW + 1) / 2;
  static constexpr int HSYM = NW * (NW + 1) / 2;
  // per-point Hessian record, AoS: upper triangle of H (row-major), g0 (NW), g1 (NW)
  static constexpr int HR_H = 0, HR_G0 = HSYM, HR_G1 = HR_G0 + NW, HR_N = HR_G1 + NW;
  // per-stage record, AoS: Ge | ge, Gm | gm
  static constexpr int SG_GE = 0, SG_GM = NS * NY1, SG_N = (TRAP ? 1 : 2) * NS * NY1;
  static constexpr int KST = NQ * NW + NQ * NC, KSTR = KST;      // gains K | kc per stage
  static constexpr int ZR = 32;                                  // block of zeros (masked sweep lanes read it) + 2 write-only slots
  static constexpr int PF = MYR_RICCATI_PF;
  static constexpr int PADH = PF * 2 * HR_N, PADS = PF * SG_N;   // the sweep's prefetch ring reads PF stages below stage 0
  __host__ __device__ static constexpr int symidx(int r, int c) { return r <= c ? r * NW - r * (r - 1) / 2 + (c - r) : c * NW - c * (c - 1) / 2 + (r - c); }
  __host__ __device__ static constexpr int npoints(int N) { return TRAP ? N + 1 : 2 * N + 1; }
  // quadrature weight and time of point j (hermite_simpson.py:212-214 / trapezoidal.py:80-94)
  __host__ __device__ static inline double wq(int K, int j, double h) {
    if (TRAP) return (j == 0 || j == K - 1) ? 0.5 * h : h;
    return S::wsimp(K, j, h);
  }
  __host__ __device__ static inline double tq(int j, double h) { return TRAP ? h * j : 0.5 * h * j; }

  // global scratch per resident wavefront (doubles): zeros | pad | hr | pad | st | gains | multipliers
  __host__ __device__ static long off_hr(int) { return ZR + PADH; }
  __host__ __device__ static long off_st(int N) { return off_hr(N) + (long)npoints(N) * HR_N + PADS; }
  __host__ __device__ static long off_kg(int N) { return off_st(N) + (long)N * SG_N; }
  __host__ __device__ static long off_lam(int N) { return off_kg(N) + (long)N * KST; }
  __host__ __device__ static long off_kg2(int N) { return off_lam(N) + (long)MLAM * N * NS; }      // W = 2: gains of the speculative second sweep
  __host__ __device__ static long off_pt(int N) { return off_kg2(N) + (W > 1 ? (long)N * KST : 0); }     // network systems: point records (SoA)
  // network systems (round 5): hidden activations of the last evaluated point and the tangents of the linearisation, per tile of 16 points
  // and lane (node_mfma.h: MODE 0 / 3 write, MODE 3 / 4 read -- always the lane that wrote them)
  __host__ __device__ static long off_hb(int N) { return off_pt(N) + (MLP ? (long)PT_N * npoints(N) : 0); }
  __host__ __device__ static long off_mb(int N) { return off_hb(N) + (MLP ? (long)NodeMfma64::ntiles(npoints(N)) * NodeMfma64::HB_TILE : 0); }
  // The two-wavefront form of the network kernel (round 5: the THROUGHPUT form for batches beyond one trajectory per CU -- two trajectories per CU, each on
  // two SIMDs, so that one trajectory's sequential sweep overlaps the other's matrix-core passes) keeps the bound multipliers zL, zU in its global
  // scratch slot instead of LDS: two workgroups of 42.7 KB + 40.5 KB of weights do not fit a CU's 160 KB, two of 26.7 KB + 40.5 KB do.
  __host__ __device__ static long off_zlu(int N) { return off_mb(N) + (MLP ? (long)NodeMfma64::ntiles(npoints(N)) * NodeMfma64::MB_TILE : 0); }
  // Wide stages (round 6): the forward phase's closed-loop maps (NW x NW | NW per stage) go through global scratch to ONE sequential pass instead of a wave
  // scan over affine maps held in registers -- three NW x NW matrices per lane are 216 doubles at NW = 8, 630 at NW = 14, against the 128 double registers a
  // lane can address: the scan of ROCKETLANDING was 41 % of its iteration, most of it spill traffic (tools/dev/exp/exp92.sh, exp93.sh).
  static constexpr bool FSEQ = (MYR_FWD_SEQ_NW > 0) && NW >= MYR_FWD_SEQ_NW && !MLP;
  static constexpr int FWS = NW * NW + NW;
  __host__ __device__ static long off_fw(int N) { return off_zlu(N) + (ZLU_GLOBAL ? 2L * npoints(N) * NW : 0); }
  __host__ __device__ static long scratch_doubles(int N) { return off_fw(N) + (FSEQ ? (long)N * FWS : 0); }
  // LDS (doubles): z | zL | zU | dz | multipliers | bound table | neighbour stash | first-point exchange
  static constexpr int NREC = NS + NS + NS * NS + NS * NU + NS;   // x, f, A, B, own: what an interval takes from its end knot
  static constexpr int EXCH = NW * NW + NW * NC + NS * NC + NU * NC;
  // exchange between blocks / wavefronts (double-buffered by round): neighbour record, block total of a scan, trial knot; partial sums
  static constexpr int NTOT = NW * NW + NW, NRED = 12;
  static constexpr int XCH = 2 * W * NREC + 2 * W * NTOT + 2 * W * 2 * NS + W * NRED + 8;
  // sweep outputs in LDS: one block (P | pc | Tnu | Ku) per set; two-level form: one per chunk, the multiplier vector theta of every chunk (+ the chunks'
  // pivot counts), and per interface the four NW x NW maps of the join's forward pass
  // TLS (closed-form systems, W = 4: batches of at most one trajectory per CU): TWO chunks and TWO rungs of the inertia ladder at a time -- wavefronts 0, 1
  // sweep the chunks with delta, wavefronts 2, 3 the same chunks with the NEXT candidate (round 5's speculative rung on top of the two-level sweep: a failed
  // rung costs a whole sweep + join, its successor is then already there; same sequence of candidates, first success wins).  Four chunks instead were
  // measured equal to two (three sequential joins, tools/dev/exp/exp84.sh).
  static constexpr bool TLS = TL && W == 4 && !NodeTraits<Sys>::mlp && (MYR_TL_SPEC != 0);
  static constexpr int NCH = TLS ? 2 : W;       // chunks of a sweep
  static constexpr int NGR = TLS ? 2 : 1;       // rungs swept side by side
  static constexpr int NXB = TL ? NCH * NGR : (W > 1 ? 2 : 1);
  static constexpr int TL_TH1 = NCH * NC, TL_JN1 = (NCH - 1) * 4 * NW * NW;      // per rung group: theta table; the interfaces' maps
  static constexpr int TL_TH = TL ? NGR * TL_TH1 + NGR * (NCH + 1) + 2 : 0, TL_JN = TL ? NGR * TL_JN1 : 0;
  __host__ __device__ static constexpr int lds_solver_doubles_z(int N, bool zlu_global) { return (zlu_global ? 2 : 4) * npoints(N) * NW + MLAM * N * NS + 6 * NW + XCH + NXB * EXCH + TL_TH + TL_JN + 8; }
  // Round 6: the same for the wide closed-form systems (at least MYR_FWD_SEQ_NW knot variables) in the forms whose LDS, with the multipliers resident, would
  // not let 4 / W workgroups onto a CU at the reference horizon of 100 intervals -- ROCKETLANDING's one-wavefront Hermite-Simpson form (65.6 -> 39.9 KB: four
  // per CU instead of two, 94.5 -> 65.0 ms per 4096 solves of 30 iterations), CARTPOLE's twin (192 -> 111 ms), ROCKETLANDING's twin (one-wavefront form
  // 109 -> 64 KB, two-wavefront form 118 -> 73 KB: two per CU instead of one, 397 -> 205 ms); tools/dev/exp/exp97.sh.  -DMYR_ZLU_GLOBAL_WIDE=0: resident.
#ifndef MYR_ZLU_FORCE
#define MYR_ZLU_FORCE 0         // experiment (tools/dev/exp/exp110.sh): 1 = every one-wavefront closed-form kernel keeps the bound multipliers in scratch
#endif
  static constexpr bool ZLU_GLOBAL = (MLP && W == 2) || ((MYR_ZLU_FORCE != 0) && !MLP && W == 1) ||
      ((MYR_ZLU_GLOBAL_WIDE != 0) && !MLP && (MYR_FWD_SEQ_NW > 0) && NW >= MYR_FWD_SEQ_NW && (long)lds_solver_doubles_z(100, false) * 8 > 40960L * W);
  __host__ __device__ static int lds_solver_doubles(int N) { return lds_solver_doubles_z(N, ZLU_GLOBAL); }
  __host__ __device__ static int lds_doubles(int N) { return lds_solver_doubles(N) + (MLP ? npoints(N) * NS + NodeTraits<Sys>::lds_doubles : 0); }
  __host__ __device__ static size_t lds_bytes(int N) { return (size_t)lds_doubles(N) * 8; }
  static constexpr int NSCAL = 48;      // scalars of the solve loop in a parked trajectory's record
  __host__ __device__ static long park_doubles(int N) { return NSCAL + lds_solver_doubles(N) + (ZLU_GLOBAL ? 2L * npoints(N) * NW : 0); }      // (+ the bound multipliers of the forms that keep them in the slot's scratch)

  struct Ctx {
    int N, K, n, lane, wave, tid;
    double h, h6, h8;
    double *z, *zL, *zU, *dz;             // the iterate and the step (LDS)
    double *hr, *st, *kg, *zr;            // global scratch of this wavefront
    double *fw;                           // wide stages: the forward phase's closed-loop maps (global scratch)
    double *kgA, *kgB, *xA, *xB;          // W = 2: the two sets of sweep outputs (gains in global scratch, first-point exchange in LDS)
    double *sTh, *sJn;                    // two-level sweep: theta per chunk | pivot counts; the interfaces' maps (LDS)
    double *pt, *sF, *wl;                 // network systems: point records (global), trial values and weights (LDS)
    double *hb, *mb;                      //   stored activations / tangents (global, per tile and lane)
    int nh, hidx;                         //   helper workgroups of this trajectory, this workgroup's index among them (0: the owner)
    unsigned int gen;                     //   generation of the trajectory (owner: its own count; helper: the one it attached under)
    NodeBoard* board; int* coop_abort;    //   the owner's board
    double *pubx, *publam, *pubf;         //   what the owner publishes for a pass / the helpers' trial values (global)
    bool h_valid;                         //   hb holds the activations of the iterate (the last trial point was accepted)
    const double *lb, *ub;                // the caller's bounds (global)
    bool uni;                             // interior points share one bound per component: served from sB
    SysParams<Sys> pp;
    bool term_pinned[NS];
    double *sLam, *sB, *sStash, *sTot, *sTr, *sRed, *sMisc, *sP, *sPc, *sTnu, *sKu;   // LDS
#ifdef MYR_PHASE_TIMING
    long long tph[16], t0;
#endif
#ifdef MYR_TRACE
    int traj;
#endif
  };
  __device__ static inline long zi(const Ctx& c, int j, int comp) { return S::zi(c.K, j, comp); }
  // point the sweep outputs (gains, P | pc | Tnu | Ku exchange) of a context at one of the two sets
  __device__ static inline void use_set(Ctx& c, double* kg, double* x) {
    c.kg = kg; c.sP = x; c.sPc = x + NW * NW; c.sTnu = c.sPc + NW * NC; c.sKu = c.sTnu + NS * NC;
  }
  __device__ static inline void wsync() {
#ifdef MYR_WG_STRONG      // experiment: agent-scope fences around the workgroup barrier (s_waitcnt vmcnt(0) + L1 invalidate) when wavefronts share a trajectory
    if constexpr (W > 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); return; }
#endif
    __syncthreads();
  }
  // partial results of the W wavefronts -> the workgroup's (op: 0 sum, 1 max, 2 min); every wavefront ends with the same values
  template <int NV>
  __device__ static inline void wg_combine(Ctx& c, double* v, const int (&op)[NV]) {
    static_assert(NV <= NRED, "partial-sum slots");
    if constexpr (W > 1) {
      // (EVERY lane stores the wave-uniform partial result -- 64 stores of one value to one address -- instead of lane 0 alone: a lane-0 region here
      // ended the passes in a join block, and a spill the compiler put in FRONT of that block's EXEC restore missed lanes 1..63)
#pragma unroll
      for (int i = 0; i < NV; ++i) c.sRed[c.wave * NRED + i] = v[i];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double r = c.sRed[i];
#pragma unroll
        for (int w = 1; w < W; ++w) {
          const double t = c.sRed[w * NRED + i];
          r = op[i] == 0 ? r + t : (op[i] == 1 ? (r > t ? r : t) : (r < t ? r : t));
        }
        v[i] = r;
      }
    } else { (void)c; (void)v; (void)op; }
  }

  // ---- helper workgroups (see NodeBoard) ------------------------------------------------------------------------------------------------------
  __device__ static inline unsigned int& coop_seq(Ctx& c) { return reinterpret_cast<unsigned int*>(c.sMisc)[6]; }         // owner: commands posted (sMisc[3])
  __device__ static inline unsigned int& coop_target(Ctx& c) { return reinterpret_cast<unsigned int*>(c.sMisc)[7]; }      // owner: helper passes to wait for, cumulative
  __device__ static inline bool coop_bounded(Ctx& c, unsigned int& spins) {      // a wait has run into its bound (or somebody else's has)
    if (++spins > MYR_COOP_SPIN_MAX || __hip_atomic_load(c.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      if (spins > MYR_COOP_SPIN_MAX) __hip_atomic_store(c.coop_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return true;
    }
    return false;
  }
  // owner, once per iteration: how many helpers are attached now?  (A change moves the tiles between wavefronts: the stored activations are recomputed.)
  __device__ static inline void coop_refresh(Ctx& c) {
    if constexpr (MLP) {
      unsigned int* w = reinterpret_cast<unsigned int*>(c.sMisc) + 8;
      if (c.tid == 0) *w = __hip_atomic_load(&c.board->att, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffu;
      __syncthreads();
      const int a = __builtin_amdgcn_readfirstlane((int)*w);
      if (c.tid == 0) { COOP_TR(0, 100 + a); } if (c.tid == 64) { COOP_TR(11, 100 + a); } if (c.tid == 128) { COOP_TR(12, 100 + a); } if (c.tid == 192) { COOP_TR(13, 100 + a); }
      if (a != c.nh) { if (c.tid == 0) COOP_DBG(1, "[wg %d] owner: helpers %d -> %d\n", (int)blockIdx.x, c.nh, a); c.nh = a; c.h_valid = false; }
    } else (void)c;
  }
  // owner: publish what pass MODE reads, then the command
  template <int MODE>
  __device__ static inline void coop_post(Ctx& c, double alpha) {
    if constexpr (MLP) {
      if (MODE == 0) { for (int i = c.tid; i < c.n; i += NT) c.pubx[i] = fma(alpha, c.dz[i], c.z[i]); }
      if (MODE == 3 && !c.h_valid) { for (int i = c.tid; i < c.n; i += NT) c.pubx[i] = c.z[i]; }
      if (MODE == 4) { for (int i = c.tid; i < MLAM * c.N * NS; i += NT) c.publam[i] = c.sLam[i]; }
      if (c.tid == 0) COOP_TR(9, 1);
      coop_release();
      __syncthreads();
      if (c.tid == 0) {
        const unsigned int seq = ++coop_seq(c);
        COOP_TR(1, seq); COOP_TR(9, 2); COOP_TR(3, MODE);
        coop_target(c) += (unsigned int)c.nh;
        if (seq < 6) COOP_DBG(2, "[wg %d] owner: post seq %u mode %d nh %d target %u\n", (int)blockIdx.x, seq, MODE, c.nh, coop_target(c));
        __hip_atomic_store(&c.board->cmd, ((unsigned long long)seq << 32) | ((unsigned long long)(c.gen & MYR_COOP_GEN_MASK) << 16) | ((unsigned long long)c.nh << 12) | (c.h_valid ? 256ull : 0ull) | (unsigned long long)MODE,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else { (void)c; (void)alpha; }
  }
  // owner: wait for the helpers' share of the pass (their records are in this slot's global scratch; trial values arrive through pubf)
  template <int MODE>
  __device__ static inline void coop_wait(Ctx& c) {
    if constexpr (MLP) {
      if (c.tid == 0) {
        const unsigned int target = coop_target(c);
        unsigned int spins = 0;
        COOP_TR(2, target); COOP_TR(9, 3);
        while ((int)(__hip_atomic_load(&c.board->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
          __builtin_amdgcn_s_sleep(4);
          if (coop_bounded(c, spins)) { COOP_DBG(4, "[wg %d] owner: wait for %u ran into its bound (done %u)\n", (int)blockIdx.x, target, c.board->done); break; }
        }
        if (coop_seq(c) < 6) COOP_DBG(4, "[wg %d] owner: seq %u done\n", (int)blockIdx.x, coop_seq(c));
      }
      if (c.tid == 0) { COOP_TR(9, 4); COOP_TR(14, __hip_atomic_load(&c.board->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); COOP_TR(15, __hip_atomic_load(c.coop_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
      __syncthreads();
      if (c.tid == 0) COOP_TR(9, 5);
      coop_acquire();
      if (MODE == 0) {
        const int ts = W * (c.nh + 1);
        for (int j = c.tid; j < c.K; j += NT)
          if (((j >> 4) % ts) >= W) {
#pragma unroll
            for (int q = 0; q < NS; ++q) c.sF[j * NS + q] = c.pubf[j * NS + q];
          }
      }
    } else (void)c;
  }
  // helper number c.hidx of the owner at c.board: serve its passes until its solve is over (commands after `seen`).
  // EVERY lane polls the command word (a broadcast load) and the loop has no lane-0-only block next to its back edge: with `if (tid == 0) add` at the
  // bottom and `if (tid == 0) poll` at the top the compiler threaded the two into one divergent region across the back edge, the structurizer turned
  // lane 0's path into a loop EXIT, and lane 0 -- parked until the other lanes leave a loop they never leave -- never raised the counter: every owner
  // waited into its bound (tools/dev/exp/exp56.sh; builds with a printf in the region came out right).
  __device__ static void coop_serve(Ctx& c, unsigned int seen) {
    if constexpr (MLP) {
      unsigned int expect = seen + 1;
      bool finished_pass = false;
      for (;;) {
        __syncthreads();                   // (the previous pass: every wavefront's results have left the CU)
        if (finished_pass && c.tid == 0) { (void)__hip_atomic_fetch_add(&c.board->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); COOP_TR(8, 3); }
        unsigned long long v; unsigned int spins = 0;
        COOP_TR(4, expect);
        for (;;) {
          v = __hip_atomic_load(&c.board->cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
          if ((int)((unsigned int)(v >> 32) - expect) >= 0) break;
          __builtin_amdgcn_s_sleep(8);
          if (coop_bounded(c, spins)) { v = ((unsigned long long)expect << 32) | MYR_COOP_EXIT; break; }
        }
        // (the wavefronts may have read different commands -- the owner posts on while this workgroup is not counted in: wavefront 0's is the workgroup's)
        unsigned long long* lcmd = reinterpret_cast<unsigned long long*>(c.sMisc) + 5;
        if (c.tid == 0) *lcmd = v;
        __syncthreads();
        v = *lcmd;
        v = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(v >> 32)) << 32) | (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)v);
        const int mode = (int)(v & 255), hv = (int)((v >> 8) & 1), nh = (int)((v >> 12) & 15);
        const unsigned int seq = (unsigned int)(v >> 32);
        COOP_TR(7, seq);
        finished_pass = false;
        if (mode == MYR_COOP_EXIT) break;
        if ((unsigned int)((v >> 16) & MYR_COOP_GEN_MASK) != (c.gen & MYR_COOP_GEN_MASK)) break;      // a command of the owner's NEXT trajectory: this one's EXIT was overwritten
        expect = seq + 1;
        if (nh < c.hidx) continue;           // the owner has not counted this workgroup in yet (it looks once per iteration)
        coop_acquire();
        if (mode == 0 || (mode == 3 && !hv)) { for (int i = c.tid; i < c.n; i += NT) c.z[i] = c.pubx[i]; }
        if (mode == 4) { for (int i = c.tid; i < MLAM * c.N * NS; i += NT) c.sLam[i] = c.publam[i]; }
        __syncthreads();
        NodeMfma64::ArgsT<nd_lds> a;
        a.z = (const nd_lds*)c.z; a.dz = (const nd_lds*)c.z; a.lam = (const nd_lds*)c.sLam; a.pt = (nd_glb*)c.pt;
        a.sF = (nd_lds*)c.sF;
        a.alpha = 0.0; a.h6 = c.h6; a.h8 = c.h8; a.K = c.K; a.N = c.N;
        a.pf_f = PT_F; a.pf_a = PT_A; a.pf_b = PT_B; a.pf_d2 = PT_D2;
        a.t0 = W * c.hidx + c.wave; a.ts = W * (nh + 1);
        a.hb = (nd_glb*)c.hb; a.mb = (nd_glb*)c.mb; a.h_valid = hv;
        if (mode == 0) NodeMfma64::pass<0, nd_lds>((const nd_lds*)c.wl, a, c.lane);
        else if (mode == 3) NodeMfma64::pass<3, nd_lds>((const nd_lds*)c.wl, a, c.lane);
        else NodeMfma64::pass<4, nd_lds>((const nd_lds*)c.wl, a, c.lane);
        if (mode == 0) {
          __syncthreads();
          for (int j = c.tid; j < c.K; j += NT) {
            const int own = ((j >> 4) % a.ts) - W * c.hidx;
            if (own >= 0 && own < W) {
#pragma unroll
              for (int q = 0; q < NS; ++q) c.pubf[j * NS + q] = c.sF[j * NS + q];
            }
          }
        }
        coop_release();
        finished_pass = true;
      }
    } else { (void)c; (void)seen; }
  }
  // a workgroup without a trajectory: attach to the running solve of this XCD with the fewest helpers, serve it to its end, look again -- until every
  // trajectory of the batch is finished
  // (A function of its own with everything it needs BY VALUE: the kernel's context must not escape to memory, and two nested endless loops inlined behind
  // the solver came out of the compiler's structurizer with the helpers' counter update in a guard block they never reached -- tools/dev/exp/exp56.sh.)
  struct HelpArgs {
    int N, K, n, lane, wave, tid; double h6, h8;
    double *z, *sLam, *sF, *wl, *sMisc;
    CoopArgs co; double* scratch; long scratch_stride; int B, nboards;
  };
  __device__ __attribute__((noinline)) static void coop_help(HelpArgs ha) {
    if constexpr (MLP) {
      Ctx c;
      c.N = ha.N; c.K = ha.K; c.n = ha.n; c.lane = ha.lane; c.wave = ha.wave; c.tid = ha.tid; c.h6 = ha.h6; c.h8 = ha.h8;
      c.z = ha.z; c.sLam = ha.sLam; c.sF = ha.sF; c.wl = ha.wl; c.sMisc = ha.sMisc; c.coop_abort = ha.co.abort;
      c.nh = 0; c.hidx = 0; c.board = nullptr; c.h_valid = false;
      const CoopArgs co = ha.co; double* const scratch = ha.scratch; const long scratch_stride = ha.scratch_stride; const int B = ha.B, nboards = ha.nboards;
      int* info = reinterpret_cast<int*>(c.sMisc) + 12;      // sMisc[6], sMisc[7]: owner, helper index, commands seen, quit | generation << 1
      for (;;) {
        if (c.tid == 0) {
          const unsigned int myx = 1u + coop_xcc_id();
          int found = -1, hid = 0, quit = 0; unsigned int seen = 0, spins = 0, gen = 0;
          COOP_TR(8, 0);
          for (;;) {
            int best = -1; unsigned int bestk = 0xffffu, besta = 0;
            for (int sl = (int)(blockIdx.x & 7u); sl < nboards; sl += 8) {        // (workgroup ids are dealt round robin over the 8 XCDs; the board says where its owner really is)
              if (sl == (int)blockIdx.x) continue;
              const unsigned int a = __hip_atomic_load(&co.boards[sl].att, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              const unsigned int k = a & 0xffffu;
              if ((a & MYR_COOP_RUNNING) && k < (unsigned int)co.maxh && k < bestk &&
                  __hip_atomic_load(&co.boards[sl].xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == myx) { best = sl; bestk = k; besta = a; }
            }
            if (best >= 0) {
              // (the commands seen BEFORE attaching: every command that counts this workgroup in is posted after the attach.  The attach word carries the
              //  trajectory's generation: the swap fails when the owner has gone on to its next trajectory since the word was read.)
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              seen = (unsigned int)(__hip_atomic_load(&co.boards[best].cmd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
              unsigned int expected = besta;
              if (__hip_atomic_compare_exchange_strong(&co.boards[best].att, &expected, besta + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                found = best; hid = (int)bestk + 1; gen = (besta >> 16) & MYR_COOP_GEN_MASK; COOP_DBG(16, "[wg %d] attached to %d as helper %d (seen %u)\n", (int)blockIdx.x, best, hid, seen); break;
              }
              continue;
            }
            if (__hip_atomic_load(co.abort + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= B) { quit = 1; COOP_DBG(32, "[wg %d] all finished: quit\n", (int)blockIdx.x); break; }
            __builtin_amdgcn_s_sleep(64);
            if (coop_bounded(c, spins)) { quit = 1; break; }
          }
          info[0] = found; info[1] = hid; info[2] = (int)seen; info[3] = quit | (int)(gen << 1);
          COOP_TR(5, found); COOP_TR(6, hid); COOP_TR(10, quit ? 999 : 0);
        }
        __syncthreads();
        const int found = __builtin_amdgcn_readfirstlane(info[0]), hid = __builtin_amdgcn_readfirstlane(info[1]);
        const unsigned int seen = (unsigned int)__builtin_amdgcn_readfirstlane(info[2]);
        const int quit_gen = __builtin_amdgcn_readfirstlane(info[3]);
        const int quit = quit_gen & 1;
        __syncthreads();
        if (quit) break;
        c.gen = (unsigned int)(quit_gen >> 1);
        double* so = scratch + (long)found * scratch_stride;
        c.pt = so + off_pt(c.N); c.hb = so + off_hb(c.N); c.mb = so + off_mb(c.N);
        c.board = co.boards + found; c.hidx = hid;
        c.pubx = co.pub + (long)found * co.pub_stride; c.publam = c.pubx + c.n; c.pubf = c.publam + MLAM * c.N * NS;
        coop_serve(c, seen);
      }
    } else (void)ha;
  }

  // network systems: one matrix-core pass over all points, tiles of 16 points dealt over the W wavefronts
  template <int MODE>
  __device__ static inline void node_pass(Ctx& c, double alpha) {
    if constexpr (MLP) {
#ifdef MYR_PHASE_TIMING
      const long long tn0_ = clock64();
#endif
      NodeMfma64::ArgsT<nd_lds> a;
      a.z = (const nd_lds*)c.z; a.dz = (const nd_lds*)c.dz; a.lam = (const nd_lds*)c.sLam; a.pt = (nd_glb*)c.pt;
      a.sF = (nd_lds*)c.sF;
      a.alpha = alpha; a.h6 = c.h6; a.h8 = c.h8; a.K = c.K; a.N = c.N;
      a.pf_f = PT_F; a.pf_a = PT_A; a.pf_b = PT_B; a.pf_d2 = PT_D2;
      a.t0 = c.wave; a.ts = W * (c.nh + 1);
      a.hb = (nd_glb*)c.hb; a.mb = (nd_glb*)c.mb; a.h_valid = c.h_valid ? 1 : 0;
      if (c.nh > 0) coop_post<MODE>(c, alpha);
      NodeMfma64::pass<MODE, nd_lds>((const nd_lds*)c.wl, a, c.lane);
      if (c.nh > 0) coop_wait<MODE>(c);
#ifdef MYR_PHASE_TIMING
      if (blockIdx.x == 0 && c.tid == 0) node_tph_[MODE] += clock64() - tn0_;
#endif
    } else { (void)c; (void)alpha; }
  }

  // bounds of the NW variables of point j
  __device__ static inline void load_bounds(const Ctx& c, int j, double* l, double* u) {
    if (c.uni) {
      const nd_lds* b = (const nd_lds*)c.sB + ((j == 0) ? 0 : ((j == c.K - 1) ? 2 * NW : 4 * NW));
#pragma unroll
      for (int q = 0; q < NW; ++q) { l[q] = b[q]; u[q] = b[NW + q]; }
    } else {
#pragma unroll
      for (int q = 0; q < NW; ++q) { const long i = zi(c, j, q); l[q] = c.lb[i]; u[q] = c.ub[i]; }
    }
  }

  struct Step { bool on; double ap, ad, mu, ksig; };
  struct Acc { double f, cmax, cmin, sm, lg; int nm; };
  struct PRec { double x[NS], f[NS], A[NS * NS], B[NS * NU], own[NS]; };   // own = w_j dg/dx + (zU - zL)_x of the point

  // the accepted step applied to the NW variables of point j (network systems: a pass of its own in front of the matrix-core
  // linearisation; same formulas as in lin_at)
  __device__ static inline void step_at(Ctx& c, const Step& st, int j, PRec&) {
    double bl[NW], bu[NW];
    load_bounds(c, j, bl, bu);
    const double iks = 1.0 / st.ksig;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const long i = zi(c, j, q);
      const double l = bl[q], u = bu[q], zv = c.z[i], d = c.dz[i], zl = c.zL[i], zu = c.zU[i];
      const bool fr = l < u;
      const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
      const double zn = fr ? fma(st.ap, d, zv) : zv;      // (an explicit fma: the network passes evaluate the trial point as fma(alpha, dz, z), and the stored activations must belong to THIS point)
      const double sl = hl ? zv - l : 1.0, su = hu ? u - zv : 1.0;
      const double snl = hl ? zn - l : 1.0, snu = hu ? u - zn : 1.0;
      double vl = zl + st.ad * (-zl + (st.mu - zl * d) * detail::rcp_(sl));
      double vu = zu + st.ad * (-zu + (st.mu + zu * d) * detail::rcp_(su));
      const double ml = st.mu * detail::rcp_(snl), mu_ = st.mu * detail::rcp_(snu);
      vl = detail::dmax(detail::dmin(vl, st.ksig * ml), ml * iks);
      vu = detail::dmax(detail::dmin(vu, st.ksig * mu_), mu_ * iks);
      c.z[i] = zn; c.zL[i] = hl ? vl : 0.0; c.zU[i] = hu ? vu : 0.0;
    }
  }

  // One point of the backward phase: the accepted step of the previous iteration is applied on the values loaded anyway
  // (same formulas, same order of operations as HsWave::points_lin), then dynamics + first derivatives, bound sums.
  // Every lane runs it (uniform control flow: the callers use cross-lane moves); `live` gates stores and sums.
  __device__ static inline void lin_at(Ctx& c, const Step& st, int j, bool live, PRec& R, Acc& a) {
    typename S::VarBlk V;
    load_bounds(c, j, V.l, V.u);
    const double iks = 1.0 / st.ksig;
    double dv[NW];
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const long i = zi(c, j, q);
      V.z[q] = c.z[i]; V.zl[q] = c.zL[i]; V.zu[q] = c.zU[i];
      dv[q] = st.on ? c.dz[i] : 0.0;
    }
    if (st.on) {
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const long i = zi(c, j, q);
        const double l = V.l[q], u = V.u[q], zv = V.z[q], d = dv[q], zl = V.zl[q], zu = V.zu[q];
        const bool fr = l < u;
        const bool hl = fr && (l > -INFINITY), hu = fr && (u < INFINITY);
        const double zn = fr ? fma(st.ap, d, zv) : zv;      // (an explicit fma: the network passes evaluate the trial point as fma(alpha, dz, z), and the stored activations must belong to THIS point)
        const double sl = hl ? zv - l : 1.0, su = hu ? u - zv : 1.0;
        const double snl = hl ? zn - l : 1.0, snu = hu ? u - zn : 1.0;
        double vl = zl + st.ad * (-zl + (st.mu - zl * d) * detail::rcp_(sl));
        double vu = zu + st.ad * (-zu + (st.mu + zu * d) * detail::rcp_(su));
        const double ml = st.mu * detail::rcp_(snl), mu_ = st.mu * detail::rcp_(snu);
        vl = detail::dmax(detail::dmin(vl, st.ksig * ml), ml * iks);
        vu = detail::dmax(detail::dmin(vu, st.ksig * mu_), mu_ * iks);
        V.z[q] = zn; V.zl[q] = hl ? vl : 0.0; V.zu[q] = hu ? vu : 0.0;
        if (live) { c.z[i] = V.z[q]; c.zL[i] = V.zl[q]; c.zU[i] = V.zu[q]; }
      }
    }
    double u_[NU], g, gw[NW];
#pragma unroll
    for (int q = 0; q < NS; ++q) R.x[q] = V.z[q];
#pragma unroll
    for (int q = 0; q < NU; ++q) u_[q] = V.z[NS + q];
    set_time<Sys>(c.pp.get(), tq(j, c.h));
    if constexpr (MLP) {          // f, A, B from the matrix-core pass (the step was applied before it); the cost is closed form
      const double* pt = c.pt + j;
      const int K = c.K;
#pragma unroll
      for (int q = 0; q < NS; ++q) R.f[q] = pt[(PT_F + q) * K];
#pragma unroll
      for (int q = 0; q < NS * NS; ++q) R.A[q] = pt[(PT_A + q) * K];
#pragma unroll
      for (int q = 0; q < NS * NU; ++q) R.B[q] = pt[(PT_B + q) * K];
      Sys::cost_grad(R.x, u_, c.pp.get(), &g, gw);
    } else
      Sys::lin(R.x, u_, c.pp.get(), R.f, R.A, R.B, &g, gw);
    const double wj = wq(c.K, j, c.h);
    if (TRAP && j == c.K - 1) fold_terminal<Sys>(R.x, u_, c.pp.get(), wj, g, gw);   // trapezoidal.py:126-127
    double cmax = a.cmax, cmin = a.cmin, sm = 0.0, slk = 1.0; int nm = 0, sexp = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      typename S::BV b = S::bound_terms(V.z[q], V.l[q], V.u[q], V.zl[q], V.zu[q], cmax, cmin);
      if (q < NS) R.own[q < NS ? q : 0] = wj * gw[q] + b.zlu;
      const bool fr = V.l[q] < V.u[q];
      const bool hl = fr && (V.l[q] > -INFINITY), hu = fr && (V.u[q] < INFINITY);
      sm += (hl ? V.zl[q] : 0.0) + (hu ? V.zu[q] : 0.0);
      nm += (hl ? 1 : 0) + (hu ? 1 : 0);
      const double sl = hl ? V.z[q] - V.l[q] : 1.0, su = hu ? V.u[q] - V.z[q] : 1.0;
      { int e_; slk *= frexp((sl > 0.0 ? sl : 1.0) * (su > 0.0 ? su : 1.0), &e_); sexp += e_; }
    }
    if (live) {
      a.cmax = cmax; a.cmin = cmin; a.sm += sm; a.nm += nm;
      a.lg -= log(slk) + sexp * 0.6931471805599453;
      a.f += wj * g;
    }
  }

  // ---- BACKWARD phase -------------------------------------------------------------------------------------------------------
  // Lane l of a block takes stage k = kb + 63 - l (descending: the suffix recursion over the stages becomes the prefix scan
  // over the lanes that the DPP idiom provides, and the end knot of stage k is the start knot of the lane below).  The top
  // block starts at the virtual stage N, whose start knot is the terminal point.
  struct BOut { double f, cmax, cmin, sm, lg, c1, cinf, lam_inf, sum_mult; int nm; };
  __device__ static void backward(Ctx& c, const Step& stp, const double* nuT, BOut& o) {
    using namespace detail;
    const int N = c.N, K = c.K, lane = c.lane;
    const double h6 = c.h6, h8 = c.h8;
    Acc acc{0.0, 0.0, INFINITY, 0.0, 0.0, 0};
    double c1 = 0, cinf = 0, li_ = 0, smu = 0;
    double piS[NS];
#pragma unroll
    for (int q = 0; q < NS; ++q) piS[q] = c.term_pinned[q] ? nuT[q] : 0.0;
    Step stp_ = stp;
    if constexpr (MLP) {          // the network pass reads the iterate: apply the step first (lanes over points), then linearise all points
      if (stp.on) {
        Acc dummy{0.0, 0.0, INFINITY, 0.0, 0.0, 0};
        for (int j = c.tid; j < K; j += NT) { PRec R; step_at(c, stp, j, R); }
        (void)dummy;
        wsync();
        stp_.on = false;
      }
      node_pass<3>(c, 0.0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      wsync();
    }
    // rounds of W blocks from the top: wavefront w takes block r0 - w of the round
    int round = 0;
#ifdef MYR_PHASE_TIMING
    long long bw_t0_ = clock64();
#endif
    for (int r0 = N / 64; r0 >= 0; r0 -= W, ++round) {
      const int blk = r0 - c.wave;
      const int kr = blk >= 0 ? blk * 64 + 63 - lane : N + 1;
      const bool on_s = kr <= N, on = kr < N;
      const int k = on_s ? kr : N;                 // (lanes above the horizon repeat the terminal point; nothing is stored)
      PRec Rs, Rm, Re;
      lin_at(c, stp_, TRAP ? k : 2 * k, on_s, Rs, acc);
      if constexpr (!TRAP) lin_at(c, stp_, on ? 2 * k + 1 : 2 * k, on, Rm, acc);
      // end knot of the stage = start knot of stage k + 1: the lane below; lane 0 takes what the block above left (the
      // wavefront above of this round, or the last wavefront of the previous round)
      {
        double* rs = reinterpret_cast<double*>(&Rs); double* re = reinterpret_cast<double*>(&Re);
        double* mine = c.sStash + ((round & 1) * W + c.wave) * NREC;
        const double* theirs = c.sStash + (c.wave > 0 ? ((round & 1) * W + c.wave - 1) : (((round + 1) & 1) * W + W - 1)) * NREC;
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < NREC; ++q) mine[q] = rs[q];
        }
        wsync();
#pragma unroll
        for (int q = 0; q < NREC; ++q) {
          const double t = lane_up1(rs[q]);
          re[q] = (lane == 0) ? theirs[q] : t;
        }
      }
      MYR_BW_PH(0)     // (phase-timing builds: linearisation of the round's points + the neighbour exchange)
      double MA[NS * NS], Mb[NS], Ld[NS * NS], ld0[NS], Li[NS * NS], li0[NS];
      if constexpr (TRAP) {
        // trapezoidal scheme (TrapCore::backward, os_solver.h): c_k = h/2 (f_k + f_{k+1}) - (x_{k+1} - x_k);
        // E = I - h/2 A_e, E dx_e = (I + h/2 A_s) dx_s + h/2 B_s du_s + h/2 B_e du_e + c_k; adjoint E^T lam_k = own_e + Pi_k,
        // Pi_{k-1} = (I + h/2 A_s)^T lam_k
        const double* xs = Rs.x; const double* fs = Rs.f; const double* As = Rs.A; const double* Bs = Rs.B;
        const double* xe = Re.x; const double* fe = Re.f; const double* Ae = Re.A; const double* Be = Re.B;
        const double hh = 0.5 * c.h;
        double owne[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) owne[q] = (k == N - 1 && c.term_pinned[q]) ? 0.0 : Re.own[q];
        double E[NS * NS], Ge[NS * NY1];
#pragma unroll
        for (int r = 0; r < NS; ++r) {
          const double cj = hh * (fs[r] + fe[r]) - (xe[r] - xs[r]);
          if (on) { c1 += fabs(cj); cinf = dmax(cinf, fabs(cj)); }
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
        if (on) {
          double* sg = c.st + (long)k * SG_N;
#pragma unroll
          for (int q = 0; q < NS * NY1; ++q) sg[SG_GE + q] = Ge[q];
        }
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
          li0[r] = 0.0;
#pragma unroll
          for (int q = 0; q < NS; ++q) Li[r * NS + q] = 0.0;
        }
#pragma unroll
        for (int r = 0; r < NS; ++r) {
#pragma unroll
          for (int q = 0; q <= NS; ++q) {
            double v = 0.0;
#pragma unroll
            for (int t = 0; t < NS; ++t) v += (((r == t) ? 1.0 : 0.0) + hh * As[t * NS + r]) * (q < NS ? Ld[t * NS + q] : ld0[t]);
            if (q < NS) MA[r * NS + q] = on ? v : ((r == q) ? 1.0 : 0.0); else Mb[r] = on ? v : 0.0;
          }
        }
      } else {
      const double* xs = Rs.x; const double* fs = Rs.f; const double* As = Rs.A; const double* Bs = Rs.B;
      const double* xm = Rm.x; const double* fm = Rm.f; const double* Am = Rm.A; const double* Bm = Rm.B;
      const double* xe = Re.x; const double* fe = Re.f; const double* Ae = Re.A; const double* Be = Re.B;
      double dk[NS], ik[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        dk[q] = (xe[q] - xs[q]) - h6 * (fs[q] + 4.0 * fm[q] + fe[q]);
        ik[q] = xm[q] - 0.5 * (xs[q] + xe[q]) - h8 * (fs[q] - fe[q]);
        if (on) {
          c1 += fabs(dk[q]) + fabs(ik[q]);
          cinf = dmax(cinf, dmax(fabs(dk[q]), fabs(ik[q])));
        }
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
      double rm[NS], owne[NS];
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        rm[q] = Rm.own[q];
        owne[q] = (k == N - 1 && c.term_pinned[q]) ? 0.0 : Re.own[q];
      }
      double* sg = c.st + (long)(on ? k : 0) * SG_N;
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
      if (on) {
#pragma unroll
        for (int q = 0; q < NS * NY1; ++q) sg[SG_GE + q] = Ge[q];
      }
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
        if (on) {
#pragma unroll
          for (int q = 0; q <= NY; ++q) sg[SG_GM + r * NY1 + q] = row[q];
        }
      }
      // adjoint maps: lam_d = Ld Pi + ld0, lam_i = Li Pi + li0, Pi_prev = M Pi + v
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double y[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) y[q] = (q == i) ? 1.0 : 0.0;
        lu_solve_t<NS>(E, y);
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
      // Pi_prev = (-I - h6 As^T) lam_d + (-I/2 - h8 As^T) lam_i  ->  the lane's affine map (identity above the horizon)
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
          if (q < NS) MA[r * NS + q] = on ? s : ((r == q) ? 1.0 : 0.0); else Mb[r] = on ? s : 0.0;
        }
      }
      }
      MYR_BW_PH(1)     // (... the stage's elimination maps and adjoint maps)
      affine_prefix_scan_dpp<NS>(MA, Mb);
      double piN[NS];                          // Pi below the round (the next round's carry), by every wavefront alike
#pragma unroll
      for (int q = 0; q < NS; ++q) piN[q] = piS[q];
      if constexpr (W > 1) {      // the blocks above in this round: their totals (lane 63's composite map) carry Pi down to this block
        double* tot = c.sTot + ((round & 1) * W + c.wave) * NTOT;
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < NS * NS; ++q) tot[q] = MA[q];
#pragma unroll
          for (int q = 0; q < NS; ++q) tot[NS * NS + q] = Mb[q];
        }
        wsync();
        double pw[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) pw[q] = piS[q];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const double* tw = c.sTot + ((round & 1) * W + w) * NTOT;
          double t[NS];
#pragma unroll
          for (int r = 0; r < NS; ++r) {
            double v = tw[NS * NS + r];
#pragma unroll
            for (int q = 0; q < NS; ++q) v += tw[r * NS + q] * piN[q];
            t[r] = v;
          }
#pragma unroll
          for (int q = 0; q < NS; ++q) { piN[q] = t[q]; if (w < c.wave) pw[q] = t[q]; }
        }
#pragma unroll
        for (int q = 0; q < NS; ++q) piS[q] = pw[q];      // Pi above THIS wavefront's block
      }
      double lo[NS], pi[NS];
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double v = Mb[r];
#pragma unroll
        for (int q = 0; q < NS; ++q) v += MA[r * NS + q] * piS[q];
        lo[r] = v;                              // Pi_{k-1}
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const double t = lane_up1(lo[q]);       // Pi_k = Pi_{k'-1} of the lane below, whose stage is k' = k + 1
        pi[q] = (lane == 0) ? piS[q] : t;
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) piS[q] = (W > 1) ? piN[q] : __shfl(lo[q], 63, 64);
      // multipliers of the stage
#pragma unroll
      for (int r = 0; r < NS; ++r) {
        double d = ld0[r], i2 = li0[r];
#pragma unroll
        for (int q = 0; q < NS; ++q) { d += Ld[r * NS + q] * pi[q]; i2 += Li[r * NS + q] * pi[q]; }
        if (on) {
          c.sLam[k * NS + r] = d;
          if (!TRAP) c.sLam[N * NS + k * NS + r] = i2;
          li_ = dmax(li_, dmax(fabs(d), fabs(i2)));
          smu += fabs(d) + fabs(i2);
        }
      }
      MYR_BW_PH(2)     // (... the adjoint scan and the multipliers)
    }
    double v[10] = {wv_sum(acc.f), wv_max(acc.cmax), wv_min(acc.cmin), wv_sum(acc.sm), (double)wv_isum(acc.nm),
                    wv_sum(acc.lg), wv_sum(c1), wv_max(cinf), wv_max(li_), wv_sum(smu)};
    const int op[10] = {0, 1, 2, 0, 0, 0, 0, 1, 1, 0};
    wg_combine<10>(c, v, op);
    o.f = v[0]; o.cmax = v[1]; o.cmin = v[2]; o.sm = v[3]; o.nm = (int)v[4]; o.lg = v[5]; o.c1 = v[6]; o.cinf = v[7];
    o.lam_inf = v[8]; o.sum_mult = v[9];
  }

  // ---- HESSIAN phase: lanes over points -- Lagrangian Hessian, gradient columns, control-row stationarity ----------------------
  __device__ static void hessian(Ctx& c, double& stat, double mu_fold = 0.0) {      // (two-level sweep: the "1" column is written as g0 + mu_fold g1)
    const int N = c.N, K = c.K;
    const double h6 = c.h6, h8 = c.h8;
    double st_ = 0;
    if constexpr (MLP) {          // the network's multiplier-contracted second derivatives of all points (reads the multipliers in LDS)
      node_pass<4>(c, 0.0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      wsync();
    }
    // Every lane runs every round (uniform trip count; `live` gates the stores and the stationarity maximum): a divergent loop here ended in a join block
    // into whose prologue -- IN FRONT of the EXEC restore -- this compiler placed spills of values all lanes need afterwards
    for (int j0 = 0; j0 < K; j0 += NT) {
      const bool live = j0 + c.tid < K;
      const int j = live ? j0 + c.tid : K - 1;
      double a[NS];
      if (TRAP) {           // a_j = h/2 (lam_{j-1} + lam_j): TrapCore's mue = mu_c + h/2 lam
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = 0.0;
          if (j >= 1) s += 0.5 * c.h * c.sLam[(j - 1) * NS + q];
          if (j < N) s += 0.5 * c.h * c.sLam[j * NS + q];
          a[q] = s;
        }
      } else if (j & 1) {
        const int k = (j - 1) >> 1;
#pragma unroll
        for (int q = 0; q < NS; ++q) a[q] = -4.0 * h6 * c.sLam[k * NS + q];
      } else {
        const int kL = (j >> 1) - 1, kR = j >> 1;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          double s = 0.0;
          if (kL >= 0) s += -h6 * c.sLam[kL * NS + q] + h8 * c.sLam[N * NS + kL * NS + q];
          if (kR < N) s += -h6 * c.sLam[kR * NS + q] - h8 * c.sLam[N * NS + kR * NS + q];
          a[q] = s;
        }
      }
      const double wj = wq(K, j, c.h);
      typename S::VarBlk V;
      load_bounds(c, j, V.l, V.u);
#pragma unroll
      for (int q = 0; q < NW; ++q) { const long i = zi(c, j, q); V.z[q] = c.z[i]; V.zl[q] = c.zL[i]; V.zu[q] = c.zU[i]; }
      HsPoint<Sys> P;
      set_time<Sys>(c.pp.get(), tq(j, c.h));
      if constexpr (MLP) {
#pragma unroll
        for (int q = 0; q < NS; ++q) P.x[q] = V.z[q];
#pragma unroll
        for (int q = 0; q < NU; ++q) P.u[q] = V.z[NS + q];
        Sys::cost_grad(P.x, P.u, c.pp.get(), &P.g, P.gw);
#pragma unroll
        for (int q = 0; q < NS * NU; ++q) P.B[q] = c.pt[(long)(PT_B + q) * K + j];
      } else
        S::lin_point(V, c.pp.get(), P);
      if (TRAP && j == K - 1) fold_terminal<Sys>(P.x, P.u, c.pp.get(), wj, P.g, P.gw);   // trapezoidal.py:126-127
      double sig[NW], g1v[NW], zlu[NW], cmx = 0.0, cmn = INFINITY;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        typename S::BV b = S::bound_terms(V.z[q], V.l[q], V.u[q], V.zl[q], V.zu[q], cmx, cmn);
        sig[q] = b.sigma; g1v[q] = b.g1; zlu[q] = b.zlu;
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double r = wj * P.gw[NS + u] + zlu[NS + u];
#pragma unroll
        for (int t = 0; t < NS; ++t) r += P.B[t * NU + u] * a[t];
        st_ = live ? detail::dmax(st_, fabs(r)) : st_;
      }
      double Wh[NW * NW];
      if constexpr (MLP) {
        double Wm[NW * (NW + 1) / 2];
#pragma unroll
        for (int q = 0; q < NW * (NW + 1) / 2; ++q) Wm[q] = c.pt[(long)(PT_D2 + q) * K + j];
        Sys::hessian_packed(P.x, P.u, Wm, wj, Wh);
      } else
        Sys::hessian(P.x, P.u, c.pp.get(), P.D2, a, wj, Wh);
      double* hr = c.hr + (long)j * HR_N;
      const bool last = (j == K - 1);
      if (live) {
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        const bool zr = last && r < NS && c.term_pinned[r < NS ? r : 0];
#pragma unroll
        for (int q = r; q < NW; ++q) {
          const bool zq = last && q < NS && c.term_pinned[q < NS ? q : 0];
          hr[HR_H + symidx(r, q)] = (zr || zq) ? 0.0 : (Wh[r * NW + q] + ((r == q) ? sig[r] : 0.0));
        }
        hr[HR_G0 + r] = zr ? 0.0 : (TL ? fma(mu_fold, g1v[r], wj * P.gw[r]) : wj * P.gw[r]);
        hr[HR_G1 + r] = zr ? 0.0 : g1v[r];
      }
      }
    }
    double v[1] = {wv_max(st_)};
    const int op[1] = {1};
    wg_combine<1>(c, v, op);
    stat = v[0];
  }

  // ---- first point of the sweep: dx_0 = 0, eliminate du_0 (HsWave::riccati_first_point on the packed record) ----------------------
  __device__ static int riccati_first_point(Ctx& c, const HsSolveOpts& o, double delta, int nreg) {
    using namespace detail;
    const int lane = c.lane;
    const double* hr = c.hr;   // point 0
    double Puu[NU * NU], ku[NU * NC], pun[NU * NS];
#pragma unroll
    for (int a = 0; a < NU; ++a) {
#pragma unroll
      for (int b = 0; b < NU; ++b) Puu[a * NU + b] = c.sP[(NS + a) * NW + NS + b] + hr[HR_H + symidx(NS + a, NS + b)] + ((a == b) ? delta : 0.0);
#pragma unroll
      for (int cc = 0; cc < NC; ++cc)
        ku[a * NC + cc] = c.sPc[(NS + a) * NC + cc] + (cc == 0 ? hr[HR_G0 + NS + a] : (cc == 1 ? hr[HR_G1 + NS + a] : 0.0));
#pragma unroll
      for (int i = 0; i < NS; ++i) pun[a * NS + i] = ku[a * NC + 2 + i];
    }
    nreg += chol_reg<NU>(Puu, o.reg_floor);
    chol_solve<NU, NC>(Puu, ku);
    wave_sync<true>();
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
    wave_sync<true>();
    return nreg;
  }

  // ---- Riccati sweep on v_mfma_f64_16x16x4_f64: HsWave::riccati_mfma (tile slots, chaining, pivot rule: see there), reading
  // the symmetric-packed point records; the prefetch ring reads PF stages below stage 0 into the padding in front of hr / st ----
  typedef double mfma_d4 __attribute__((ext_vector_type(4)));
  __device__ static int riccati_mfma(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg) {
    using namespace detail;
    const int lane = c.lane, N = c.N;
    const int g = lane >> 4, j = lane & 15;
    const int scol = j < 4 ? (j < NS ? j : -1) : (j < 6 ? NS : -1);
    const int ycol = scol >= 0 ? scol : ((j == 12 || j == 13) ? NS + 1 : ((j == 8 || j == 9) ? NS + 2 : -1));
    const int cc = j == 6 ? 0 : (j == 7 ? 1 : (j == 10 ? 2 : (j == 11 ? 3 : (j == 14 ? 4 : (j == 15 ? 5 : -1)))));
    const int rcc = (cc >= 0 && cc < NC) ? cc : -1;
    const bool rowx = g < NS;
    const double* he = c.hr + (long)(2 * (N - 1) + 2) * HR_N;
    const double* hm = he - HR_N;
    const double* st = c.st + (long)(N - 1) * SG_N;
    auto hsel = [&](const double* rec, int row, bool on) -> const double* {
      if (!on) return c.zr;
      if (scol >= 0) return rec + HR_H + (scol <= row ? scol * NW - scol * (scol - 1) / 2 + (row - scol) : row * NW - row * (row - 1) / 2 + (scol - row));
      if (rcc == 0) return rec + HR_G0 + row;
      if (rcc == 1) return rec + HR_G1 + row;
      return c.zr;
    };
    auto gsel = [&](int off) -> const double* {
      if (!rowx) return c.zr;
      if (ycol >= 0) return st + off + g * NY1 + ycol;
      if (rcc == 0) return st + off + g * NY1 + NY;
      return c.zr;
    };
    const double* ptr[6] = {hsel(he, g, rowx), hsel(he, NS, g < 2), gsel(SG_GE), hsel(hm, g, rowx), hsel(hm, NS, g < 2), gsel(SG_GM)};
    long stp[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) stp[q] = (ptr[q] == c.zr) ? 0 : ((q == 2 || q == 5) ? (long)SG_N : 2L * HR_N);
    const bool pinr = rowx && c.term_pinned[rowx ? g : 0];
    double X0 = (pinr && scol == g) ? o.rho_term - delta : ((pinr && rcc == 2 + g) ? 1.0 : 0.0), X1 = 0.0;
    const double dv0 = (rowx && scol == g) ? delta : 0.0, dv1 = (g < 2 && scol == NS) ? delta : 0.0;
    const double f_a1 = j < 6 ? 1.0 : 0.0;
    const double f_keep = rcc >= 0 ? 1.0 : 0.0;
    const double f_she = (j == 8 || j == 9) ? 1.0 : 0.0, f_shm = (j == 12 || j == 13) ? 1.0 : 0.0;
    const double f_x1 = g < 2 ? 1.0 : 0.0, f_t1 = g == 2 ? 1.0 : 0.0, f_t23 = g >= 2 ? 1.0 : 0.0;
    const bool a3_on = g < 2 && (j < 6 || j == 10 || j == 11 || j == 14 || j == 15);
    const double f_a3m = (a3_on && g == 0) ? -1.0 : 0.0, f_a3e = (a3_on && g == 1) ? -1.0 : 0.0;
    const int k_off = (g == 0 && scol >= 0 && j != 5) ? scol : ((g == 0 && rcc >= 0) ? NQ * NW + rcc : -1);
    const int k_str = k_off < 0 ? 1 : ((scol >= 0) ? NW : NC);
    double* k_ptr = k_off >= 0 ? c.kg + (long)(N - 1) * KSTR + k_off : c.zr + ZR - 2;
    const long k_step = k_off >= 0 ? KSTR : 0;
    double reg_floor = o.reg_floor;
    asm volatile("" : "+v"(reg_floor));
    int nreg = 0;
    const bool abort_u = uniform_if<(MYR_SWEEP_UNIFORM < 0 ? (W > 1) : (MYR_SWEEP_UNIFORM != 0))>(abort_on_reg);
    double in[PF][6];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int q = 0; q < 6; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
    }
    auto mid_part = [&](double n0, double n1, double Gm) -> mfma_d4 {
      n0 += dv0; n1 += dv1;
      const double s0 = W0::dpp_row_shr8(n0), s1 = W0::dpp_row_shr8(n1);
      mfma_d4 C;
      C[0] = fma(s0, f_shm, n0 * f_keep);
      C[1] = fma(s1, f_shm, n1 * f_keep);
      C[2] = 0.0; C[3] = 0.0;
      const mfma_d4 R = __builtin_amdgcn_mfma_f64_16x16x4f64(n0 * f_a1, Gm, C, 0, 0, 0);
      mfma_d4 C2;
      C2[0] = 0.0; C2[1] = 0.0; C2[2] = 0.0; C2[3] = R[1];
      return __builtin_amdgcn_mfma_f64_16x16x4f64(Gm, R[0], C2, 0, 0, 0);
    };
#if MYR_SWEEP_CARRY >= 2
    double QmU = 0.0;
    mfma_d4 Qm;
    {
      double n0 = in[0][3] + dv0, n1 = in[0][4] + dv1;
      const double s0 = W0::dpp_row_shr8(n0), s1 = W0::dpp_row_shr8(n1);
      mfma_d4 C;
      C[0] = fma(s0, f_shm, n0 * f_keep);
      C[1] = fma(s1, f_shm, n1 * f_keep);
      C[2] = 0.0; C[3] = 0.0;
      const mfma_d4 R = __builtin_amdgcn_mfma_f64_16x16x4f64(n0 * f_a1, in[0][5], C, 0, 0, 0);
      Qm = __builtin_amdgcn_mfma_f64_16x16x4f64(in[0][5], R[0], mfma_d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
      QmU = R[1];
    }
#else
    mfma_d4 Qm = mid_part(in[0][3], in[0][4], in[0][5]);
#endif
    mfma_d4 D3 = {X0, X1, 0.0, 0.0};
#if MYR_SWEEP_CARRY
    mfma_d4 D1c = {0.0, 0.0, 0.0, 0.0}, Rmc = {0.0, 0.0, 0.0, 0.0};

#endif
    for (int kb = N - 1; kb >= 0; kb -= PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int k = kb - u;
        if (k < 0) break;
        X0 = D3[0] + (in[u][0] + dv0); X1 = fma(D3[1], f_x1, in[u][1] + dv1);
        const double G = in[u][2];
        const double nn0 = in[(u + 1) % PF][3], nn1 = in[(u + 1) % PF][4], nGm = in[(u + 1) % PF][5];
#pragma unroll
        for (int q = 0; q < 6; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
        const double sh0 = W0::dpp_row_shr4(X0), sh1 = W0::dpp_row_shr4(X1);
#if MYR_SWEEP_CARRY
        // (rows 8..15 of R~ are zero -- the A operand has no such rows -- and stay zero from stage to stage: the previous result IS the next C operand's
        //  upper half, no zero fill)
        mfma_d4 C1 = D1c;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
        D1c = D1;
#else
        mfma_d4 C1;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        C1[2] = 0.0; C1[3] = 0.0;
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
#endif
        mfma_d4 C2;
        C2[0] = Qm[0]; C2[1] = fma(D3[1], f_t1, Qm[1]); C2[2] = fma(D3[2], f_t23, Qm[2]) + D1[1]; C2[3] = fma(D3[3], f_t23, Qm[3]);
#if MYR_SWEEP_CARRY >= 2
        C2[3] += QmU;
#endif
        const mfma_d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(G, D1[0], C2, 0, 0, 0);
        // midpoint part of stage k-1 (independent of the recursion): issued HERE, so that its two dependent products run on
        // the matrix pipe while the vector pipe waits for D2 and computes the gains -- behind D3 they delayed the next stage's D1
        double m0 = nn0 + dv0, m1 = nn1 + dv1;
        const double ms0 = W0::dpp_row_shr8(m0), ms1 = W0::dpp_row_shr8(m1);
#if MYR_SWEEP_CARRY
        mfma_d4 Cm = Rmc;
        Cm[0] = fma(ms0, f_shm, m0 * f_keep);
        Cm[1] = fma(ms1, f_shm, m1 * f_keep);
        const mfma_d4 Rm = __builtin_amdgcn_mfma_f64_16x16x4f64(m0 * f_a1, nGm, Cm, 0, 0, 0);
        Rmc = Rm;
#else
        mfma_d4 Cm;
        Cm[0] = fma(ms0, f_shm, m0 * f_keep);
        Cm[1] = fma(ms1, f_shm, m1 * f_keep);
        Cm[2] = 0.0; Cm[3] = 0.0;
        const mfma_d4 Rm = __builtin_amdgcn_mfma_f64_16x16x4f64(m0 * f_a1, nGm, Cm, 0, 0, 0);
#endif
#if MYR_SWEEP_CARRY >= 2
        // (the control rows of R enter rows 12..15 of Q: added where Q is consumed -- QmU -- instead of through a C operand that is zero but for them)
        Qm = __builtin_amdgcn_mfma_f64_16x16x4f64(nGm, Rm[0], mfma_d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        QmU = Rm[1];
#else
        mfma_d4 Cq;
        Cq[0] = 0.0; Cq[1] = 0.0; Cq[2] = 0.0; Cq[3] = Rm[1];
        Qm = __builtin_amdgcn_mfma_f64_16x16x4f64(nGm, Rm[0], Cq, 0, 0, 0);
#endif
        const double q00 = W0::rdlane(D2[3], 12), q10 = W0::rdlane(D2[2], 12), q11 = W0::rdlane(D2[2], 8);
        const double det = fma(q00, q11, -(q10 * q10));
        const double rdet = fast_rcp(det);
        const double b0 = D2[3], b1 = D2[2];
        double kk0 = fma(q11, b0, -(q10 * b1)) * rdet;
        double kk1 = fma(q00, b1, -(q10 * b0)) * rdet;
        // (the pivot test is wave-uniform -- the pivots come from v_readlane -- but the compiler sees per-lane values: decided in the
        //  vector unit and, in the W > 1 kernels, made a SCALAR branch through readfirstlane, so that no matrix instruction of their sweep
        //  sits inside an EXEC-masked region: v_mfma does not honour EXEC on this part, tools/dev/litmus/mfma_exec.hip)
        const bool rare_ = !(q00 > reg_floor) || !(det > reg_floor * q00);
        if (uniform_if<(MYR_SWEEP_UNIFORM < 0 ? (W > 1) : (MYR_SWEEP_UNIFORM != 0))>(rare_)) {                                       // rare
          const double u00 = q00, u10 = q10, u11 = q11;
          double d0 = u00;
          if (!(d0 > reg_floor)) { d0 = dmax(fabs(d0), reg_floor); ++nreg; }
          const double i0 = fast_rcp(d0);
          const double l10 = u10 * i0;
          double d1 = u11 - l10 * l10 * d0;
          if (!(d1 > reg_floor)) { d1 = dmax(fabs(d1), reg_floor); ++nreg; }
          if (uniform_if<(MYR_SWEEP_UNIFORM < 0 ? (W > 1) : (MYR_SWEEP_UNIFORM != 0))>(nreg > 0) && abort_u) return nreg;
          const double i1 = fast_rcp(d1);
          kk0 = b0; kk1 = b1;
          kk1 -= l10 * kk0;
          kk0 *= i0; kk1 *= i1;
          kk0 -= l10 * kk1;
        }
        k_ptr[0] = kk0; k_ptr[k_str] = kk1;
        k_ptr -= k_step;
        const double A3 = fma(D2[3], f_a3m, D2[2] * f_a3e);
        const double B3 = g == 0 ? kk0 : (g == 1 ? kk1 : 0.0);
        D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, B3, D2, 0, 0, 0);
      }
    }
    X0 = D3[0]; X1 = D3[1];
    const double T1 = D3[1], T2 = D3[2], T3 = D3[3];
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
    wave_sync<true>();
    if (g == 2 && rcc >= 2) c.sTnu[(rcc - 2) * NC + 0] += T1;
    wave_sync<true>();
    return riccati_first_point(c, o, delta, nreg);
  }

  // ---- Two-level sweep, level 1: the stages [k_lo, k_hi) of ONE chunk on the tile of riccati_mfma -----------------------------------------
  // The recursion over the stages is sequential and, in one wavefront, issue-bound (profiles/r05/sweep_issue_bound.md); the W wavefronts a
  // trajectory owns at small batch sizes can only share it if every wavefront starts somewhere.  Chunk c < W - 1 therefore starts at its end knot
  // e from the terminal form  rho/2 |w_e|^2 + nu^T w_e  with the NW multipliers nu of the continuity condition w_e = w_a(chunk c + 1) as
  // right-hand-side COLUMNS -- exactly what the tile already does for the pinned terminal states of the last stage (the nu_T columns; here every
  // row is "pinned", and the control row's multiplier takes the slot of the mu column: the barrier parameter is known before the sweep, so the
  // caller folds g0 + mu g1 into the "1" column -- tl_fold).  The last chunk is the plain sweep down to its first stage.  A chunk returns the
  // symmetric form over (w_a ; theta_c), theta_c = (1, nu_u, nu_x):  P | pc | T  -- T row m = d/d nu_m of the chunk's value = the end knot w_e as an
  // affine function of theta_c (riccati_mfma's terminal-multiplier bookkeeping, with the control's row 7 kept as well) -- in ONE exchange block
  // (the layout of sP | sPc | sTnu | sKu, the control's T row where sKu is); tl_join eliminates the multipliers.  Gains K | kc as before (kc column
  // 1 now multiplies nu_u).  Same pivot rule; no first point.
  __device__ static int riccati_chunk(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg, int k_lo, int k_hi, bool last, double* xo) {
    using namespace detail;
    static_assert(!TL || (NC == NS + 2 && NU == 1), "slot 7 takes the control's multiplier");
    const int lane = c.lane;
    const int g = lane >> 4, j = lane & 15;
    const int scol = j < 4 ? (j < NS ? j : -1) : (j < 6 ? NS : -1);
    const int ycol = scol >= 0 ? scol : ((j == 12 || j == 13) ? NS + 1 : ((j == 8 || j == 9) ? NS + 2 : -1));
    const int cc = j == 6 ? 0 : (j == 7 ? 1 : (j == 10 ? 2 : (j == 11 ? 3 : (j == 14 ? 4 : (j == 15 ? 5 : -1)))));
    const int rcc = (cc >= 0 && cc < NC) ? cc : -1;
    const bool rowx = g < NS;
    const double* he = c.hr + (long)(2 * (k_hi - 1) + 2) * HR_N;
    const double* hm = he - HR_N;
    const double* st = c.st + (long)(k_hi - 1) * SG_N;
    auto hsel = [&](const double* rec, int row, bool on) -> const double* {
      if (!on) return c.zr;
      if (scol >= 0) return rec + HR_H + (scol <= row ? scol * NW - scol * (scol - 1) / 2 + (row - scol) : row * NW - row * (row - 1) / 2 + (scol - row));
      if (rcc == 0) return rec + HR_G0 + row;      // (g0 + mu g1: folded by the caller)
      return c.zr;
    };
    auto gsel = [&](int off) -> const double* {
      if (!rowx) return c.zr;
      if (ycol >= 0) return st + off + g * NY1 + ycol;
      if (rcc == 0) return st + off + g * NY1 + NY;
      return c.zr;
    };
    const double* ptr[6] = {hsel(he, g, rowx), hsel(he, NS, g < 2), gsel(SG_GE), hsel(hm, g, rowx), hsel(hm, NS, g < 2), gsel(SG_GM)};
    long stp[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) stp[q] = (ptr[q] == c.zr) ? 0 : ((q == 2 || q == 5) ? (long)SG_N : 2L * HR_N);
    const bool pinr = rowx && (!last || c.term_pinned[rowx ? g : 0]);
    // (a pinned terminal state takes no inertia correction: rho - delta, the stage adds delta back; an interface knot is an ordinary point whose
    // Hessian record -- with its delta -- is added by the chunk's last stage on top of the terminal form rho)
    const double rho0 = last ? o.rho_term - delta : o.rho_term;
    double X0 = (pinr && scol == g) ? rho0 : ((pinr && rcc == 2 + g) ? 1.0 : 0.0);
    double X1 = (!last && g < 2) ? (scol == NS ? rho0 : (rcc == 1 ? 1.0 : 0.0)) : 0.0;
    const double dv0 = (rowx && scol == g) ? delta : 0.0, dv1 = (g < 2 && scol == NS) ? delta : 0.0;
    const double f_a1 = j < 6 ? 1.0 : 0.0;
    const double f_keep = rcc >= 0 ? 1.0 : 0.0;
    const double f_she = (j == 8 || j == 9) ? 1.0 : 0.0, f_shm = (j == 12 || j == 13) ? 1.0 : 0.0;
    const double f_x1 = g < 2 ? 1.0 : 0.0, f_t1 = g >= 2 ? 1.0 : 0.0, f_t23 = g >= 2 ? 1.0 : 0.0;      // (row 7 -- the control's multiplier -- is carried like row 6)
    const bool a3_on = g < 2 && (j < 6 || j == 7 || j == 10 || j == 11 || j == 14 || j == 15);          // ... and takes the rank-2 update like the rows nu_x
    const double f_a3m = (a3_on && g == 0) ? -1.0 : 0.0, f_a3e = (a3_on && g == 1) ? -1.0 : 0.0;
    const int k_off = (g == 0 && scol >= 0 && j != 5) ? scol : ((g == 0 && rcc >= 0) ? NQ * NW + rcc : -1);
    const int k_str = k_off < 0 ? 1 : ((scol >= 0) ? NW : NC);
    double* k_ptr = k_off >= 0 ? c.kg + (long)(k_hi - 1) * KSTR + k_off : c.zr + ZR - 2;
    const long k_step = k_off >= 0 ? KSTR : 0;
    double reg_floor = o.reg_floor;
    asm volatile("" : "+v"(reg_floor));
    int nreg = 0;
    const bool abort_u = uniform_if<true>(abort_on_reg);
    double in[PF][6];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int q = 0; q < 6; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
    }
    auto mid_part = [&](double n0, double n1, double Gm) -> mfma_d4 {
      n0 += dv0; n1 += dv1;
      const double s0 = W0::dpp_row_shr8(n0), s1 = W0::dpp_row_shr8(n1);
      mfma_d4 C;
      C[0] = fma(s0, f_shm, n0 * f_keep);
      C[1] = fma(s1, f_shm, n1 * f_keep);
      C[2] = 0.0; C[3] = 0.0;
      const mfma_d4 R = __builtin_amdgcn_mfma_f64_16x16x4f64(n0 * f_a1, Gm, C, 0, 0, 0);
      mfma_d4 C2;
      C2[0] = 0.0; C2[1] = 0.0; C2[2] = 0.0; C2[3] = R[1];
      return __builtin_amdgcn_mfma_f64_16x16x4f64(Gm, R[0], C2, 0, 0, 0);
    };
#if MYR_SWEEP_CARRY >= 2
    double QmU = 0.0;
    mfma_d4 Qm;
    {
      double n0 = in[0][3] + dv0, n1 = in[0][4] + dv1;
      const double s0 = W0::dpp_row_shr8(n0), s1 = W0::dpp_row_shr8(n1);
      mfma_d4 C;
      C[0] = fma(s0, f_shm, n0 * f_keep);
      C[1] = fma(s1, f_shm, n1 * f_keep);
      C[2] = 0.0; C[3] = 0.0;
      const mfma_d4 R = __builtin_amdgcn_mfma_f64_16x16x4f64(n0 * f_a1, in[0][5], C, 0, 0, 0);
      Qm = __builtin_amdgcn_mfma_f64_16x16x4f64(in[0][5], R[0], mfma_d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
      QmU = R[1];
    }
#else
    mfma_d4 Qm = mid_part(in[0][3], in[0][4], in[0][5]);
#endif
    mfma_d4 D3 = {X0, X1, 0.0, 0.0};
#if MYR_SWEEP_CARRY
    mfma_d4 D1c = {0.0, 0.0, 0.0, 0.0}, Rmc = {0.0, 0.0, 0.0, 0.0};
#endif
    for (int kb = k_hi - 1; kb >= k_lo; kb -= PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int k = kb - u;
        if (k < k_lo) break;
        X0 = D3[0] + (in[u][0] + dv0); X1 = fma(D3[1], f_x1, in[u][1] + dv1);
        const double G = in[u][2];
        const double nn0 = in[(u + 1) % PF][3], nn1 = in[(u + 1) % PF][4], nGm = in[(u + 1) % PF][5];
#pragma unroll
        for (int q = 0; q < 6; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
        const double sh0 = W0::dpp_row_shr4(X0), sh1 = W0::dpp_row_shr4(X1);
#if MYR_SWEEP_CARRY
        mfma_d4 C1 = D1c;      // (riccati_mfma: the zero upper half is inherited)
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
        D1c = D1;
#else
        mfma_d4 C1;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        C1[2] = 0.0; C1[3] = 0.0;
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
#endif
        mfma_d4 C2;
        C2[0] = Qm[0]; C2[1] = fma(D3[1], f_t1, Qm[1]); C2[2] = fma(D3[2], f_t23, Qm[2]) + D1[1]; C2[3] = fma(D3[3], f_t23, Qm[3]);
#if MYR_SWEEP_CARRY >= 2
        C2[3] += QmU;
#endif
        const mfma_d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(G, D1[0], C2, 0, 0, 0);
        double m0 = nn0 + dv0, m1 = nn1 + dv1;
        const double ms0 = W0::dpp_row_shr8(m0), ms1 = W0::dpp_row_shr8(m1);
#if MYR_SWEEP_CARRY
        mfma_d4 Cm = Rmc;
        Cm[0] = fma(ms0, f_shm, m0 * f_keep);
        Cm[1] = fma(ms1, f_shm, m1 * f_keep);
        const mfma_d4 Rm = __builtin_amdgcn_mfma_f64_16x16x4f64(m0 * f_a1, nGm, Cm, 0, 0, 0);
        Rmc = Rm;
#else
        mfma_d4 Cm;
        Cm[0] = fma(ms0, f_shm, m0 * f_keep);
        Cm[1] = fma(ms1, f_shm, m1 * f_keep);
        Cm[2] = 0.0; Cm[3] = 0.0;
        const mfma_d4 Rm = __builtin_amdgcn_mfma_f64_16x16x4f64(m0 * f_a1, nGm, Cm, 0, 0, 0);
#endif
#if MYR_SWEEP_CARRY >= 2
        Qm = __builtin_amdgcn_mfma_f64_16x16x4f64(nGm, Rm[0], mfma_d4{0.0, 0.0, 0.0, 0.0}, 0, 0, 0);
        QmU = Rm[1];
#else
        mfma_d4 Cq;
        Cq[0] = 0.0; Cq[1] = 0.0; Cq[2] = 0.0; Cq[3] = Rm[1];
        Qm = __builtin_amdgcn_mfma_f64_16x16x4f64(nGm, Rm[0], Cq, 0, 0, 0);
#endif
        const double q00 = W0::rdlane(D2[3], 12), q10 = W0::rdlane(D2[2], 12), q11 = W0::rdlane(D2[2], 8);
        const double det = fma(q00, q11, -(q10 * q10));
        const double rdet = fast_rcp(det);
        const double b0 = D2[3], b1 = D2[2];
        double kk0 = fma(q11, b0, -(q10 * b1)) * rdet;
        double kk1 = fma(q00, b1, -(q10 * b0)) * rdet;
        const bool rare_ = !(q00 > reg_floor) || !(det > reg_floor * q00);
        if (uniform_if<true>(rare_)) {                                       // rare; a scalar branch (no matrix instruction inside an EXEC-masked region)
          const double u00 = q00, u10 = q10, u11 = q11;
          double d0 = u00;
          if (!(d0 > reg_floor)) { d0 = dmax(fabs(d0), reg_floor); ++nreg; }
          const double i0 = fast_rcp(d0);
          const double l10 = u10 * i0;
          double d1 = u11 - l10 * l10 * d0;
          if (!(d1 > reg_floor)) { d1 = dmax(fabs(d1), reg_floor); ++nreg; }
          if (uniform_if<true>(nreg > 0) && abort_u) return nreg;
          const double i1 = fast_rcp(d1);
          kk0 = b0; kk1 = b1;
          kk1 -= l10 * kk0;
          kk0 *= i0; kk1 *= i1;
          kk0 -= l10 * kk1;
        }
        k_ptr[0] = kk0; k_ptr[k_str] = kk1;
        k_ptr -= k_step;
        const double A3 = fma(D2[3], f_a3m, D2[2] * f_a3e);
        const double B3 = g == 0 ? kk0 : (g == 1 ? kk1 : 0.0);
        D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, B3, D2, 0, 0, 0);
      }
    }
    X0 = D3[0]; X1 = D3[1];
    const double T1 = D3[1], T2 = D3[2], T3 = D3[3];
    double* xP = xo; double* xPc = xo + NW * NW; double* xT = xPc + NW * NC;      // T: rows nu_x (NS), then the row of nu_u
    if (scol >= 0 && j != 5) {
      if (rowx) xP[g * NW + scol] = X0;
      if (g == 0) xP[NS * NW + scol] = X1;
    }
    if (rcc >= 0) {
      if (rowx) xPc[g * NC + rcc] = X0;
      if (g == 0) xPc[NS * NC + rcc] = X1;
      if (g >= 2 && g - 2 < NS) xT[(g - 2) * NC + rcc] = T2;
      if (g >= 2 && g < NS) xT[g * NC + rcc] = T3;
      if (g == 3) xT[NS * NC + rcc] = T1;
    }
    wave_sync<true>();
    // the part of T[nu_m]["1"] that the products leave in row "1" (g^T pc', riccati_mfma)
    if (g == 2 && rcc >= 2) xT[(rcc - 2) * NC + 0] += T1;
    if (g == 2 && rcc == 1) xT[NS * NC + 0] += T1;
    wave_sync<true>();
    return nreg;
  }

  // The same sweep for the trapezoidal scheme (HsWave::riccati_mfma_trap): y = (dx_s, du_s, du_e), ONE eliminated control per stage, no
  // midpoint part -- three matrix instructions per stage, the single pivot Q[du_e][du_e] read from lane 8 of register 2; the end point
  // of stage k is point k + 1.
  __device__ static int riccati_mfma_trap(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg) {
    using namespace detail;
    static_assert(!TRAP || NQ == 1, "one control");
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
      if (scol >= 0) return he + HR_H + (scol <= row ? scol * NW - scol * (scol - 1) / 2 + (row - scol) : row * NW - row * (row - 1) / 2 + (scol - row));
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
    double* k_ptr = k_off >= 0 ? c.kg + (long)(N - 1) * KSTR + k_off : c.zr + ZR - 2;
    const long k_step = k_off >= 0 ? KSTR : 0;
    double reg_floor = o.reg_floor;
    asm volatile("" : "+v"(reg_floor));
    int nreg = 0;
    const bool abort_u = uniform_if<(MYR_SWEEP_UNIFORM < 0 ? (W > 1) : (MYR_SWEEP_UNIFORM != 0))>(abort_on_reg);
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
        const double sh0 = W0::dpp_row_shr4(X0), sh1 = W0::dpp_row_shr4(X1);
        mfma_d4 C1;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        C1[2] = 0.0; C1[3] = 0.0;
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
        mfma_d4 C2;
        C2[0] = 0.0; C2[1] = D3[1] * f_t1; C2[2] = fma(D3[2], f_t23, D1[1]); C2[3] = D3[3] * f_t23;
        const mfma_d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(G, D1[0], C2, 0, 0, 0);
        const double q11 = W0::rdlane(D2[2], 8);
        double d = q11;
        const bool rare_ = !(d > reg_floor);
        if (uniform_if<(MYR_SWEEP_UNIFORM < 0 ? (W > 1) : (MYR_SWEEP_UNIFORM != 0))>(rare_)) {                                       // rare (same pivot rule as chol_reg); a scalar branch, see riccati_mfma
          d = dmax(fabs(d), reg_floor); ++nreg;
          if (abort_u) return nreg;
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
    wave_sync<true>();
    if (g == 2 && rcc >= 2) c.sTnu[(rcc - 2) * NC + 0] += T1;
    wave_sync<true>();
    return riccati_first_point(c, o, delta, nreg);
  }
  // ---- The sweep for WIDER stages on the matrix pipe (round 5): two controls (BEARPOPULATIONS, ROCKETLANDING: six states) and the elastic twins
  // (nu + ns controls).  riccati_mfma above lives on one hand-placed 16x16 tile (NS <= 4, one control, the selector rows of G^ done by lane
  // shifts); here a stage is a small block algebra over 16x16 tiles in the accumulator layout of v_mfma_f64_16x16x4_f64 (element (i,j) of a tile
  // in lane 16 (i%4) + j, register i/4), built on two facts about that layout:
  //   * register r of a tile M IS the B operand "rows 4r..4r+3 of M" and ALSO the A operand "columns 4r..4r+3 of M^T": for two tiles in this layout
  //     M^T Y = sum_r mfma(M[r], Y[r]) -- both products of a stage (R~ = P'^T [G^|g^] + [0|pc'],  [Q|qc] = [Qm|qcm] + G^^T R~) chain without moving
  //     a value between lanes (P' is symmetric), and so does the Schur step [P|pc] = Q - Qq^T K with A = the q rows of Q itself;
  //   * rows and columns share ONE slot map: w = (dx_s, du_s) in slots 0..NW-1; the eliminated controls q = (du_m, du_e) in registers of their own
  //     -- slots 8.. of the first tile while w and q fit eight slots each, else a tile of their own (the twins of CARTPOLE and ROCKETLANDING: 10 and
  //     16 of them) -- so that ONE three-swap all-gather over the four lane groups hands every lane the four q entries of its column that a
  //     register holds; the right-hand sides ("1", mu, nu_1..nu_NS) in the slots left over, spilling into a further column tile if need be.
  // The q are eliminated FOUR AT A TIME (one register of q rows): gather, 4x4 pivot block by v_readlane, per-lane L D L^T solve, Schur update of
  // every row that is still alive -- w, the later q, the bookkeeping rows -- by one matrix instruction per tile; the gains of a block then refer
  // to the later q as well, which a back-substitution over the blocks removes (v_readlane of the 4x4 coupling blocks, per-lane products).  No
  // factor of the whole NQ x NQ block is ever held (136 + 16 doubles per lane for ROCKETLANDING's twin in the vector form).  As in riccati_mfma
  // the dual bookkeeping rows (g^T pc' in row "1", -qc_q^T kc in rows nu_i) fall out of the same products as rows that are otherwise unused.
  // Same stage algebra, pivot order and rule, and outputs (gains K | kc per stage; P, pc, Tnu for the first point) as HsWave::riccati, whose
  // column-per-lane vector form these systems ran on up to round 4.  Matrix instructions per Hermite-Simpson stage (counted in the listings):
  // BEARPOPULATIONS 9, ROCKETLANDING 14, CARTPOLE's twin 57, ROCKETLANDING's twin 85; no LDS, no barrier.
  static constexpr bool GEN = !(NU == 1 && NS <= 4);
  static constexpr bool G_BIG = NW > 8 || NQ > 8;
  static constexpr int G_QT = G_BIG ? 1 : 0;                  // column / row tile of the q slots
  static constexpr int G_QS = G_BIG ? 16 : 8;                 // first q slot (w: slots 0 .. NW-1 in front of it)
  static constexpr int G_NB = (NW + 3) / 4, G_NQB = (NQ + 3) / 4, G_NYT = G_QT + 1;
  static constexpr int G_FREE_LO = G_QS - NW, G_FREE_HI = 16 * (G_QT + 1) - G_QS - NQ;
  static constexpr int G_REST = NC - G_FREE_LO - G_FREE_HI;   // right-hand sides in a tile behind the q
  static constexpr int G_CT = G_QT + 1 + (G_REST > 0 ? 1 : 0);
  static constexpr bool GEN_OK = NW <= 14 && NQ <= 16 && (G_BIG ? G_FREE_LO : G_FREE_LO + G_FREE_HI) >= 2 && G_FREE_HI >= 0 && G_REST <= 16;
  static constexpr bool SUPPORTED = !GEN || GEN_OK;      // riccati_mfma / riccati_mfma_trap / riccati_mfma_gen
  __host__ __device__ static constexpr int g_rhs_slot(int cc) {
    return cc < G_FREE_LO ? NW + cc : (cc - G_FREE_LO < G_FREE_HI ? G_QS + NQ + (cc - G_FREE_LO) : 16 * (G_QT + 1) + (cc - G_FREE_LO - G_FREE_HI));
  }
  __host__ __device__ static constexpr int g_slot_cc(int slot) {
    int cc = -1;
    if (slot < NW) cc = -1;
    else if (slot < G_QS) cc = slot - NW;
    else if (slot < G_QS + NQ) cc = -1;
    else if (slot < 16 * (G_QT + 1)) cc = G_FREE_LO + (slot - G_QS - NQ);
    else cc = G_FREE_LO + G_FREE_HI + (slot - 16 * (G_QT + 1));
    return cc < NC ? cc : -1;
  }
  __host__ __device__ static constexpr bool g_map_ok() {      // every right-hand side has a slot of its own, "1" and mu in the first tile, all inside the tiles
    for (int cc = 0; cc < NC; ++cc)
      if (g_slot_cc(g_rhs_slot(cc)) != cc || g_rhs_slot(cc) >= 16 * G_CT) return false;
    return g_rhs_slot(0) < 16 && g_rhs_slot(1) < 16;
  }
  // every lane group <- the values of lane groups 0..3 (same lane within the group): v_permlane32_swap, then v_permlane16_swap twice
  __device__ static inline void gather4(double x, double* o) {
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(x), __double2loint(x), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(x), __double2hiint(x), false, false);
    // [0]: groups (0, 1, 0, 1); [1]: groups (2, 3, 2, 3)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const auto l2 = __builtin_amdgcn_permlane16_swap(lo[h], lo[h], false, false);
      const auto h2 = __builtin_amdgcn_permlane16_swap(hi[h], hi[h], false, false);
      o[2 * h] = __hiloint2double(h2[0], l2[0]);
      o[2 * h + 1] = __hiloint2double(h2[1], l2[1]);
    }
  }
  __device__ static int riccati_mfma_gen(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg) {
    using namespace detail;
    static_assert(GEN_OK && g_map_ok(), "slot map: w in front of the q, '1' and mu in the first column tile");
    constexpr int NB = G_NB, NQB = G_NQB, CT = G_CT, QT = G_QT, NYT = G_NYT;
    constexpr int QL = G_QS & 15;                     // first q lane / row inside its tile
    constexpr int QR = QL / 4;                        // ... and its first register
    constexpr int PFG = G_BIG ? 2 : PF;               // prefetch depth (<= PF: the padding in front of hr / st covers it)
    constexpr int NL1 = NB * (1 + NYT);               // loads per lane and stage half: H (first tile), G (every y tile), NB registers each
    constexpr int NL = (TRAP ? 1 : 2) * NL1;
    constexpr int L_HE = 0, L_GE = NB, L_HM = NL1, L_GM = NL1 + NB;
    constexpr long H_STEP = TRAP ? (long)HR_N : 2L * HR_N;
    const int lane = c.lane, N = c.N;
    const int g = lane >> 4, j = lane & 15;
    // this lane's column in column tile C: a w component, an eliminated control, a right-hand side, or nothing
    int wc[CT], qc[CT], rc[CT], yc[NYT];
#pragma unroll
    for (int C = 0; C < CT; ++C) {
      const int slot = 16 * C + j;
      wc[C] = slot < NW ? slot : -1;
      qc[C] = (slot >= G_QS && slot < G_QS + NQ) ? slot - G_QS : -1;
      rc[C] = g_slot_cc(slot);
    }
#pragma unroll
    for (int C = 0; C < NYT; ++C) yc[C] = wc[C] >= 0 ? wc[C] : (qc[C] >= 0 ? NW + qc[C] : -1);      // component of y
    const double* he = c.hr + (TRAP ? (long)N * HR_N : (long)(2 * (N - 1) + 2) * HR_N);
    const double* hm = he - HR_N;
    const double* st = c.st + (long)(N - 1) * SG_N;
    const double* ptr[NL];
    long stp[NL];
    double gce[NYT][NB], gcm[NYT][NB], mWrow[NB], dv[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      const int row = 4 * r + g;
      auto hsel = [&](const double* rec) -> const double* {
        if (row >= NW) return c.zr;
        if (wc[0] >= 0) return rec + HR_H + symidx(row, wc[0]);
        if (rc[0] == 0) return rec + HR_G0 + row;
        if (rc[0] == 1) return rec + HR_G1 + row;
        return c.zr;
      };
      auto gsel = [&](int off, int C) -> const double* {
        if (row >= NS) return c.zr;
        if (yc[C] >= 0) return st + off + row * NY1 + yc[C];
        if (rc[C] == 0) return st + off + row * NY1 + NY;
        return c.zr;
      };
      ptr[L_HE + r] = hsel(he);
      if constexpr (!TRAP) ptr[L_HM + r] = hsel(hm);
      // selector rows of G^ (row NS + a picks du_e[a]) and of G^m (picks du_m[a]): constants
      const int a = row - NS;
#pragma unroll
      for (int C = 0; C < NYT; ++C) {
        ptr[L_GE + C * NB + r] = gsel(SG_GE, C);
        if constexpr (!TRAP) ptr[L_GM + C * NB + r] = gsel(SG_GM, C);
        gce[C][r] = (a >= 0 && a < NU && qc[C] == (TRAP ? a : NU + a)) ? 1.0 : 0.0;
        gcm[C][r] = (a >= 0 && a < NU && qc[C] == a) ? 1.0 : 0.0;
      }
      mWrow[r] = row < NW ? 1.0 : 0.0;
      dv[r] = (row < NW && wc[0] == row) ? delta : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NL; ++q) stp[q] = (ptr[q] == c.zr) ? 0 : (((q % NL1) < NB) ? H_STEP : (long)SG_N);
    const double mWcol = wc[0] >= 0 ? 1.0 : 0.0;
    double mRhs[CT], mA3[CT], mTrow[CT][4];
#pragma unroll
    for (int C = 0; C < CT; ++C) {
      mRhs[C] = rc[C] >= 0 ? 1.0 : 0.0;
      mA3[C] = (wc[C] >= 0 || rc[C] >= 2) ? -1.0 : 0.0;       // output rows of a Schur step: w, the nu rows of the bookkeeping (and the later q: below)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const int cc = g_slot_cc(16 * C + 4 * r + g); mTrow[C][r] = (cc == 0 || cc >= 2) ? 1.0 : 0.0; }
    }
    // gains: lane group g stores rows g, g + 4, .. of its column
    double* kp[CT][NQB];
    long kstep[CT][NQB];
#pragma unroll
    for (int C = 0; C < CT; ++C)
#pragma unroll
      for (int b = 0; b < NQB; ++b) {
        const int t = 4 * b + g;
        const int off = t >= NQ ? -1 : (wc[C] >= 0 ? t * NW + wc[C] : (rc[C] >= 0 ? NQ * NW + t * NC + rc[C] : -1));
        kp[C][b] = off >= 0 ? c.kg + (long)(N - 1) * KSTR + off : c.zr + ZR - 2;
        kstep[C][b] = off >= 0 ? KSTR : 0;
      }
    double reg_floor = o.reg_floor;
    asm volatile("" : "+v"(reg_floor));
    int nreg = 0;
    mfma_d4 Q[CT][CT];          // the stage's matrix; between stages: [P | pc] in the w rows, the bookkeeping in its rows
#pragma unroll
    for (int R = 0; R < CT; ++R)
#pragma unroll
      for (int C = 0; C < CT; ++C) Q[R][C] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int C = 0; C < CT; ++C)
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const int row = 4 * r + g;
        const bool pin = row < NS && c.term_pinned[row < NS ? row : 0];
        Q[0][C][r] = (pin && wc[C] == row) ? o.rho_term - delta : ((pin && rc[C] == 2 + row) ? 1.0 : 0.0);
      }
    double in[PFG][NL];
#pragma unroll
    for (int u = 0; u < PFG; ++u) {
#pragma unroll
      for (int q = 0; q < NL; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
    }
    // midpoint part of a stage: [Qm | qcm] = G^m^T ((H_m + delta I) [G^m | g^m] + [0 | g0_m g1_m]) -- the y tiles only
    struct MidT { mfma_d4 t[NYT][NYT]; };
    auto mid_part = [&](const double* v) -> MidT {
      MidT M;
      mfma_d4 R[NYT];
      double X[NB], Gm[NYT][NB];
#pragma unroll
      for (int r = 0; r < NB; ++r) X[r] = v[L_HM + r] + dv[r];
#pragma unroll
      for (int C = 0; C < NYT; ++C) {
        R[C] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < NB; ++r) { Gm[C][r] = v[L_GM + C * NB + r] + gcm[C][r]; if (C == 0) R[C][r] = X[r] * mRhs[0]; }
#pragma unroll
        for (int r = 0; r < NB; ++r) R[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[r] * mWcol, Gm[C][r], R[C], 0, 0, 0);
      }
#pragma unroll
      for (int Rr = 0; Rr < NYT; ++Rr)
#pragma unroll
        for (int C = 0; C < NYT; ++C) {
          M.t[Rr][C] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int r = 0; r < NB; ++r) M.t[Rr][C] = __builtin_amdgcn_mfma_f64_16x16x4f64(Gm[Rr][r], R[C][r], M.t[Rr][C], 0, 0, 0);
        }
      return M;
    };
    MidT Qm{};
    if constexpr (!TRAP) Qm = mid_part(in[0]);
    for (int kb = N - 1; kb >= 0; kb -= PFG) {
#pragma unroll
      for (int u = 0; u < PFG; ++u) {
        const int k = kb - u;
        if (k < 0) break;
        double X[CT][NB], G[NYT][NB], nmid[NL];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
          X[0][r] = fma(Q[0][0][r], mWrow[r], in[u][L_HE + r] + dv[r]);
#pragma unroll
          for (int C = 1; C < CT; ++C) X[C][r] = Q[0][C][r] * mWrow[r];
#pragma unroll
          for (int C = 0; C < NYT; ++C) G[C][r] = in[u][L_GE + C * NB + r] + gce[C][r];
        }
        if constexpr (!TRAP) {
#pragma unroll
          for (int q = 0; q < NL; ++q) nmid[q] = in[(u + 1) % PFG][q];
        }
#pragma unroll
        for (int q = 0; q < NL; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
        // R~ = P'^T [G^ | g^] + [0 | pc']: rows w; a column tile behind the y tiles holds right-hand sides only (G^ has no column there)
        mfma_d4 Rt[CT];
#pragma unroll
        for (int C = 0; C < CT; ++C) {
          Rt[C] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int r = 0; r < NB; ++r) Rt[C][r] = X[C][r] * mRhs[C];
          if (C < NYT) {
#pragma unroll
            for (int r = 0; r < NB; ++r) Rt[C] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[0][r] * mWcol, G[C < NYT ? C : 0][r], Rt[C], 0, 0, 0);
          }
        }
        // [Q | qc] = [Qm | qcm] + G^^T R~, on top of the bookkeeping rows carried from the stage above (a row tile behind the y tiles: those only)
#pragma unroll
        for (int R = 0; R < NYT; ++R)
#pragma unroll
          for (int C = 0; C < CT; ++C) {
#pragma unroll
            for (int r = 0; r < 4; ++r) Q[R][C][r] = (!TRAP && C < NYT) ? fma(Q[R][C][r], mTrow[R][r], Qm.t[R][C < NYT ? C : 0][r]) : Q[R][C][r] * mTrow[R][r];
#pragma unroll
            for (int r = 0; r < NB; ++r) Q[R][C] = __builtin_amdgcn_mfma_f64_16x16x4f64(G[R][r], Rt[C][r], Q[R][C], 0, 0, 0);
          }
        // the midpoint part of stage k - 1 does not depend on the recursion: issued here, it runs while the vector pipe solves for the gains
        if constexpr (!TRAP) Qm = mid_part(nmid);
        // eliminate the q four at a time
        double kk[CT][4 * NQB];
#pragma unroll
        for (int b = 0; b < NQB; ++b) {
          const int nbq = (NQ - 4 * b) < 4 ? (NQ - 4 * b) : 4;      // (compile-time after unrolling)
#pragma unroll
          for (int C = 0; C < CT; ++C) gather4(Q[QT][C][QR + b], &kk[C][4 * b]);
          double Lq[16], dinv[4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q <= r; ++q) Lq[r * 4 + q] = (r < nbq) ? W0::rdlane(kk[QT][4 * b + r], QL + 4 * b + q) : (r == q ? 1.0 : 0.0);
          nreg += ldl_reg<4>(Lq, dinv, reg_floor);
          if (nreg > 0 && abort_on_reg) return nreg;
          const double mQ = (qc[QT] >= 4 * (b + 1)) ? -1.0 : 0.0;       // ... and the q rows that are still alive
#pragma unroll
          for (int C = 0; C < CT; ++C) {
#pragma unroll
            for (int e = nbq; e < 4; ++e) kk[C][4 * b + e] = 0.0;
            ldl_solve<4>(Lq, dinv, &kk[C][4 * b]);
            double B3 = 0.0;                      // row 4 b + g of this block's gains (rows past NQ: zero)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < nbq) B3 = (g == e) ? kk[C][4 * b + e] : B3;
#pragma unroll
            for (int R = 0; R < CT; ++R) {
              if (R == QT && b == NQB - 1 && G_FREE_HI == 0 && QT > 0) continue;      // a tile of q rows only, none of them alive
              const double A3 = Q[QT][R][QR + b] * (R == QT ? (mA3[R] + mQ) : mA3[R]);
              Q[R][C] = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, B3, Q[R][C], 0, 0, 0);
            }
          }
        }
        // the gains of block b refer to the q of the blocks behind it: substitute those (final ones) -- K_b <- K_b - K_b[:, q_b'] K_b'
#pragma unroll
        for (int b = NQB - 2; b >= 0; --b) {
          double m[4][4 * NQB];
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 4 * (b + 1); e < NQ; ++e) m[t][e] = W0::rdlane(kk[QT][4 * b + t], QL + e);
#pragma unroll
          for (int C = 0; C < CT; ++C)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              double s = kk[C][4 * b + t];
#pragma unroll
              for (int e = 4 * (b + 1); e < NQ; ++e) s = fma(-m[t][e], kk[C][e], s);
              kk[C][4 * b + t] = s;
            }
        }
#pragma unroll
        for (int C = 0; C < CT; ++C)
#pragma unroll
          for (int b = 0; b < NQB; ++b) {
            double v = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (4 * b + e < NQ) v = (g == e) ? kk[C][4 * b + e] : v;
            kp[C][b][0] = v;
            kp[C][b] -= kstep[C][b];
          }
      }
    }
    // P, pc, Tnu -> LDS for the first point
#pragma unroll
    for (int C = 0; C < CT; ++C) {
#pragma unroll
      for (int r = 0; r < NB; ++r) {
        const int row = 4 * r + g;
        if (row < NW && wc[C] >= 0) c.sP[row * NW + wc[C]] = Q[0][C][r];
        if (row < NW && rc[C] >= 0) c.sPc[row * NC + rc[C]] = Q[0][C][r];
      }
#pragma unroll
      for (int R = 0; R < CT; ++R)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cc = g_slot_cc(16 * R + 4 * r + g);
          if (cc >= 2 && rc[C] >= 0) c.sTnu[(cc - 2) * NC + rc[C]] = Q[R][C][r];
        }
    }
    wave_sync<true>();
    {
      constexpr int s1 = g_rhs_slot(0);
      static_assert(s1 < 16, "row '1' in the first tile");
#pragma unroll
      for (int C = 0; C < CT; ++C)
        if (g == (s1 & 3) && rc[C] >= 2) c.sTnu[(rc[C] - 2) * NC + 0] += Q[0][C][s1 >> 2];
    }
    wave_sync<true>();
    return riccati_first_point(c, o, delta, nreg);
  }
  // The sweep is a function of its own (-DMYR_SWEEP_INLINE folds it back): its register allocation is then independent of the ~20 k
  // instructions around it (the masks of the tile trick stay in VGPRs instead of being re-read from AGPRs every stage; the kernel
  // around it spills 120 values instead of 186).  Arguments by value -- the context itself must not escape into a call, or every pass
  // would find it in scratch (profiles/r03, experiment 1).  Measured: B=256 3.70 -> 3.56 ms, B=4096 15.5 -> 15.0 ms, B=512 (W = 2)
  // 3.37 -> 3.09 ms.
  struct SwArgs { nd_glb *hr, *st, *zr, *kg; nd_lds* xs; int N, lane, pinned, abort; double reg_floor, rho_term, delta; };
  __device__ __attribute__((noinline)) static int sweep_call(SwArgs a) {
    Ctx c;
    c.N = a.N; c.lane = a.lane; c.hr = (double*)a.hr; c.st = (double*)a.st; c.zr = (double*)a.zr;
    use_set(c, (double*)a.kg, (double*)a.xs);
#pragma unroll
    for (int q = 0; q < NS; ++q) c.term_pinned[q] = ((a.pinned >> q) & 1) != 0;
    HsSolveOpts o;
    o.reg_floor = a.reg_floor; o.rho_term = a.rho_term;
    if constexpr (GEN) return riccati_mfma_gen(c, o, a.delta, a.abort != 0);
    else if constexpr (TRAP) return riccati_mfma_trap(c, o, a.delta, a.abort != 0);
    else return riccati_mfma(c, o, a.delta, a.abort != 0);
  }
  __device__ __attribute__((always_inline)) static int sweep_inl(SwArgs a) {
    Ctx c;
    c.N = a.N; c.lane = a.lane; c.hr = (double*)a.hr; c.st = (double*)a.st; c.zr = (double*)a.zr;
    use_set(c, (double*)a.kg, (double*)a.xs);
#pragma unroll
    for (int q = 0; q < NS; ++q) c.term_pinned[q] = ((a.pinned >> q) & 1) != 0;
    HsSolveOpts o;
    o.reg_floor = a.reg_floor; o.rho_term = a.rho_term;
    if constexpr (GEN) return riccati_mfma_gen(c, o, a.delta, a.abort != 0);
    else if constexpr (TRAP) return riccati_mfma_trap(c, o, a.delta, a.abort != 0);
    else return riccati_mfma(c, o, a.delta, a.abort != 0);
  }
  __device__ static inline int sweep(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg) {
#ifndef MYR_SWEEP_INLINE
    // The sweep is a function of its own in every form (SwArgs: address-space-qualified pointers, a frame of its own; W > 1: called by wavefront 0 -- and
    // by wavefront 1 for the speculative rung -- under a branch on the wave-uniform wavefront index, a scalar branch).  History: round 3 had the index as a
    // per-lane value, the inlined sweep then sat inside an EXEC-masked region and results differed from handle to handle; round 4 made the index uniform
    // and kept W > 1 on the inlined form because call + speculative rung together still failed the fresh-handle gate; round 5 found why -- a spill in
    // front of a join block's EXEC restore, nothing to do with the call -- and holds every build with a listing scan and the
    // register-fill gate.  -DMYR_SWEEP_CALL_W=0: round 4's form; -DMYR_W2_QUALIFIED: the inlined body on SwArgs (faults in some instantiations, exp9).
#if !defined(MYR_SWEEP_CALL_W) || MYR_SWEEP_CALL_W
    constexpr bool CALL = true, QUAL = false;      // (the sweep is a call in every form since round 5; -DMYR_SWEEP_CALL_W=0: round 4's inlined sweep for W > 1)
#elif defined(MYR_W2_QUALIFIED)
    constexpr bool CALL = W == 1, QUAL = true;
#else
    constexpr bool CALL = W == 1, QUAL = false;
#endif
    if constexpr (!CALL && !QUAL) {
      if constexpr (GEN) return riccati_mfma_gen(c, o, delta, abort_on_reg);
      else if constexpr (TRAP) return riccati_mfma_trap(c, o, delta, abort_on_reg);
      else return riccati_mfma(c, o, delta, abort_on_reg);
    }
    SwArgs a;
    a.hr = (nd_glb*)c.hr; a.st = (nd_glb*)c.st; a.zr = (nd_glb*)c.zr; a.kg = (nd_glb*)c.kg; a.xs = (nd_lds*)c.sP;
    a.N = c.N; a.lane = c.lane; a.abort = abort_on_reg ? 1 : 0;
    a.pinned = 0;
#pragma unroll
    for (int q = 0; q < NS; ++q) a.pinned |= c.term_pinned[q] ? (1 << q) : 0;
    a.reg_floor = o.reg_floor; a.rho_term = o.rho_term; a.delta = delta;
    if constexpr (CALL) return sweep_call(a);
    else return sweep_inl(a);
#else
    if constexpr (GEN) return riccati_mfma_gen(c, o, delta, abort_on_reg);
    else if constexpr (TRAP) return riccati_mfma_trap(c, o, delta, abort_on_reg);
    else return riccati_mfma(c, o, delta, abort_on_reg);
#endif
  }

  // The same for the trapezoidal scheme: riccati_mfma_trap over the stages [k_lo, k_hi) of one chunk (stage k ends at point k + 1; one eliminated control).
  __device__ static int riccati_chunk_trap(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg, int k_lo, int k_hi, bool last, double* xo) {
    using namespace detail;
    const int lane = c.lane;
    const int g = lane >> 4, j = lane & 15;
    const int scol = j < 4 ? (j < NS ? j : -1) : (j < 6 ? NS : -1);
    const int ycol = scol >= 0 ? scol : ((j == 8 || j == 9) ? NW : -1);
    const int cc = j == 6 ? 0 : (j == 7 ? 1 : (j == 10 ? 2 : (j == 11 ? 3 : (j == 14 ? 4 : (j == 15 ? 5 : -1)))));
    const int rcc = (cc >= 0 && cc < NC) ? cc : -1;
    const bool rowx = g < NS;
    const double* he = c.hr + (long)k_hi * HR_N;
    const double* st = c.st + (long)(k_hi - 1) * SG_N;
    auto hsel = [&](int row, bool on) -> const double* {
      if (!on) return c.zr;
      if (scol >= 0) return he + HR_H + (scol <= row ? scol * NW - scol * (scol - 1) / 2 + (row - scol) : row * NW - row * (row - 1) / 2 + (scol - row));
      if (rcc == 0) return he + HR_G0 + row;      // (g0 + mu g1: folded by the hessian pass / tl_fold)
      return c.zr;
    };
    const double* ptr[3] = {hsel(g, rowx), hsel(NS, g < 2),
                            !rowx ? c.zr : (ycol >= 0 ? st + SG_GE + g * NY1 + ycol : (rcc == 0 ? st + SG_GE + g * NY1 + NY : c.zr))};
    long stp[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) stp[q] = (ptr[q] == c.zr) ? 0 : (q == 2 ? (long)SG_N : (long)HR_N);
    const bool pinr = rowx && (!last || c.term_pinned[rowx ? g : 0]);
    const double rho0 = last ? o.rho_term - delta : o.rho_term;      // (see riccati_chunk)
    const double X0i = (pinr && scol == g) ? rho0 : ((pinr && rcc == 2 + g) ? 1.0 : 0.0);
    const double X1i = (!last && g < 2) ? (scol == NS ? rho0 : (rcc == 1 ? 1.0 : 0.0)) : 0.0;
    const double dv0 = (rowx && scol == g) ? delta : 0.0, dv1 = (g < 2 && scol == NS) ? delta : 0.0;
    const double f_a1 = j < 6 ? 1.0 : 0.0, f_keep = rcc >= 0 ? 1.0 : 0.0, f_she = (j == 8 || j == 9) ? 1.0 : 0.0;
    const double f_x1 = g < 2 ? 1.0 : 0.0, f_t1 = g >= 2 ? 1.0 : 0.0, f_t23 = g >= 2 ? 1.0 : 0.0;
    const double f_a3 = (g == 0 && (j < 6 || j == 7 || j == 10 || j == 11 || j == 14 || j == 15)) ? -1.0 : 0.0;
    const int k_off = (g == 0 && scol >= 0 && j != 5) ? scol : ((g == 0 && rcc >= 0) ? NQ * NW + rcc : -1);
    double* k_ptr = k_off >= 0 ? c.kg + (long)(k_hi - 1) * KSTR + k_off : c.zr + ZR - 2;
    const long k_step = k_off >= 0 ? KSTR : 0;
    double reg_floor = o.reg_floor;
    asm volatile("" : "+v"(reg_floor));
    int nreg = 0;
    const bool abort_u = uniform_if<true>(abort_on_reg);
    double in[PF][3];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
    }
    mfma_d4 D3 = {X0i, X1i, 0.0, 0.0};
    for (int kb = k_hi - 1; kb >= k_lo; kb -= PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int k = kb - u;
        if (k < k_lo) break;
        const double X0 = D3[0] + (in[u][0] + dv0), X1 = fma(D3[1], f_x1, in[u][1] + dv1);
        const double G = in[u][2];
#pragma unroll
        for (int q = 0; q < 3; ++q) { in[u][q] = *ptr[q]; ptr[q] -= stp[q]; }
        const double sh0 = W0::dpp_row_shr4(X0), sh1 = W0::dpp_row_shr4(X1);
        mfma_d4 C1;
        C1[0] = fma(sh0, f_she, X0 * f_keep);
        C1[1] = fma(sh1, f_she, X1 * f_keep);
        C1[2] = 0.0; C1[3] = 0.0;
        const mfma_d4 D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X0 * f_a1, G, C1, 0, 0, 0);
        mfma_d4 C2;
        C2[0] = 0.0; C2[1] = D3[1] * f_t1; C2[2] = fma(D3[2], f_t23, D1[1]); C2[3] = D3[3] * f_t23;
        const mfma_d4 D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(G, D1[0], C2, 0, 0, 0);
        const double q11 = W0::rdlane(D2[2], 8);
        double d = q11;
        const bool rare_ = !(d > reg_floor);
        if (uniform_if<true>(rare_)) {
          d = dmax(fabs(d), reg_floor); ++nreg;
          if (abort_u) return nreg;
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
    double* xP = xo; double* xPc = xo + NW * NW; double* xT = xPc + NW * NC;
    if (scol >= 0 && j != 5) {
      if (rowx) xP[g * NW + scol] = X0;
      if (g == 0) xP[NS * NW + scol] = X1;
    }
    if (rcc >= 0) {
      if (rowx) xPc[g * NC + rcc] = X0;
      if (g == 0) xPc[NS * NC + rcc] = X1;
      if (g >= 2 && g - 2 < NS) xT[(g - 2) * NC + rcc] = T2;
      if (g >= 2 && g < NS) xT[g * NC + rcc] = T3;
      if (g == 3) xT[NS * NC + rcc] = T1;
    }
    wave_sync<true>();
    if (g == 2 && rcc >= 2) xT[(rcc - 2) * NC + 0] += T1;
    if (g == 2 && rcc == 1) xT[NS * NC + 0] += T1;
    wave_sync<true>();
    return nreg;
  }

  // ---- Two-level sweep: wrappers and level 2 ---------------------------------------------------------------------------------------------------
  __host__ __device__ static inline int tl_edge(int N, int ci) { return (int)(((long)ci * N) / NCH); }      // chunk ci = stages [tl_edge(ci), tl_edge(ci + 1))
  __host__ __device__ static inline int tl_chunk(int N, int k) {
    int ci = 0;
#pragma unroll
    for (int w = 1; w < NCH; ++w) ci += (k >= tl_edge(N, w)) ? 1 : 0;
    return ci;
  }
  struct SwArgsC { nd_glb *hr, *st, *zr, *kg; nd_lds* xo; int lane, pinned, abort, k_lo, k_hi, last; double reg_floor, rho_term, delta; };
  __device__ __attribute__((noinline)) static int chunk_call(SwArgsC a) {
    Ctx c;
    c.lane = a.lane; c.hr = (double*)a.hr; c.st = (double*)a.st; c.zr = (double*)a.zr; c.kg = (double*)a.kg;
#pragma unroll
    for (int q = 0; q < NS; ++q) c.term_pinned[q] = ((a.pinned >> q) & 1) != 0;
    HsSolveOpts o;
    o.reg_floor = a.reg_floor; o.rho_term = a.rho_term;
    if constexpr (TL && TRAP) return riccati_chunk_trap(c, o, a.delta, a.abort != 0, a.k_lo, a.k_hi, a.last != 0, (double*)a.xo);
    else if constexpr (TL) return riccati_chunk(c, o, a.delta, a.abort != 0, a.k_lo, a.k_hi, a.last != 0, (double*)a.xo);
    else return 0;
  }
  __device__ static inline int sweep_chunk(Ctx& c, const HsSolveOpts& o, double delta, bool abort_on_reg, int ci, int grp = 0) {
    SwArgsC a;
    a.hr = (nd_glb*)c.hr; a.st = (nd_glb*)c.st; a.zr = (nd_glb*)c.zr; a.kg = (nd_glb*)(grp ? c.kgB : c.kgA); a.xo = (nd_lds*)(c.xA + (grp * NCH + ci) * EXCH);
    a.lane = c.lane; a.abort = abort_on_reg ? 1 : 0;
    a.pinned = 0;
#pragma unroll
    for (int q = 0; q < NS; ++q) a.pinned |= c.term_pinned[q] ? (1 << q) : 0;
    a.k_lo = tl_edge(c.N, ci); a.k_hi = tl_edge(c.N, ci + 1); a.last = (ci == NCH - 1) ? 1 : 0;
    a.reg_floor = o.reg_floor; a.rho_term = o.rho_term; a.delta = delta;
    return chunk_call(a);
  }
  // g0 + mu g1 -> the "1" column of every point record (the mu column's slot carries the control's continuity multiplier in riccati_chunk).  The
  // hessian pass writes it with the barrier parameter it knows; this pass adds dmu g1 in the iterations that change the parameter afterwards.
  __device__ static inline void tl_fold(Ctx& c, double mu) {
    // (uniform trip count, `live` gates the stores: a divergent loop ends in a join block, and this compiler has put spills in front of such a block's
    // EXEC restore -- DESIGN.md section 8.1; the first form of this loop was caught by tools/dev/scan_exec_prologue.py)
    for (int j0 = 0; j0 < c.K; j0 += NT) {
      const bool live = j0 + c.tid < c.K;
      double* hr = c.hr + (long)(live ? j0 + c.tid : c.K - 1) * HR_N;
      double v[NW];
#pragma unroll
      for (int r = 0; r < NW; ++r) v[r] = fma(mu, hr[HR_G1 + r], hr[HR_G0 + r]);
      if (live) {
#pragma unroll
        for (int r = 0; r < NW; ++r) hr[HR_G0 + r] = v[r];
      }
    }
  }
  // Level 2: the interfaces, last first (one wavefront; lane l < 2 NW owns COLUMN l of every product -- the coefficient of w_a[l], resp. of
  // theta_g[l - NW], theta_g = (1, nu_T) --, the NW x NW factors are computed by every lane alike).  Behind interface ci the TRUE value form of the
  // rest of the horizon  1/2 w^T Pt w + w^T pt theta_g + ...  is known (block tb, block layout: nu_T in the columns 2.., column 1 zero).  Chunk ci
  // delivered P, pc = [pc1 | E], T = [t1 | Tnn] for its own multipliers nu: w_e = E^T w_a + t1 + Tnn nu, and nu has to equal the gradient of the
  // true form less the terminal form it was swept with:  nu = M w_e + pt theta_g,  M = Pt - rho I.  With S = -Tnn = L L^T (positive
  // semidefinite: the chunk's pivots were positive) and C = I + L^T M L:
  //     (I - Tnn M)^-1 v = v - L C^-1 L^T M v         (no inverse of S: unreachable directions of w_e stay where they are)
  //     w_e = We [w_a ; theta_g],  nu = Nu [w_a ; theta_g]         (stored for tl_theta)
  //     Pt' = P + E Nu_w,   pt' = [pc1 | 0] + E Nu_t,   Tt'[i] = Tt[i] + pt[:, nu_T i]^T We_t
  // and the reduced Hessian of the whole horizon is positive definite iff the chunks' pivots are positive AND every C is: its Cholesky pivots are
  // counted like the stages' (inertia correction).  tools/dev/twolevel/model.py is this algebra in numpy against the plain recursion.
  struct JnArgs { nd_lds *xb, *jn; int N, lane; double rho, floor_c; };
#ifndef MYR_TL_JOIN_INLINE
#define MYR_TL_JOIN_INLINE 0      // (as a function of its own the join saves and restores 189 callee-saved registers around 700 instructions of work; inlined it measures the same -- tools/dev/exp/exp88.sh: B = 512 180.6 k against 182.2 k -- so the validated form stays)
#endif
#if MYR_TL_JOIN_INLINE
  __device__ __attribute__((always_inline)) static int tl_join(JnArgs a) {
#else
  __device__ __attribute__((noinline)) static int tl_join(JnArgs a) {
#endif
    using namespace detail;
    constexpr int NG = NS + 1, NCOL = NW + NG, NN = NW * NW;
    static_assert(NCOL == 2 * NW && NN <= 64, "stash layout; one element of an NW x NW product per lane");
    const int lane = a.lane;
    const int col = lane < NCOL ? lane : 0;
    const bool isw = col < NW;
    const int gi = isw ? 0 : col - NW;
    const int gcc = gi == 0 ? 0 : gi + 1;                 // block column of theta_g[gi]
    auto mcc = [](int m) { return m < NS ? 2 + m : 1; };  // block column of the multiplier of w component m
    const int el = lane < NN ? lane : 0, er = el / NW, eq = el - er * NW;      // this lane's element of the NW x NW products
    int nreg = 0, tb = NCH - 1;
#pragma unroll 1
    for (int ci = NCH - 2; ci >= 0; --ci) {
      if (tl_edge(a.N, ci + 1) <= tl_edge(a.N, ci)) continue;      // an empty chunk (N < W)
      const nd_lds* Tb = a.xb + tb * EXCH; nd_lds* Bb = a.xb + ci * EXCH; nd_lds* J = a.jn + ci * 4 * NN;
      const nd_lds* Tpc = Tb + NN; const nd_lds* TT = Tpc + NW * NC;
      nd_lds* Bpc = Bb + NN; nd_lds* BT = Bpc + NW * NC;
      nd_lds *sM = J, *sL = J + NN, *sG = J + 2 * NN, *sC = J + 3 * NN;      // (the interface's stash serves as the work space until We | Nu are written)
      // M = sym(Pt) - rho I and S = -sym(Tnn), one element per lane
      {
        const double m = 0.5 * (Tb[er * NW + eq] + Tb[eq * NW + er]) - ((er == eq) ? a.rho : 0.0);
        const double sv = -0.5 * (BT[er * NC + mcc(eq)] + BT[eq * NC + mcc(er)]);
        sM[el] = m; sL[el] = sv;            // (lanes >= NN repeat element 0)
      }
      wave_sync<true>();
      // S = L L^T by every lane alike (the factor is needed whole by every lane; pivots of unreachable directions are floored, their columns then vanish)
      double L[NN];
      {
        double smax = 0.0;
#pragma unroll
        for (int r = 0; r < NW; ++r)
#pragma unroll
          for (int q = 0; q <= r; ++q) { L[r * NW + q] = sL[r * NW + q]; if (r == q) smax = dmax(smax, fabs(L[r * NW + q])); }
        (void)chol_reg<NW>(L, 1e-14 * dmax(smax, 1e-300));
      }
      wave_sync<true>();                  // (every lane has read S)
#pragma unroll
      for (int r = 0; r < NW; ++r)
#pragma unroll
        for (int q = 0; q < NW; ++q) sL[r * NW + q] = (q <= r) ? L[r * NW + q] : 0.0;      // (every lane stores the same values: no lane-0 region)
      wave_sync<true>();
      {                                   // G = L^T M
        double g = 0.0;
#pragma unroll
        for (int r = 0; r < NW; ++r) g += sL[r * NW + er] * sM[r * NW + eq];
        sG[el] = g;
      }
      wave_sync<true>();
      {                                   // C = I + G L
        double cv = (er == eq) ? 1.0 : 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) cv += sG[er * NW + q] * sL[q * NW + eq];
        sC[el] = cv;
      }
      wave_sync<true>();
      double C[NN], dinv[NW];
#pragma unroll
      for (int r = 0; r < NW; ++r)
#pragma unroll
        for (int q = 0; q <= r; ++q) C[r * NW + q] = 0.5 * (sC[r * NW + q] + sC[q * NW + r]);
      nreg += ldl_reg<NW>(C, dinv, a.floor_c);
      // this lane's column
      double v[NW], y[NW], we[NW], nu[NW];
#pragma unroll
      for (int m = 0; m < NW; ++m) {
        const double e = Bpc[col * NC + mcc(m)];          // (w column: E[col][m]; read by every lane, used by the first NW)
        double t = (gi == 0) ? BT[m * NC + 0] : 0.0;
#pragma unroll
        for (int r = 0; r < NW; ++r) t += BT[m * NC + mcc(r)] * Tpc[r * NC + gcc];
        v[m] = isw ? e : t;
      }
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NW; ++q) t += sG[i * NW + q] * v[q];
        y[i] = t;
      }
      ldl_solve<NW>(C, dinv, y);
#pragma unroll
      for (int m = 0; m < NW; ++m) {
        double t = v[m];
#pragma unroll
        for (int i = 0; i <= m; ++i) t -= L[m * NW + i] * y[i];
        we[m] = t;
      }
#pragma unroll
      for (int m = 0; m < NW; ++m) {
        double t = isw ? 0.0 : Tpc[m * NC + gcc];
#pragma unroll
        for (int r = 0; r < NW; ++r) t += sM[m * NW + r] * we[r];
        nu[m] = t;
      }
      // the new true form's column
      double cn[NW], tn[NS];
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        double t = isw ? Bb[r * NW + col] : ((gi == 0) ? Bpc[r * NC + 0] : 0.0);
#pragma unroll
        for (int m = 0; m < NW; ++m) t += Bpc[r * NC + mcc(m)] * nu[m];
        cn[r] = t;
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        double t = TT[i * NC + gcc];
#pragma unroll
        for (int r = 0; r < NW; ++r) t += Tpc[r * NC + 2 + i] * we[r];
        tn[i] = t;
      }
      wave_sync<true>();                 // every lane has read what it needs of the two blocks and of the work space
      if (lane < NCOL) {
#pragma unroll
        for (int m = 0; m < NW; ++m) { J[m * NCOL + col] = we[m]; J[NW * NCOL + m * NCOL + col] = nu[m]; }
        if (isw) {
#pragma unroll
          for (int r = 0; r < NW; ++r) Bb[r * NW + col] = cn[r];
        } else {
#pragma unroll
          for (int r = 0; r < NW; ++r) Bpc[r * NC + gcc] = cn[r];
#pragma unroll
          for (int i = 0; i < NS; ++i) BT[i * NC + gcc] = tn[i];
        }
      } else if (lane == NCOL) {
#pragma unroll
        for (int r = 0; r < NW; ++r) Bpc[r * NC + 1] = 0.0;
#pragma unroll
        for (int i = 0; i < NS; ++i) BT[i * NC + 1] = 0.0;
      }
      wave_sync<true>();
      tb = ci;
    }
    if (tb != 0) {                        // (N < W: chunk 0 is empty) the first point reads block 0
      constexpr int NCP = NN + NW * NC + NS * NC;
      for (int i0 = 0; i0 < NCP; i0 += 64) {      // (uniform trip count)
        const int i = i0 + lane < NCP ? i0 + lane : NCP - 1;
        a.xb[i] = a.xb[tb * EXCH + i];
      }
      wave_sync<true>();
    }
    return nreg;
  }
  // the multipliers of every chunk from the first point's control step and nu_T: forward over the interfaces (one wavefront, every lane alike)
  __device__ static void tl_theta(Ctx& c, const double* thg) {      // thg = (1, 0, nu_T): the last chunk's theta (rung group 0's maps: the ladder leaves the standing rung there)
    constexpr int NCOL = 2 * NW;
    double wa[NW], tg[NS + 1];
#pragma unroll
    for (int q = 0; q < NS; ++q) wa[q] = 0.0;
    { double v = 0.0;
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) v -= c.sKu[cc] * thg[cc];
      wa[NS] = v; }
    tg[0] = 1.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) tg[1 + i] = thg[2 + i];
#pragma unroll 1
    for (int ci = 0; ci < NCH - 1; ++ci) {
      if (tl_edge(c.N, ci + 1) <= tl_edge(c.N, ci)) continue;
      const nd_lds* J = (const nd_lds*)c.sJn + ci * 4 * NW * NW;
      double we[NW], nu[NW];
#pragma unroll
      for (int m = 0; m < NW; ++m) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) { a += J[m * NCOL + i] * wa[i]; b += J[NW * NCOL + m * NCOL + i] * wa[i]; }
#pragma unroll
        for (int g = 0; g <= NS; ++g) { a += J[m * NCOL + NW + g] * tg[g]; b += J[NW * NCOL + m * NCOL + NW + g] * tg[g]; }
        we[m] = a; nu[m] = b;
      }
      double* th = c.sTh + ci * NC;
      th[0] = 1.0; th[1] = nu[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) th[2 + i] = nu[i];
#pragma unroll
      for (int m = 0; m < NW; ++m) wa[m] = we[m];
    }
    double* th = c.sTh + (NCH - 1) * NC;
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) th[cc] = thg[cc];
  }

  __device__ static inline int knot(int k) { return TRAP ? k : 2 * k; }      // the point that starts stage k
  // The forward recursion s_{k+1} = A_k s_k + b_k over stages k0 .. k1 - 1, one wavefront: lane r < NW owns row r (the lanes above repeat the last row), the
  // state is broadcast from the lanes with v_readlane; the rows travel through a ring of FG stages (loaded FG stages ahead: they do not depend on the
  // state).  In: c.dz at knot k0.  Out: c.dz at the knots k0 + 1 .. k1.  (The same treatment of the backward phase's adjoint recursion -- NS x NS maps, 126
  // doubles in the scan at NS = 6 -- was measured SLOWER than its scan, 94.5 -> 97.5 ms per 4096 ROCKETLANDING solves, exp93.sh: not kept.)
  __device__ static void fwd_seq(Ctx& c, int k0, int k1) {
    constexpr int FG = 6;
    const int row = c.lane < NW ? c.lane : NW - 1;
    double x = c.dz[zi(c, knot(k0), row)];
    double ring[FG][NW + 1];
#pragma unroll
    for (int g = 0; g < FG; ++g) {
      const int k = (k0 + g < k1) ? k0 + g : k1 - 1;
      const double* f = c.fw + (long)k * FWS;
#pragma unroll
      for (int j = 0; j < NW; ++j) ring[g][j] = f[row * NW + j];
      ring[g][NW] = f[NW * NW + row];
    }
#pragma unroll 1
    for (int kb = k0; kb < k1; kb += FG) {
#pragma unroll
      for (int g = 0; g < FG; ++g) {
        const int k = kb + g;
        if (k < k1) {       // (wave-uniform)
          double a0 = ring[g][NW], a1 = 0.0;
#pragma unroll
          for (int j = 0; j < NW; j += 2) {
            a0 = fma(ring[g][j], W0::rdlane(x, j), a0);
            if (j + 1 < NW) a1 = fma(ring[g][j + 1], W0::rdlane(x, j + 1), a1);
          }
          x = a0 + a1;
          c.dz[zi(c, knot(k + 1), row)] = x;
          const int kn = (k + FG < k1) ? k + FG : k1 - 1;
          const double* f = c.fw + (long)kn * FWS;
#pragma unroll
          for (int j = 0; j < NW; ++j) ring[g][j] = f[row * NW + j];
          ring[g][NW] = f[NW * NW + row];
        }
      }
    }
  }

  // ---- FORWARD phase: closed-loop maps -> wave scan -> step of the stage's midpoint and end knot -> their step limits ----------
  __device__ static void forward(Ctx& c, const HsSolveOpts& o, double mu, const double* th, typename S::FwdOut& fo) {
    const int N = c.N, K = c.K, lane = c.lane;
    const double tau = detail::dmax(o.tau_min, 1.0 - mu);
    typename S::FwdOut l; l.alpha_p = 1.0; l.alpha_d = 1.0; l.gphi = 0.0;
    // step limits / merit slope of the NW variables of point j with step d; stores the step
    auto apply = [&](int j, const double* d, bool live) {
      typename S::VarBlk V;
      load_bounds(c, j, V.l, V.u);
      double gg, gw[NW];
#pragma unroll
      for (int q = 0; q < NW; ++q) { const long i = zi(c, j, q); V.z[q] = c.z[i]; V.zl[q] = c.zL[i]; V.zu[q] = c.zU[i]; }
      set_time<Sys>(c.pp.get(), tq(j, c.h));
      Sys::cost_grad(V.z, V.z + NS, c.pp.get(), &gg, gw);
      const double wj = wq(K, j, c.h);
      if (TRAP && j == K - 1) fold_terminal<Sys>(V.z, V.z + NS, c.pp.get(), wj, gg, gw);
      typename S::FwdOut t = l;
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        if (live) c.dz[zi(c, j, q)] = d[q];
        S::step_limits(V.z[q], V.l[q], V.u[q], V.zl[q], V.zu[q], d[q], mu, wj * gw[q], tau, t);
      }
      if (live) l = t;
    };
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
    apply(0, s0, c.tid == 0);
    int round = 0;
    for (int base0 = 0; base0 < N; base0 += NT, ++round) {
      const int kr = base0 + c.tid;             // (wavefront w takes block base0 / 64 + w of the round)
      const bool on = kr < N;
      const int k = on ? kr : N - 1;
      const double* Kst = c.kg + (long)k * KSTR;
      const double* sg = c.st + (long)k * SG_N;
      double Kk[NQ * NW], kq[NQ], Ge[NS * NY1], Gm[TRAP ? 1 : NS * NY1];
#pragma unroll
      for (int q = 0; q < NQ * NW; ++q) Kk[q] = Kst[q];
      double thk[NC];                           // two-level sweep: the multipliers of the stage's chunk
#pragma unroll
      for (int cc = 0; cc < NC; ++cc) thk[cc] = th[cc];
      if constexpr (TL) {
        const double* tc = c.sTh + tl_chunk(N, k) * NC;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) thk[cc] = tc[cc];
      }
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        double v = 0.0;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) v += Kst[NQ * NW + t * NC + cc] * thk[cc];
        kq[t] = v;
      }
#pragma unroll
      for (int q = 0; q < NS * NY1; ++q) { Ge[q] = sg[SG_GE + q]; if constexpr (!TRAP) Gm[q] = sg[SG_GM + q]; }
      // closed-loop map of the stage (HsWave::stage_phi), identity beyond the last stage
      double A[NW * NW], b[NW];
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const bool pin = (k == N - 1) && c.term_pinned[i];
#pragma unroll
        for (int q = 0; q < NW; ++q) {
          double v = Ge[i * NY1 + q];
#pragma unroll
          for (int t = 0; t < NQ; ++t) v -= Ge[i * NY1 + NW + t] * Kk[t * NW + q];
          A[i * NW + q] = on ? (pin ? 0.0 : v) : ((i == q) ? 1.0 : 0.0);
        }
        double v = Ge[i * NY1 + NY];
#pragma unroll
        for (int t = 0; t < NQ; ++t) v -= Ge[i * NY1 + NW + t] * kq[t];
        b[i] = (on && !pin) ? v : 0.0;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
#pragma unroll
        for (int q = 0; q < NW; ++q) A[(NS + a) * NW + q] = on ? -Kk[(QE + a) * NW + q] : ((NS + a == q) ? 1.0 : 0.0);
        b[NS + a] = on ? -kq[QE + a] : 0.0;
      }
      double sn[NW], y[NY];
      if constexpr (FSEQ) {
        // wide stages: the maps of the round go to global scratch, wavefront 0 runs the recursion over them (fwd_seq: one row per lane, the state broadcast
        // with v_readlane) and leaves the knots' steps in c.dz; every lane then takes the states on either side of its stage from there
        if (on) {
          double* f = c.fw + (long)k * FWS;
#pragma unroll
          for (int q = 0; q < NW * NW; ++q) f[q] = A[q];
#pragma unroll
          for (int q = 0; q < NW; ++q) f[NW * NW + q] = b[q];
        }
        wsync();
        if (c.wave == 0) fwd_seq(c, base0, base0 + NT < N ? base0 + NT : N);
        wsync();
#pragma unroll
        for (int q = 0; q < NW; ++q) { y[q] = c.dz[zi(c, knot(k), q)]; sn[q] = c.dz[zi(c, knot(k + 1), q)]; }
      } else {
      affine_prefix_scan_dpp<NW>(A, b);
      double sN[NW];                            // state behind the round, by every wavefront alike
#pragma unroll
      for (int q = 0; q < NW; ++q) sN[q] = s0[q];
      if constexpr (W > 1) {
        double* tot = c.sTot + ((round & 1) * W + c.wave) * NTOT;
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < NW * NW; ++q) tot[q] = A[q];
#pragma unroll
          for (int q = 0; q < NW; ++q) tot[NW * NW + q] = b[q];
        }
        wsync();
        double sw[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) sw[q] = s0[q];
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const double* tw = c.sTot + ((round & 1) * W + w) * NTOT;
          double t[NW];
#pragma unroll
          for (int r = 0; r < NW; ++r) {
            double v = tw[NW * NW + r];
#pragma unroll
            for (int q = 0; q < NW; ++q) v += tw[r * NW + q] * sN[q];
            t[r] = v;
          }
#pragma unroll
          for (int q = 0; q < NW; ++q) { sN[q] = t[q]; if (w < c.wave) sw[q] = t[q]; }
        }
#pragma unroll
        for (int q = 0; q < NW; ++q) s0[q] = sw[q];       // state in front of THIS wavefront's block
      }
#pragma unroll
      for (int r = 0; r < NW; ++r) {
        double v = b[r];
#pragma unroll
        for (int q = 0; q < NW; ++q) v += A[r * NW + q] * s0[q];
        sn[r] = v;                               // s_{k+1}
      }
#pragma unroll
      for (int q = 0; q < NW; ++q) {
        const double t = lane_up1(sn[q]);
        y[q] = (lane == 0) ? s0[q] : t;          // s_k
      }
#pragma unroll
      for (int q = 0; q < NW; ++q) s0[q] = (W > 1) ? sN[q] : __shfl(sn[q], 63, 64);
      }
#pragma unroll
      for (int t = 0; t < NQ; ++t) {
        double v = -kq[t];
#pragma unroll
        for (int q = 0; q < NW; ++q) v -= Kk[t * NW + q] * y[q];
        y[NW + t] = v;
      }
      double dm[NW], de[NW];
#pragma unroll
      for (int r = 0; r < NS; ++r) de[r] = sn[r];                           // = Ge y + ge (0 on a pinned terminal state)
#pragma unroll
      for (int a = 0; a < NU; ++a) de[NS + a] = FSEQ ? sn[NS + a] : y[NW + QE + a];      // (FSEQ: the bits the recursion left in c.dz -- the next stage has read them from there)
      if constexpr (TRAP) {       // every point is a knot
        (void)dm;
        apply(k + 1, de, on);
      } else {
#pragma unroll
        for (int r = 0; r < NS; ++r) {
          double vm = Gm[r * NY1 + NY];
#pragma unroll
          for (int q = 0; q < NY; ++q) vm += Gm[r * NY1 + q] * y[q];
          dm[r] = vm;
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) dm[NS + a] = y[NW + a];
        apply(2 * k + 1, dm, on);
        apply(2 * k + 2, de, on);
      }
    }
    double v[3] = {wv_min(l.alpha_p), wv_min(l.alpha_d), wv_sum(l.gphi)};
    const int op[3] = {2, 2, 0};
    wg_combine<3>(c, v, op);
    fo.alpha_p = v[0]; fo.alpha_d = v[1]; fo.gphi = v[2];
  }

  // ---- merit trial at z + alpha dz: lanes over intervals, each evaluating its midpoint and end knot; the start knot's (x, f)
  // come from the lane below (registers), so nothing is staged -- same sums as HsWave::trial ----------------------------------------
  struct TPt { double x[NS], f[NS]; };
  __device__ static inline void trial_point(Ctx& c, int j, double alpha, bool live, TPt& P, double& fa, double& ba, int& bad) {
    double u[NU], bl[NW], bu[NW];
    load_bounds(c, j, bl, bu);
    double slk = 1.0; int sexp = 0, bd = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const long i = zi(c, j, q);
      const double v = c.z[i] + alpha * c.dz[i];
      const double l = bl[q], ub = bu[q];
      const bool fr = l < ub;
      const bool hl = fr && (l > -INFINITY), hu = fr && (ub < INFINITY);
      const double sl = hl ? v - l : 1.0, su = hu ? ub - v : 1.0;
      bd += (sl > 0.0 ? 0 : 1) + (su > 0.0 ? 0 : 1);
      { int e_; slk *= frexp((sl > 0.0 ? sl : 1.0) * (su > 0.0 ? su : 1.0), &e_); sexp += e_; }
      if (q < NS) P.x[q < NS ? q : 0] = v; else u[q - NS] = v;
    }
    if constexpr (MLP) {
#pragma unroll
      for (int q = 0; q < NS; ++q) P.f[q] = c.sF[j * NS + q];
    } else
      Sys::f(P.x, u, c.pp.get(), P.f);
    set_time<Sys>(c.pp.get(), tq(j, c.h));
    double gj = Sys::g(P.x, u, c.pp.get());
    if (TRAP && j == c.K - 1) fold_terminal<Sys>(P.x, u, c.pp.get(), wq(c.K, j, c.h), gj, nullptr);
    if (live) {
      ba -= log(slk) + sexp * 0.6931471805599453;
      fa += wq(c.K, j, c.h) * gj;
      bad += bd;
    }
  }
  __device__ static bool trial(Ctx& c, double alpha, double mu, double& f, double& bar, double& c1) {
    const int N = c.N, lane = c.lane;
    double fa = 0, ba = 0, ca = 0; int bad = 0;
    if constexpr (MLP) {          // f of every trial point by the matrix-core pass -> LDS
      node_pass<0>(c, alpha);
      wsync();
    }
    TPt Pc;                                   // knot in front of the block (uniform)
    trial_point(c, 0, alpha, c.tid == 0, Pc, fa, ba, bad);
    int round = 0;
    for (int base0 = 0; base0 < N; base0 += NT, ++round) {
      const int kr = base0 + c.tid;
      const bool on = kr < N;
      const int k = on ? kr : N - 1;
      TPt Pm, Pe, Ps;
      if constexpr (!TRAP) trial_point(c, 2 * k + 1, alpha, on, Pm, fa, ba, bad);
      trial_point(c, TRAP ? k + 1 : 2 * k + 2, alpha, on, Pe, fa, ba, bad);
      if constexpr (W > 1) {      // the knot in front of this wavefront's block is the end knot of the block below
        double* mine = c.sTr + ((round & 1) * W + c.wave) * 2 * NS;
        if (lane == 63) {
#pragma unroll
          for (int q = 0; q < NS; ++q) { mine[q] = Pe.x[q]; mine[NS + q] = Pe.f[q]; }
        }
        wsync();
        if (c.wave > 0) {
          const double* theirs = c.sTr + ((round & 1) * W + c.wave - 1) * 2 * NS;
#pragma unroll
          for (int q = 0; q < NS; ++q) { Pc.x[q] = theirs[q]; Pc.f[q] = theirs[NS + q]; }
        }
      }
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        const double tx = lane_up1(Pe.x[q]), tf = lane_up1(Pe.f[q]);
        Ps.x[q] = (lane == 0) ? Pc.x[q] : tx; Ps.f[q] = (lane == 0) ? Pc.f[q] : tf;
      }
      if constexpr (W > 1) {
        const double* last = c.sTr + ((round & 1) * W + W - 1) * 2 * NS;
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pc.x[q] = last[q]; Pc.f[q] = last[NS + q]; }
      } else {
#pragma unroll
        for (int q = 0; q < NS; ++q) { Pc.x[q] = __shfl(Pe.x[q], 63, 64); Pc.f[q] = __shfl(Pe.f[q], 63, 64); }
      }
      if (on) {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
          if constexpr (TRAP) ca += fabs(0.5 * c.h * (Ps.f[q] + Pe.f[q]) - (Pe.x[q] - Ps.x[q]));
          else {
            ca += fabs((Pe.x[q] - Ps.x[q]) - c.h6 * (Ps.f[q] + 4.0 * Pm.f[q] + Pe.f[q]));
            ca += fabs(Pm.x[q] - 0.5 * (Ps.x[q] + Pe.x[q]) - c.h8 * (Ps.f[q] - Pe.f[q]));
          }
        }
      }
    }
    double v[4] = {wv_sum(fa), wv_sum(ba), wv_sum(ca), (double)wv_isum(bad)};
    const int op[4] = {0, 0, 0, 0};
    wg_combine<4>(c, v, op);
    f = v[0]; bar = mu * v[1]; c1 = v[2];
    if (v[3] != 0.0) return false;
    if (!detail::finite_(f)) return false;
    if (!detail::finite_(c1)) return false;
    return detail::finite_(bar);
  }

  // ---- the parallel passes as functions of their own (-DMYR_PASS_NOINLINE; measured SLOWER, off): the context travels BY VALUE (a copy
  // on the stack; the caller's context does not escape), results come back by value.  Every pass then compiles without a single spill
  // (backward 256 + 176 registers, hessian 248, forward 256 + 78, trial 248) -- and the launch takes 17.9 instead of 14.8 ms (B=256: 3.96
  // against 3.58): the pointers arrive through scratch as per-lane values, so addresses are formed on the vector pipe and the LDS
  // accesses lose their address space.  Only the sweep, whose inputs are a dozen values, gains from a frame of its own. ----------------
#ifdef MYR_PASS_NOINLINE
#define MYR_PASS_ATTR __attribute__((noinline))
#else
#define MYR_PASS_ATTR inline
#endif
  struct NuT { double v[NS]; };
  __device__ MYR_PASS_ATTR static BOut backward_pass(Ctx c, Step stp, NuT nu) { BOut o; backward(c, stp, nu.v, o); return o; }
  __device__ MYR_PASS_ATTR static double hessian_pass(Ctx c, double mu_fold) { double st; hessian(c, st, mu_fold); return st; }
  struct ThT { double v[NC]; };
  __device__ MYR_PASS_ATTR static typename S::FwdOut forward_pass(Ctx c, HsSolveOpts o, double mu, ThT th) { typename S::FwdOut fo; forward(c, o, mu, th.v, fo); return fo; }
  struct TrialOut { double f, bar, c1; bool ok; };
  __device__ MYR_PASS_ATTR static TrialOut trial_pass(Ctx c, double alpha, double mu) { TrialOut t; t.ok = trial(c, alpha, mu, t.f, t.bar, t.c1); return t; }

  // ---- start: the caller's point pushed inside its bounds (HsWave::init), into LDS; bound table ---------------------------------
  __device__ static void init(Ctx& c, const double* zg) {
    const double k1 = 1e-2, k2 = 1e-2;
    const int K = c.K;
    int same = 1;
    for (int i = c.tid; i < c.n; i += NT) {
      const double l = c.lb[i], u = c.ub[i], v0 = zg[i];
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
      // does every interior point have the bounds of point 1?  (variable i = component q of point j)
      const int j = i < K * NS ? i / NS : (i - K * NS) / NU;
      const int q = i < K * NS ? i - j * NS : NS + (i - K * NS) - j * NU;
      if (j >= 1 && j <= K - 2) {
        const long i1 = S::zi(K, 1, q);
        const double l1 = c.lb[i1], u1 = c.ub[i1];
        same &= ((l == l1) && (u == u1)) ? 1 : 0;      // (infinities compare equal; a NaN bound switches the table off)
      }
    }
    double sv[1] = {(double)wv_isum(same)};
    const int sop[1] = {0};
    wg_combine<1>(c, sv, sop);
    same = (sv[0] == (double)NT) ? 1 : 0;
    c.uni = __builtin_amdgcn_readfirstlane(same) != 0;
    if (c.tid < NW) {
      const int q = c.tid;
      const long i0 = S::zi(K, 0, q), iT = S::zi(K, K - 1, q), i1 = S::zi(K, 1, q);
      c.sB[q] = c.lb[i0]; c.sB[NW + q] = c.ub[i0];
      c.sB[2 * NW + q] = c.lb[iT]; c.sB[3 * NW + q] = c.ub[iT];
      c.sB[4 * NW + q] = c.lb[i1]; c.sB[5 * NW + q] = c.ub[i1];
    }
    if (c.tid < ZR) c.zr[c.tid] = 0.0;
  }

  // ---- the solve (control flow identical to HsWave::solve / HsSolver::solve) -----------------------------------------------------
  __device__ static void solve(Ctx& c, const HsSolveOpts& o, const double* zg, HsSolveResult& res, int park_mode = 0, int park_k1 = 0,
                               double* sv = nullptr) {
    using namespace detail;
    constexpr int NMMAX = 8;
    static_assert(17 + NS + NMMAX + 5 <= NSCAL, "record of a parked trajectory");
    double mu = o.mu_init, pen = 1.0;
    int pen_over = 0, pen_cuts = 0;
    double nuT[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) nuT[i] = 0.0;
    int stall = 0, small_steps = 0;
    double delta_last = 0.0, lm = 0.0;
    double hist[NMMAX]; int nhist = 0, hpos = 0; double hist_mu = -1.0, hist_pen = -1.0;
#pragma unroll
    for (int i = 0; i < NMMAX; ++i) hist[i] = 0.0;
    Step pending{false, 0.0, 0.0, 0.0, o.kappa_sigma};
    int it0 = 0;
    res.cost = 0.0; res.feas = 0.0; res.stat = 0.0; res.compl_ = 0.0;
    c.h_valid = false;                      // a new or resumed trajectory: the activation store belongs to the slot's previous one
    if (park_mode == 2) {
      // resume a parked trajectory: the solver's LDS as it was at the top of iteration k1, the scalars of the loop, the slot's block of zeros
      const int nl = lds_solver_doubles(c.N);
      for (int i = c.tid; i < nl; i += NT) c.z[i] = __builtin_nontemporal_load(&sv[NSCAL + i]);     // (read once: keep it out of the caches the next kernels use)
      if constexpr (ZLU_GLOBAL) {
        for (int i = c.tid; i < 2 * c.n; i += NT) c.zL[i] = __builtin_nontemporal_load(&sv[NSCAL + nl + i]);      // (zU follows zL)
      }
      if (c.tid < ZR) c.zr[c.tid] = 0.0;
      mu = sv[0]; pen = sv[1]; pen_over = (int)sv[2]; pen_cuts = (int)sv[3]; stall = (int)sv[4]; small_steps = (int)sv[5];
      delta_last = sv[6]; lm = sv[7]; nhist = (int)sv[8]; hpos = (int)sv[9]; hist_mu = sv[10]; hist_pen = sv[11];
      pending.on = sv[12] != 0.0; pending.ap = sv[13]; pending.ad = sv[14]; pending.mu = sv[15]; pending.ksig = sv[16];
#pragma unroll
      for (int i = 0; i < NS; ++i) nuT[i] = sv[17 + i];
#pragma unroll
      for (int i = 0; i < NMMAX; ++i) hist[i] = sv[17 + NS + i];
      c.uni = sv[17 + NS + NMMAX] != 0.0;
      res.cost = sv[18 + NS + NMMAX]; res.feas = sv[19 + NS + NMMAX]; res.stat = sv[20 + NS + NMMAX]; res.compl_ = sv[21 + NS + NMMAX];
      it0 = park_k1;
    } else
      init(c, zg);
    wsync();
#pragma unroll
    for (int q = 0; q < NS; ++q) c.term_pinned[q] = !(c.sB[2 * NW + q] < c.sB[3 * NW + q]);
    const double mu_min = dmin(o.tol_compl, o.tol_stat) * 0.1;
    res.status = 1; res.iters = o.max_iter;
    for (int it = it0; it <= o.max_iter; ++it) {
      if (park_mode == 1 && it == park_k1) {
        // park: everything the rest of the solve needs (the previous iteration ended with a workgroup barrier)
        const int nl = lds_solver_doubles(c.N);
        for (int i = c.tid; i < nl; i += NT) __builtin_nontemporal_store(c.z[i], &sv[NSCAL + i]);
        if constexpr (ZLU_GLOBAL) {
          for (int i = c.tid; i < 2 * c.n; i += NT) __builtin_nontemporal_store(c.zL[i], &sv[NSCAL + nl + i]);
        }
        if (c.tid == 0) {
          sv[0] = mu; sv[1] = pen; sv[2] = (double)pen_over; sv[3] = (double)pen_cuts; sv[4] = (double)stall; sv[5] = (double)small_steps;
          sv[6] = delta_last; sv[7] = lm; sv[8] = (double)nhist; sv[9] = (double)hpos; sv[10] = hist_mu; sv[11] = hist_pen;
          sv[12] = pending.on ? 1.0 : 0.0; sv[13] = pending.ap; sv[14] = pending.ad; sv[15] = pending.mu; sv[16] = pending.ksig;
#pragma unroll
          for (int i = 0; i < NS; ++i) sv[17 + i] = nuT[i];
#pragma unroll
          for (int i = 0; i < NMMAX; ++i) sv[17 + NS + i] = hist[i];
          sv[17 + NS + NMMAX] = c.uni ? 1.0 : 0.0;
          sv[18 + NS + NMMAX] = res.cost; sv[19 + NS + NMMAX] = res.feas; sv[20 + NS + NMMAX] = res.stat; sv[21 + NS + NMMAX] = res.compl_;
        }
        res.status = MYR_STATUS_PARKED_; res.iters = it;
        return;
      }
      if constexpr (MLP) { if (c.board) coop_refresh(c); }      // helper workgroups attached since the last iteration?
      BOut p1;
      { NuT nu_; for (int q = 0; q < NS; ++q) nu_.v[q] = nuT[q]; p1 = backward_pass(c, pending, nu_); }
      pending.on = false;
      c.h_valid = false;                    // (from here on the store holds the activations of THIS iterate, whichever pass wrote them; the flag is for the next linearisation)
      wsync();
      MYR_PH(0)
      double stat_raw;
      const double mu_hess = mu;              // (two-level sweep: the records' "1" column holds g0 + mu_hess g1)
      stat_raw = hessian_pass(c, mu);
      wsync();
      MYR_PH(4)
      const double c1 = p1.c1, cinf = p1.cinf, sum_mult = p1.sum_mult;
      double delta = lm;
      if (o.delta_warm && delta_last > o.delta_warm_min) delta = dmax(delta, delta_last / DELTA_WARM_DIV);
      int nreg = 0;
      auto next_delta = [&](double d) {       // the inertia-correction ladder (IPOPT's: first 1e-4 or a third of the last one, then x 100 / x 8)
        return d == 0.0 ? ((delta_last > 0.0) ? dmax(1e-8, delta_last / 3.0) : 1e-4) : d * ((delta_last > 0.0) ? 8.0 : 100.0);
      };
      // (Two-level sweep: the barrier parameter is folded into the sweep's "1" column, so the tests and the barrier update come BEFORE the sweep -- they
      // depend on the two passes above only.  -DMYR_EARLY_EXIT=1 gives every form this order: the converged iteration then returns WITHOUT its sweep, one
      // sweep in twenty-one of a headline solve, same bits.  Round 5's order -- the last iteration's sweep run and dropped -- is the default for the other forms.)
      constexpr bool EARLY = TL || (MYR_EARLY_EXIT != 0);
      if constexpr (EARLY) {
        const int nm = MLAM * c.N * NS + p1.nm;
        const double sd = nm > 0 ? dmax(1.0, (sum_mult + p1.sm) / nm / 100.0) : 1.0;
        const double stat = stat_raw / sd, comp = p1.cmax / sd;
        res.cost = p1.f; res.feas = cinf; res.stat = stat; res.compl_ = comp;
        if (__builtin_amdgcn_readfirstlane((int)!(finite_(p1.f) && finite_(cinf) && finite_(stat_raw)))) { res.status = 2; res.iters = it; return; }
        if (__builtin_amdgcn_readfirstlane((int)(cinf <= o.tol_feas && stat <= o.tol_stat && comp <= o.tol_compl))) { res.status = 0; res.iters = it; return; }
        if (it == o.max_iter) break;
        for (int guard = 0; guard < 8; ++guard) {
          const double cerr = (p1.cmin <= p1.cmax) ? dmax(fabs(p1.cmax - mu), fabs(p1.cmin - mu)) : 0.0;
          const double emu = dmax(dmax(stat, cinf), cerr / sd);
          if (emu <= o.kappa_eps * mu && mu > mu_min) {
            const double nmu = dmax(mu_min, dmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));
            mu = nmu;
          } else break;
        }
      }
      // (Speculative second rung, -DMYR_FUSED_SPEC -- the W > 1 forms without a two-level sweep: trapezoidal, the block sweep of the wider systems.
      // Round 4 saw TIMBERHARVEST N = 6 take another path from handle to handle in this form; round 5 traced it to a spill the compiler placed in front of
      // a join block's EXEC restore, removed the two sites, and holds every build with tools/dev/scan_exec_prologue.py -- per-unit
      // fallback to -DMYR_FUSED_SPEC=0 -- and the register-fill gate of tests/test_gpu_poison.py.)
#if !defined(MYR_FUSED_SPEC) || MYR_FUSED_SPEC
      constexpr bool SPEC = W > 1;          // (on since round 5: the build guard tools/dev/scan_exec_prologue.py and the register-fill gate hold it; -DMYR_FUSED_SPEC=0 switches it off)
#else
      constexpr bool SPEC = false;
#endif
      if constexpr (TL) {
        // Two-level sweep: every wavefront condenses its chunk of the horizon (riccati_chunk), wavefront 0 joins the chunks at their interfaces
        // (tl_join) and eliminates the first point.  One rung of the inertia ladder = W chunk sweeps side by side + the join; a rung fails when a
        // stage pivot of any chunk, a pivot of an interface or the first point's is not positive.  (The speculative second rung of round 5 is gone:
        // wavefront 1 has its own chunk to sweep.)
        if (__builtin_amdgcn_readfirstlane((int)(mu != mu_hess))) {      // the barrier parameter moved after the hessian pass wrote g0 + mu g1: add the difference
          tl_fold(c, mu - mu_hess);
          wsync();
        }
        MYR_PH(3)
        if constexpr (TLS) {
          // two chunks x two rungs: wavefront w sweeps chunk (w & 1) of rung group (w >> 1); wavefronts 0 and 2 join their group's chunks and eliminate the first point
          for (int tr_ = 0; tr_ < 12; tr_ += 2) {
            const double delta_b = next_delta(delta);
            const bool abort_a = (tr_ < 11) && !(delta > 1e8), abort_b = (tr_ + 1 < 11) && !(delta_b > 1e8);
            const int grp = c.wave >> 1, ci = c.wave & 1;
            const double my_delta = grp ? delta_b : delta;
            const bool my_abort = grp ? abort_b : abort_a;
            double* cnt = c.sTh + NGR * TL_TH1 + grp * (NCH + 1);      // this group's pivot counts: the chunks', the join's
            int nr = 0;
            if (tl_edge(c.N, ci + 1) > tl_edge(c.N, ci)) nr = sweep_chunk(c, o, my_delta, my_abort, ci, grp);
            cnt[ci] = (double)nr;
            MYR_PH(13)
            wsync();
            MYR_PH(5)
            const int nsg = __builtin_amdgcn_readfirstlane((int)cnt[0] + (int)cnt[1]);
            if (ci == 0) {
              int nj = 0;
              if (nsg == 0 || !my_abort) {
                Ctx cw = c;
                use_set(cw, grp ? c.kgB : c.kgA, c.xA + grp * NCH * EXCH);
                JnArgs ja;
                ja.xb = (nd_lds*)(c.xA + grp * NCH * EXCH); ja.jn = (nd_lds*)(c.sJn + grp * TL_JN1); ja.N = c.N; ja.lane = c.lane; ja.rho = o.rho_term; ja.floor_c = MYR_TL_FLOOR;
                nj = tl_join(ja);
                nj = riccati_first_point(cw, o, my_delta, nj);
              }
              cnt[NCH] = (double)nj;
            }
            wsync();
            const double* cn0 = c.sTh + NGR * TL_TH1;
            const int na = __builtin_amdgcn_readfirstlane((int)cn0[0] + (int)cn0[1] + (int)cn0[NCH]);
            const int nb = __builtin_amdgcn_readfirstlane((int)cn0[NCH + 1] + (int)cn0[NCH + 2] + (int)cn0[2 * NCH + 1]);
            wsync();
            MYR_PH(6)
            nreg = na;
            if (na == 0 || !abort_a) break;
            delta = delta_b; nreg = nb;
            if (nb == 0 || !abort_b) {        // the speculative rung stands: its gains, its first block and its interface maps become rung group 0's
              for (int i = c.tid; i < c.N * KST; i += NT) c.kgA[i] = c.kgB[i];
              for (int i = c.tid; i < EXCH; i += NT) c.xA[i] = c.xA[NCH * EXCH + i];
              for (int i = c.tid; i < TL_JN1; i += NT) c.sJn[i] = c.sJn[TL_JN1 + i];
              wsync();
              break;
            }
            delta = next_delta(delta_b);
          }
        } else {
        double* cnt = c.sTh + NGR * TL_TH1;         // pivot counts: the chunks', the join's
        for (int tr_ = 0; tr_ < 12; ++tr_) {
          const bool abort_on_reg = (tr_ < 11) && !(delta > 1e8);
          int nr = 0;
          if (tl_edge(c.N, c.wave + 1) > tl_edge(c.N, c.wave)) nr = sweep_chunk(c, o, delta, abort_on_reg, c.wave);
          cnt[c.wave] = (double)nr;             // (every lane: the count is wave-uniform)
          MYR_PH(13)
          wsync();
          MYR_PH(5)
          int ns = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) ns += (int)cnt[w];
          ns = __builtin_amdgcn_readfirstlane(ns);
          if (ns == 0 || !abort_on_reg) {       // (the same in every wavefront)
            if (c.wave == 0) {
              JnArgs ja;
              ja.xb = (nd_lds*)c.xA; ja.jn = (nd_lds*)c.sJn; ja.N = c.N; ja.lane = c.lane; ja.rho = o.rho_term; ja.floor_c = MYR_TL_FLOOR;
              int nj = tl_join(ja);
              MYR_PH(9)
              nj = riccati_first_point(c, o, delta, nj);
              cnt[W] = (double)nj;
              MYR_PH(10)
            }
            wsync();
            ns += (int)cnt[W];
          }
          nreg = __builtin_amdgcn_readfirstlane(ns);
#ifdef MYR_TRACE
          if (c.tid == 0 && c.traj < MYR_TRACE) {
            printf("L b%d it%d rung %d delta=%.9g chunks", c.traj, it, tr_, delta);
            for (int w = 0; w < W; ++w) printf(" %d", (int)cnt[w]);
            printf(" join+first %d\n", (ns == nreg && ((int)cnt[0] + (W > 1 ? (int)cnt[1] : 0)) == 0) ? (int)cnt[W] : -1);
          }
#endif
          wsync();
          MYR_PH(6)
          if (nreg == 0 || !abort_on_reg) break;
          delta = next_delta(delta);
        }
        }
      } else
      if constexpr (SPEC) {
        // Two rungs of the ladder at a time: wavefront 0 sweeps with delta, wavefront 1 -- idle otherwise -- with the NEXT candidate
        // into a second set of outputs.  A failed first sweep costs a whole sweep (the bad pivot shows up near stage 0: 5 of 22
        // sweeps of a typical solve); its successor is then already there.  Same sequence of candidates, first success wins: the
        // iterates are those of the one-wavefront form.
        use_set(c, c.kgA, c.xA);
        for (int tr_ = 0; tr_ < 12; tr_ += 2) {
          const double delta_b = next_delta(delta);
          const bool abort_a = (tr_ < 11) && !(delta > 1e8), abort_b = (tr_ + 1 < 11) && !(delta_b > 1e8);
          {
            Ctx cw = c;
            if (c.wave == 1) use_set(cw, c.kgB, c.xB);
            if (c.wave < 2) {
              const int nr = sweep(cw, o, c.wave == 0 ? delta : delta_b, c.wave == 0 ? abort_a : abort_b);
              c.sMisc[1 + c.wave] = (double)nr;        // (every lane: the count is wave-uniform; no lane-0 region in front of the barrier)
            }
          }
          wsync();
          const int na = (int)c.sMisc[1], nb = (int)c.sMisc[2];
          wsync();
          MYR_PH(6)
          nreg = na;
          if (na == 0 || !abort_a) break;
          delta = delta_b; nreg = nb;
          if (nb == 0 || !abort_b) {        // the speculative sweep stands: its outputs become set A (2.2 k doubles for CARTPOLE N = 100)
            for (int i = c.tid; i < c.N * KST; i += NT) c.kgA[i] = c.kgB[i];
            for (int i = c.tid; i < EXCH; i += NT) c.xA[i] = c.xB[i];
            wsync();
            break;
          }
          delta = next_delta(delta_b);
        }
      } else {
        for (int tr_ = 0; tr_ < 12; ++tr_) {
          const bool abort_on_reg = (tr_ < 11) && !(delta > 1e8);
          if constexpr (W > 1) {
            if (c.wave == 0) {
              nreg = sweep(c, o, delta, abort_on_reg);
              c.sMisc[1] = (double)nreg;
            }
            wsync();
            nreg = (int)c.sMisc[1];
          } else nreg = sweep(c, o, delta, abort_on_reg);
          wsync();
          MYR_PH(6)
          if (nreg == 0) break;
          if (!abort_on_reg) break;
          delta = next_delta(delta);
        }
      }
      delta_last = (delta > lm) ? delta : 0.0;
#ifdef MYR_TRACE
      if (c.lane == 0 && c.traj < MYR_TRACE)
        printf("T b%d w%d it%d f=%.17g c1=%.17g cinf=%.17g stat=%.17g sm=%.17g lg=%.17g nreg=%d delta=%.9g mu=%.9g pen=%.9g\n", c.traj, c.wave, it,
               p1.f, p1.c1, p1.cinf, stat_raw, p1.sum_mult, p1.lg, nreg, delta, mu, pen);
#endif
      if constexpr (!EARLY) {
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
      }
      typename S::SweepOut so;
#pragma unroll
      for (int i = 0; i < NS * NC; ++i) so.Tnu[i] = c.sTnu[i];
#pragma unroll
      for (int i = 0; i < NS; ++i) so.term_pinned[i] = c.term_pinned[i];
      double nu[NS];
      S::solve_nu(so, TL ? 0.0 : mu, nu);      // (two-level sweep: mu is folded into the "1" column)
      double th[NC];
      th[0] = 1.0; th[1] = TL ? 0.0 : mu;
#pragma unroll
      for (int i = 0; i < NS; ++i) th[2 + i] = nu[i];
      if constexpr (TL) {
        if (c.wave == 0) tl_theta(c, th);
        wsync();
      }
      MYR_PH(7)
      typename S::FwdOut fo;
      { ThT th_; for (int q = 0; q < NC; ++q) th_.v[q] = th[q]; fo = forward_pass(c, o, mu, th_); }
      wsync();
      MYR_PH(8)
      if (!(finite_(fo.gphi) && finite_(fo.alpha_p))) { res.status = 2; res.iters = it; return; }
      if (c1 > 0.0) {
        const double need = fo.gphi / (0.9 * c1);
        if (pen < need) pen = need + 1.0;
        if (PEN_RELAX > 0) {
          const double want = 2.0 * dmax(need, 0.0) + 1.0;
          pen_over = (pen > PEN_RELAX_RATIO * want) ? pen_over + 1 : 0;
          if (pen_over >= PEN_RELAX && pen_cuts < PEN_RELAX_MAX) { pen = want; pen_over = 0; ++pen_cuts; }
        }
      }
      const double Dphi = fo.gphi - pen * c1;
      const double f0 = p1.f, bar0 = mu * p1.lg, c10 = c1;
      const double phi0 = f0 + bar0 + pen * c10;
      if (mu != hist_mu || pen != hist_pen) { nhist = 0; hpos = 0; hist_mu = mu; hist_pen = pen; }
      double phiref = phi0;
      for (int j = 0; j < nhist; ++j) phiref = dmax(phiref, hist[j]);
      if (o.nonmono > 0) { hist[hpos % o.nonmono] = phi0; ++hpos; if (nhist < o.nonmono) ++nhist; }
      double a = fo.alpha_p;
      bool ok = false;
      for (int ls = 0; ls < 40; ++ls) {
        double ft, bt, ct;
        const TrialOut tr1 = trial_pass(c, a, mu);
        ft = tr1.f; bt = tr1.bar; ct = tr1.c1;
        if (tr1.ok) {
          const double phit = ft + bt + pen * ct;
          if (phit <= phiref + 1e-8 * a * Dphi + 1e-13 * fabs(phi0)) { ok = true; break; }
        }
        a *= 0.5;
      }
      if (!ok) {
        if (++stall > 5) { res.status = 3; res.iters = it; return; }
      } else stall = 0;
      c.h_valid = ok;                       // network systems: the accepted trial point is the next iterate, its activations are in the store
      MYR_PH(11)
#ifdef MYR_TRACE
      if (c.lane == 0 && c.traj < MYR_TRACE)
        printf("S b%d w%d it%d ap=%.17g ad=%.17g gphi=%.17g a=%.17g ok=%d nu0=%.17g\n", c.traj, c.wave, it, fo.alpha_p, fo.alpha_d, fo.gphi, a, (int)ok, nu[0]);
#endif
      pending.on = true; pending.ap = a; pending.ad = o.dual_follow ? fo.alpha_d * (a / fo.alpha_p) : fo.alpha_d; pending.mu = mu;
#pragma unroll
      for (int i = 0; i < NS; ++i) nuT[i] += a * (nu[i] - nuT[i]);
      if (o.lm_init > 0.0) {
        const double ratio = o.lm_abs ? a : a / fo.alpha_p;
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

// Persistent, one trajectory per wavefront (workgroup = one wavefront): every workgroup pulls trajectories from `ticket` until the
// batch is done and owns ONE scratch block that it re-uses for all of them.
#ifndef MYR_FUSED_OCC
#define MYR_FUSED_OCC 1      // workgroups per SIMD the register allocation must allow (experiment, tools/dev/exp/exp109.sh: 2 = 256 registers per lane)
#endif
template <class Sys, int NWAVES = 1, int SCHEME = 0>
__global__ __launch_bounds__(64 * NWAVES, MYR_FUSED_OCC)
void hs_solve_fused_kernel(int B, int* ticket, HsSolveOpts o, VarScale vs, double* __restrict__ z, const double* __restrict__ lb,
                           const double* __restrict__ ub, double* lam, double* scratch, long scratch_stride,
                           const double* __restrict__ params, int params_stride, double* cost, int32_t* status,
                           int32_t* iters, double* kkt, unsigned long long poison, ParkArgs pk, CoopArgs co) {
  using W = HsFused<Sys, NWAVES, SCHEME>;
  extern __shared__ __attribute__((aligned(16))) char smem_fused[];
  typename W::Ctx c;
  c.N = o.N; c.K = W::npoints(o.N); c.n = c.K * W::NW; c.lane = threadIdx.x & 63; c.tid = threadIdx.x;
#ifdef MYR_WAVE_DIVERGENT       // (round 3's form: the compiler cannot see that the wavefront index is the same in every lane)
  c.wave = threadIdx.x >> 6;
#else
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform BY CONSTRUCTION: `if (c.wave == 0)` is a scalar branch
#endif
  c.h = o.h; c.h6 = o.h / 6.0; c.h8 = o.h / 8.0;
  // helper workgroups of the network kernel (NodeBoard): owners are workgroups 0..B-1, helper h of owner w is workgroup h B + w
  const bool coop = W::MLP && co.maxh > 0;
  const bool fixed = coop && B <= (int)gridDim.x;        // a batch that leaves workgroups without a trajectory: trajectory b belongs to workgroup b
  const int nboards = fixed ? B : (int)gridDim.x;        // workgroups that can own a trajectory (boards, published vectors and scratch slots exist for these)
  const bool can_own = !fixed || (int)blockIdx.x < B;
  c.nh = 0; c.hidx = 0;
  c.board = (coop && can_own) ? co.boards + blockIdx.x : nullptr; c.coop_abort = co.abort;
  c.pubx = (coop && can_own) ? co.pub + (long)blockIdx.x * co.pub_stride : nullptr;
  c.publam = c.pubx ? c.pubx + c.n : nullptr; c.pubf = c.pubx ? c.publam + W::MLAM * c.N * W::NS : nullptr;
  double* s = scratch + (long)(can_own ? blockIdx.x : 0u) * scratch_stride;
  c.zr = s; c.hr = s + W::off_hr(c.N); c.st = s + W::off_st(c.N); c.kg = s + W::off_kg(c.N);
  c.kgA = c.kg; c.kgB = s + W::off_kg2(c.N); c.pt = s + W::off_pt(c.N);
  c.hb = s + W::off_hb(c.N); c.mb = s + W::off_mb(c.N); c.h_valid = false;
  c.fw = s + W::off_fw(c.N);
  double* const lam_own = s + W::off_lam(c.N);
  double* l = reinterpret_cast<double*>(smem_fused);
  c.z = l; l += c.n;
  if constexpr (W::ZLU_GLOBAL) { c.zL = s + W::off_zlu(c.N); c.zU = c.zL + c.n; }
  else { c.zL = l; l += c.n; c.zU = l; l += c.n; }
  c.dz = l; l += c.n;
  c.sLam = l; l += W::MLAM * c.N * W::NS;
  c.sB = l; l += 6 * W::NW;
  c.sStash = l; l += 2 * NWAVES * W::NREC;
  c.sTot = l; l += 2 * NWAVES * W::NTOT;
  c.sTr = l; l += 2 * NWAVES * 2 * W::NS;
  c.sRed = l; l += NWAVES * W::NRED;
  c.sMisc = l; l += 8;
  c.xA = l; c.xB = l + W::EXCH;
  c.sTh = l + W::NXB * W::EXCH; c.sJn = c.sTh + W::TL_TH;
  W::use_set(c, c.kgA, c.xA);
  c.sF = reinterpret_cast<double*>(smem_fused) + W::lds_solver_doubles(c.N);
  c.wl = c.sF + c.K * W::NS;
  if constexpr (W::MLP) {        // weights shared by the batch: loaded once per workgroup
    if (params_stride == 0) {
      SysParams<Sys> pw;
      pw.load(params, 0, 0);
      NodeMfma64::load_weights(pw.get(), c.wl, c.tid, W::NT);
      __syncthreads();
    }
  }
  if constexpr (W::MLP) {
    if (coop) {
      if (c.tid == 0) {
        W::coop_seq(c) = 0; W::coop_target(c) = 0;
        if (c.board) __hip_atomic_store(&c.board->xcc, 1u + coop_xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
  }
  bool took = false;
  c.gen = 0;
  for (;;) {
    int t = 0;
    if (c.tid == 0) t = fixed ? ((can_own && !took) ? (int)blockIdx.x : B) : atomicAdd(ticket, 1);
    took = true;
    if constexpr (NWAVES > 1) {
      if (c.tid == 0) reinterpret_cast<int*>(c.sMisc)[0] = t;
      __syncthreads();
      t = reinterpret_cast<int*>(c.sMisc)[0];
    }
    long b = __builtin_amdgcn_readfirstlane(t);
    if (pk.mode == 2) {          // resume order: ticket t takes the t-th longest parked trajectory
      if (b >= pk.count[0]) break;
      b = pk.perm[b];
    } else if (b >= B) break;
    double* zg = z + b * (long)c.n;
    c.lb = lb + b * (long)c.n; c.ub = ub + b * (long)c.n;
    double* lamg = lam ? lam + b * (long)(W::MLAM * c.N * W::NS) : lam_own;
    if (poison) {
      // MYRIAD_POISON (tests/test_gpu_poison.py): everything a trajectory inherits from its predecessor in this slot -- the solver's
      // LDS and the slot's global scratch -- is overwritten with one bit pattern (a signalling NaN, or plain garbage).  A solve
      // whose result depends on the pattern reads something before it writes it.
      __syncthreads();           // (every wavefront has read the ticket from sMisc)
      const unsigned int keep_seq = W::coop_seq(c), keep_target = W::coop_target(c);      // (the owner's command counters live in the poisoned LDS)
      __syncthreads();
      double* l0 = reinterpret_cast<double*>(smem_fused);
      const int nl = W::lds_solver_doubles(c.N) + (W::MLP ? c.K * W::NS : 0);
      const unsigned long long salt = (unsigned long long)b * 1315423911ULL + blockIdx.x;
      for (int i = c.tid; i < nl; i += W::NT) l0[i] = poison_value(poison, (unsigned long long)i, salt);
      for (long i = c.tid; i < scratch_stride; i += W::NT) s[i] = poison_value(poison, (unsigned long long)i + (1ULL << 32), salt);
      __syncthreads();
      if (c.tid == 0) { W::coop_seq(c) = keep_seq; W::coop_target(c) = keep_target; }
      __syncthreads();
    }
#ifdef MYR_TRACE
    c.traj = (int)b;
#endif
    c.pp.load(params, b, params_stride);
    c.pp.set_scale(vs.s);
    if constexpr (W::MLP) {      // a weight set per trajectory
      if (params_stride != 0) {
        NodeMfma64::load_weights(c.pp.get(), c.wl, c.tid, W::NT);
        __syncthreads();
      }
    }
    HsSolveResult r;
#ifdef MYR_PHASE_TIMING
    for (int i = 0; i < 16; ++i) c.tph[i] = 0;
    c.t0 = clock64();
#endif
    if constexpr (W::MLP) {
      c.nh = 0;
      c.gen = (c.gen + 1u) & MYR_COOP_GEN_MASK;
      if (coop && c.tid == 0) {      // helpers may attach from here on (the previous trajectory's EXIT has left the CU: see below)
        __hip_atomic_store(&c.board->att, MYR_COOP_RUNNING | (c.gen << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if constexpr (NWAVES == 1 || W::MLP) W::solve(c, o, zg, r, pk.mode, pk.k1, pk.state + b * pk.stride);      // (W = 2 serves batches of one round only)
    else W::solve(c, o, zg, r);
    if (r.status != MYR_STATUS_PARKED_) {
      for (int i = c.tid; i < c.n; i += W::NT) zg[i] = c.z[i];
      for (int i = c.tid; i < W::MLAM * c.N * W::NS; i += W::NT) lamg[i] = c.sLam[i];
    }
#ifdef MYR_PHASE_TIMING
    if (c.lane == 0 && b < 4) {
      printf("traj %ld wave %d it %d: backward %lld hess %lld fold %lld chunk %lld wait %lld join %lld first %lld ricc %lld nu %lld forward %lld ls %lld\n",
             b, c.wave, r.iters, c.tph[0], c.tph[4], c.tph[3], c.tph[13], c.tph[5], c.tph[9], c.tph[10], c.tph[6], c.tph[7], c.tph[8], c.tph[11]);
    }
    if (!W::MLP && c.tid == 0 && blockIdx.x == 0) {
      printf("  workgroup 0, traj %ld it %d: backward pass in parts (cycles): linearisation %lld stage maps %lld scan + multipliers %lld\n", b, r.iters, bw_tph_[0], bw_tph_[1], bw_tph_[2]);
      bw_tph_[0] = bw_tph_[1] = bw_tph_[2] = 0;
    }
    if (W::MLP && c.tid == 0 && blockIdx.x == 0) {
      printf("  workgroup 0, traj %ld it %d: network passes of wavefront 0 (cycles): MODE0 %lld MODE3 %lld MODE4 %lld\n", b, r.iters, node_tph_[0], node_tph_[3], node_tph_[4]);
      node_tph_[0] = node_tph_[3] = node_tph_[4] = 0;
      for (int m = 0; m < 5; m += (m == 0 ? 3 : 1)) {
        printf("    MODE %d segments: inputs %lld | layer 1 + sigmoids %lld | layer 2 %lld | sigmoids 2 %lld | activations stored / loaded %lld | last layer %lld | tangents / contraction %lld\n", m,
               node_seg_[m][0], node_seg_[m][1], node_seg_[m][2], node_seg_[m][3], node_seg_[m][4], node_seg_[m][5], node_seg_[m][6]);
        for (int q = 0; q < 8; ++q) node_seg_[m][q] = 0;
      }
    }
#endif
    if (c.tid == 0) {
      if (cost) cost[b] = r.cost;
      if (status) status[b] = r.status;
      if (iters) iters[b] = r.iters;
      if (kkt) { kkt[3 * b] = r.feas; kkt[3 * b + 1] = r.stat; kkt[3 * b + 2] = r.compl_; }
    }
    W::wsync();      // the slot's scratch and LDS are handed to the next trajectory
    if constexpr (W::MLP) {
      if (coop) {        // no more attaching; release the helpers; one more trajectory finished
        if (c.tid == 0) {
          (void)__hip_atomic_exchange(&c.board->att, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          coop_release();            // (the exchange has been performed at the L2 before the EXIT is issued, the EXIT before the next trajectory's attach word)
          const unsigned int seq = ++W::coop_seq(c);
          __hip_atomic_store(&c.board->cmd, ((unsigned long long)seq << 32) | ((unsigned long long)c.gen << 16) | (unsigned long long)MYR_COOP_EXIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          coop_release();
          (void)__hip_atomic_fetch_add(co.abort + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        c.nh = 0;
        __syncthreads();
      }
    }
  }
  if constexpr (W::MLP) {
    if (coop) {
      typename W::HelpArgs ha;
      ha.N = c.N; ha.K = c.K; ha.n = c.n; ha.lane = c.lane; ha.wave = c.wave; ha.tid = c.tid; ha.h6 = c.h6; ha.h8 = c.h8;
      ha.z = c.z; ha.sLam = c.sLam; ha.sF = c.sF; ha.wl = c.wl; ha.sMisc = c.sMisc;
      ha.co = co; ha.scratch = scratch; ha.scratch_stride = scratch_stride; ha.B = B; ha.nboards = nboards;
      W::coop_help(ha);
    }
  }
}

}  // namespace myriad
