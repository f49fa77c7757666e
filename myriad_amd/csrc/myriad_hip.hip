// myriad_hip.hip -- C-ABI (include/myriad_hip.h) over the gfx950 kernels.  No CPU fallback exists:
// every entry point either runs the HIP kernels or fails with an error code.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <thread>
#include <atomic>
#include <chrono>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/myriad_hip.h"
#include "dbg_regfill.h"
#include "hs_eval.h"
#include "colloc_products.h"
#include "hs_solver.h"
#include "hs_solver_wave.h"
#include "hs_solver_fused.h"
#include "shoot_solver_wave.h"
#include "os_solver.h"
#include "shoot_eval.h"
#include "rollout.h"
#include "fbsm.h"
#include "systems_gen.h"
#include "node_system.h"

// closed-form systems (tools/gen_systems.py): X(NAME) expands once per system; the enum values are MYR_SYS_<NAME>
#define MYR_CLOSED_FORM_SYSTEMS(X)                                                                               \
  X(CARTPOLE) X(VANDERPOL) X(CANCERTREATMENT) X(SIMPLECASE) X(BIOREACTOR) X(GLUCOSE) X(MOULDFUNGICIDE)           \
  X(SIMPLECASEWITHBOUNDS) X(HIVTREATMENT) X(EPIDEMICSEIRN) X(SEIR) X(BEARPOPULATIONS) X(PENDULUM) X(MOUNTAINCAR)     \
  X(ROCKETLANDING) X(BACTERIA) X(TUMOUR) X(HARVEST) X(TIMBERHARVEST) X(PREDATORPREY)                               \
  X(PENDULUM_ELASTIC) X(ROCKETLANDING_ELASTIC) X(CARTPOLE_ELASTIC) X(VANDERPOL_ELASTIC) X(MOUNTAINCAR_ELASTIC)


using namespace myriad;

// Translation units (see __graft_entry__.build): the library is either this one file compiled as is, or -- to build in
// parallel -- one object with -DMYR_TU_MAIN (C-ABI + dispatch, no per-system kernels) plus one object per system with
// -DMYR_TU_SYSTEM=<Sys...> (explicit instantiation of that system's entry points, nothing else exported).
#if defined(MYR_TU_SYSTEM)
extern thread_local std::string g_err;
#else
thread_local std::string g_err;
#endif
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (expr);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return fail(MYR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
  } while (0)

struct SysInfo { int ns, nu, np; bool cost_dep_x; };
static bool sys_info(int id, SysInfo* s) {
  switch (id) {
#define X(N) case MYR_SYS_##N: *s = {Sys##N::NS, Sys##N::NU, Sys##N::NP, Sys##N::COST_DEP_X}; return true;
    MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
    case MYR_SYS_NODE_CARTPOLE: *s = {SysNODE_CARTPOLE::NS, SysNODE_CARTPOLE::NU, SysNODE_CARTPOLE::NP, SysNODE_CARTPOLE::COST_DEP_X}; return true;
    case MYR_SYS_INVASIVEPLANT: *s = {DiscINVASIVEPLANT::NS, DiscINVASIVEPLANT::NU, DiscINVASIVEPLANT::NP, false}; return true;   // myr_fbsm only
  }
  return false;
}

struct myr_handle_s;
static int device_cus_early(myr_handle_s* h);
struct KTimer {
  hipEvent_t a = nullptr, b = nullptr;
  double sum_ms = 0.0;
  int launches = 0;
};

struct myr_handle_s {
  myr_problem_desc d;
  myr_dims dims;
  SysInfo si;
  hipStream_t stream = nullptr;
  KTimer kt[MYR_K_COUNT];
  // staging for MYR_MEM_HOST calls
  void* dbuf = nullptr;
  size_t dbuf_bytes = 0;
  int eval_wpt = 8;
  int eval_nt = 1;      // non-temporal stores for the c / J-block streams
  int solve_mode = 1;   // 1: one trajectory per wavefront (hs_solver_fused.h / hs_solver_wave.h); 0: one trajectory per lane (hs_solver.h)
  int solve_fused = 1;  // 1: the fused-phase wavefront kernel where it is built (MYRIAD_SOLVE_MODE=wave1 selects round 2's HsWave)
  int solve_lpw = 16;   // trajectories (active lanes) per wavefront in the solve kernel
  // solver scratch (batch-minor / SoA)
  void* sbuf = nullptr;
  size_t sbuf_bytes = 0;
  int* ticket = nullptr;      // work counter of the persistent solve kernel (one int)
  int solve_slots = 0;        // MYRIAD_SOLVE_SLOTS: resident wavefronts of the solve kernel (0 = what the device holds)
  std::map<std::tuple<const void*, int, size_t>, int> occ;   // kernel_slots(): workgroups per CU of (kernel, block size, dynamic LDS)
  size_t eval_attr_lds[6] = {0, 0, 0, 0, 0, 0};   // dynamic-LDS attribute already set for the eval kernel variants (W = 1 / 4 / 8, nt)
  int fused_waves = 0;        // MYRIAD_FUSED_WAVES: wavefronts per trajectory (0 = by batch size)
  // two-phase launch of the one-wavefront fused kernel (hs_solver_fused.h: ParkArgs): records of the parked trajectories, resume order
  void* park_state = nullptr; size_t park_bytes = 0;
  int* park_perm = nullptr; size_t park_n = 0;      // [park_n] resume order, then PARK_BUCKETS + 1 counters
  int park_iter = -1;         // MYRIAD_PARK_ITER: iterations of phase 1 (-1 = by batch size, 0 = whole solves only)
  // restoration inside myr_solve (myr_solve_opts.restoration): the twin handle, device scratch, and the account of the last call
  myr_handle_s* twin = nullptr;
  bool twin_unavailable = false;
  void* rbuf = nullptr; size_t rbuf_bytes = 0;      // copy of the caller's guess + status / iters when the caller passed none
  void* fbuf = nullptr; size_t fbuf_bytes = 0;      // working set of the failed instances
  int32_t* nfail_host = nullptr;                    // pinned: instances the first attempt left without a KKT point (copied from nfail_dev)
  int32_t* nfail_dev = nullptr;                     // the device word the count kernel adds to (device-scope atomics on device memory: no PCIe atomics needed)
  std::vector<int32_t> info_start, info_attempts, info_restored;
  int last_solve_form = 1;         // kernel form of the last solve launch on this handle: 1 = a wavefront kernel, 0 = the lane kernel
  unsigned long long poison = 0;   // MYRIAD_POISON: bit pattern written over a slot's LDS and scratch at every trajectory hand-over (tests)
  int cus = 0;                // compute units of the device (cached)
  // variable scaling of the solve path (myr_set_var_scale): the solver kernels see z/s, lb/s, ub/s
  VarScale vscale{{1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}};
  bool vscale_on = false;
  void* vbuf = nullptr;       // scaled copies of lb, ub
  size_t vbuf_bytes = 0;
  // helper workgroups of the network kernel (hs_solver_fused.h: NodeBoard): boards | abort word | published vectors
  void* coop_buf = nullptr; size_t coop_bytes = 0;
  int node_helpers = -1;      // MYRIAD_NODE_HELPERS: helper workgroups per trajectory (-1 = by batch size)
  int32_t plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // myr_solve_plan: how the last first-attempt solve was launched
  bool plan_frozen = false;                     //   (second starts re-launch on this handle: they leave the first attempt's record alone)
  int reg_fill = 0;                    // MYRIAD_REG_FILL (tests): pattern left in every VGPR / AGPR of every SIMD before every solver launch (1 nan, 2 finite, 3 big, 4 zero)
  unsigned long long stack_fill = 0;   // MYRIAD_STACK_FILL (tests / experiments): bit pattern left in the queue's private-segment memory before every solver launch
};

// MYRIAD_STACK_FILL: what a kernel's spill slots and stack objects inherit.  The private segment ("scratch") of a queue is not cleared between
// launches: a kernel that reads a stack slot before writing it gets what the previous kernel ON THIS QUEUE left there -- the same leftovers call after
// call on one handle, something else on a fresh handle (new stream, new queue, new backing memory).  This kernel overwrites 4 KB of private memory per
// lane of every resident wavefront slot with one pattern (MYRIAD_POISON does the same for the LDS and the solver's global scratch slots).
template <int WORDS>
static __global__ __launch_bounds__(64) void stack_fill_kernel(unsigned long long pat, int salt, unsigned long long* sink) {
  volatile unsigned long long a[WORDS];
  for (int i = 0; i < WORDS; ++i) a[i] = pat == 2 ? (0x3ff0000000000000ULL + ((unsigned long long)(i * 2654435761u + threadIdx.x * 40503u + salt) << 20)) : pat;
  unsigned long long acc = 0;
  for (int i = threadIdx.x & 7; i < WORDS; i += 61) acc ^= a[i];
  if (acc == 0x1234567ULL) sink[0] = acc;
}
// ... and the form for locating ONE slot: exactly `BYTES` of private memory per lane (the solver kernel's own private-segment size, so that the wave
// slots of the two kernels coincide), every dword the NaN pattern except dwords [z0, z1), which are zero
template <int DWORDS>
static __global__ __launch_bounds__(64) void stack_fill_window_kernel(int z0, int z1, unsigned* sink) {
  volatile unsigned a[DWORDS];
  for (int i = 0; i < DWORDS; ++i) a[i] = (i >= z0 && i < z1) ? 0u : ((i & 1) ? 0x7ff40000u : 0x00000001u);
  unsigned acc = 0;
  for (int i = threadIdx.x & 7; i < DWORDS; i += 61) acc ^= a[i];
  if (acc == 0x1234567u) sink[0] = acc;
}
static int stack_fill(myr_handle h) {
  if (const char* w = getenv("MYRIAD_STACK_FILL_WINDOW")) {      // "bytes:z0:z1"
    int bytes = 0, z0 = 0, z1 = 0;
    if (sscanf(w, "%d:%d:%d", &bytes, &z0, &z1) == 3) {
      if (!h->ticket) HIPCHK(hipMalloc(&h->ticket, sizeof(int)));
      for (int rep = 0; rep < 3; ++rep) {
        if (bytes == 528) hipLaunchKernelGGL((stack_fill_window_kernel<132>), dim3((unsigned)(device_cus_early(h) * 16)), dim3(64), 0, h->stream, z0, z1, (unsigned*)h->ticket);
        else hipLaunchKernelGGL((stack_fill_window_kernel<512>), dim3((unsigned)(device_cus_early(h) * 16)), dim3(64), 0, h->stream, z0, z1, (unsigned*)h->ticket);
      }
      HIPCHK(hipGetLastError());
      return MYR_OK;
    }
  }
  if (h->reg_fill) {      // MYRIAD_REG_FILL: what the registers inherit (dbg_regfill.h)
    const unsigned rp = h->reg_fill == 1 ? 0x7ff40000u : (h->reg_fill == 3 ? 0x4415af1du : (h->reg_fill == 4 ? 0u : 0x3ff12345u));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(myriad::reg_fill_kernel, dim3((unsigned)(device_cus_early(h) * 16)), dim3(64), 0, h->stream, rp);
    HIPCHK(hipGetLastError());
  }
  if (!h->stack_fill) return MYR_OK;
  if (!h->ticket) HIPCHK(hipMalloc(&h->ticket, sizeof(int)));
  const unsigned long long pat = h->stack_fill == 1 ? 0x7ff4000000000000ULL /* signalling NaN */ : (h->stack_fill == 3 ? 0x4415af1d78b58c40ULL /* 1e20 */ : (h->stack_fill == 4 ? 0ULL : 2ULL));
  for (int rep = 0; rep < 3; ++rep)
    hipLaunchKernelGGL((stack_fill_kernel<512>), dim3((unsigned)(device_cus_early(h) * 16)), dim3(64), 0, h->stream, pat, rep, (unsigned long long*)h->ticket);
  HIPCHK(hipGetLastError());
  return MYR_OK;
}

static int device_cus(myr_handle h) {
  if (h->cus <= 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) h->cus = cus;
    else h->cus = 256;
  }
  return h->cus;
}

static int device_cus_early(myr_handle_s* h) { return device_cus(h); }

// Workgroups of `kern` a CU keeps resident at this block size and dynamic-LDS size; sets the kernel's dynamic-LDS limit on the
// way.  Asked of the runtime once per handle and configuration, not on every solve call.
static int kernel_blocks_per_cu(myr_handle h, const void* kern, int threads, size_t lds, int* out) {
  auto key = std::make_tuple(kern, threads, lds);
  auto it = h->occ.find(key);
  if (it == h->occ.end()) {
    int per_cu = 0;
    // (the limit is a property of the FUNCTION, shared by every handle of the process: raise it to the hardware's 160 KB once
    // instead of to this handle's size, which a handle with a shorter horizon would lower again)
    HIPCHK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
    it = h->occ.emplace(key, per_cu).first;
  }
  *out = it->second;
  return MYR_OK;
}

static int ensure_buf(void** buf, size_t* have, size_t need) {
  if (need <= *have) return MYR_OK;
  if (*buf) HIPCHK(hipFree(*buf));
  *buf = nullptr; *have = 0;
  HIPCHK(hipMalloc(buf, need));
  *have = need;
  return MYR_OK;
}

static int ensure_dbuf(myr_handle h, size_t bytes) {
  if (bytes <= h->dbuf_bytes) return 0;
  if (h->dbuf) HIPCHK(hipFree(h->dbuf));
  h->dbuf = nullptr; h->dbuf_bytes = 0;
  HIPCHK(hipMalloc(&h->dbuf, bytes));
  h->dbuf_bytes = bytes;
  return 0;
}

// ------------------------------------------------------------------------------------------------
// eval
// ------------------------------------------------------------------------------------------------
template <class Sys, int SCHEME>
static int launch_hs_eval(myr_handle h, int B, const double* z, const double* params, int pstride,
                          double* f, double* g, double* c, double* j) {
  const int N = h->d.intervals;
  const double hstep = h->d.T / N;
  int wpt = h->eval_wpt;
  // LDS is sized for the number of wavefronts ACTUALLY launched (the non-NT fallback always runs 4 per workgroup): when the
  // preferred width does not fit a CU the launch steps down (8 -> 4 -> 1) before it gives up
  const bool nt_ = h->eval_nt != 0;
  auto launched = [&](int w) { return nt_ ? (w == 1 ? 1 : (w == 8 ? 8 : 4)) : 4; };
  while (hs_eval_lds_bytes<Sys, SCHEME>(N, launched(wpt)) > 160 * 1024 && nt_ && wpt > 1) wpt = wpt == 8 ? 4 : 1;
  if (hs_eval_lds_bytes<Sys, SCHEME>(N, launched(wpt)) > 160 * 1024) return fail(MYR_E_CAPACITY, "hs_eval: intervals too large for the 160 KiB LDS record");
  KTimer& kt = h->kt[MYR_K_EVAL];
  // the start event is recorded after the host-side attribute call, directly in front of the launch: the interval
  // between the two events is the kernel plus its dispatch, not host work
#define MYR_EVAL_LAUNCH(W, NTV)                                                                                   \
  {                                                                                                               \
    auto kern = hs_eval_kernel<Sys, W, NTV, SCHEME>;                                                                   \
    const size_t lds = hs_eval_lds_bytes<Sys, SCHEME>(N, W);                                                      \
    if (h->eval_attr_lds[(W == 1 ? 0 : (W == 4 ? 1 : 2)) + (NTV ? 3 : 0)] == 0) {      /* function-wide limit: the hardware's, once */ \
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      h->eval_attr_lds[(W == 1 ? 0 : (W == 4 ? 1 : 2)) + (NTV ? 3 : 0)] = 1;                                     \
    }                                                                                                             \
    if (int rc_fill = stack_fill(h)) return rc_fill;                                                              \
    HIPCHK(hipEventRecord(kt.a, h->stream));                                                                      \
    hipLaunchKernelGGL(kern, dim3(B), dim3(64 * W), lds, h->stream, N, hstep, z, params, pstride, f, g, c, j);    \
  }
  const bool nt = h->eval_nt != 0;
  switch (wpt) {
    case 1: if (nt) MYR_EVAL_LAUNCH(1, true) else MYR_EVAL_LAUNCH(4, false) break;
    case 8: if (nt) MYR_EVAL_LAUNCH(8, true) else MYR_EVAL_LAUNCH(4, false) break;
    default: if (nt) MYR_EVAL_LAUNCH(4, true) else MYR_EVAL_LAUNCH(4, false) break;
  }
#undef MYR_EVAL_LAUNCH
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms;
  kt.launches += 1;
  return MYR_OK;
}

template <class Sys>
static int launch_shoot_eval(myr_handle h, int B, const double* z, const double* params, int pstride,
                             double* f, double* g, double* c, double* j) {
  const int I = h->d.intervals, cpi = h->d.controls_per_interval, method = h->d.integration_method;
  const size_t need = (size_t)B * (size_t)(cpi + 1) * Sys::NS * 8;
  if (need > h->sbuf_bytes) {
    if (h->sbuf) HIPCHK(hipFree(h->sbuf));
    h->sbuf = nullptr; h->sbuf_bytes = 0;
    HIPCHK(hipMalloc(&h->sbuf, need));
    h->sbuf_bytes = need;
  }
  KTimer& kt = h->kt[MYR_K_EVAL];
  if (int rc_fill = stack_fill(h)) return rc_fill;
  HIPCHK(hipEventRecord(kt.a, h->stream));
  if (method == MYR_INT_RK4)
    hipLaunchKernelGGL((shoot_eval_kernel<Sys, 2>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, B, I, cpi, method, h->d.T,
                       z, params, pstride, f, g, c, j, (double*)h->sbuf, (const double*)nullptr, 1);
  else
    hipLaunchKernelGGL((shoot_eval_kernel<Sys, 1>), dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, B, I, cpi, method, h->d.T,
                       z, params, pstride, f, g, c, j, (double*)h->sbuf, (const double*)nullptr, 1);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms;
  kt.launches += 1;
  return MYR_OK;
}

template <class Sys>
int eval_for_system(myr_handle h, int B, const double* z, const double* params, int pstride,
                           double* f, double* g, double* c, double* j) {
  if constexpr (Sys::PARAMS_BY_POINTER) {   // neural-ODE systems: Hermite-Simpson (config 5) and shooting (the reference's default route for a NodeSystem,
                                            // config.py:66 + shooting.py:144-167, 212-228); its parametrised trapezoid is dead code (trapezoidal.py:204-206, quirk Q8)
    if (h->d.transcription == MYR_TR_SHOOTING) return launch_shoot_eval<Sys>(h, B, z, params, pstride, f, g, c, j);
    if (h->d.transcription != MYR_TR_HERMITE_SIMPSON) return fail(MYR_E_UNSUPPORTED, "myr_eval: NODE systems are built for HERMITE_SIMPSON and SHOOTING");
    return launch_hs_eval<Sys, EVAL_HS>(h, B, z, params, pstride, f, g, c, j);
  } else {
    switch (h->d.transcription) {
      case MYR_TR_HERMITE_SIMPSON: return launch_hs_eval<Sys, EVAL_HS>(h, B, z, params, pstride, f, g, c, j);
      case MYR_TR_TRAPEZOIDAL: return launch_hs_eval<Sys, EVAL_TRAP>(h, B, z, params, pstride, f, g, c, j);
      case MYR_TR_SHOOTING: return launch_shoot_eval<Sys>(h, B, z, params, pstride, f, g, c, j);
    }
  }
  return fail(MYR_E_ARG, "eval: unknown transcription");
}

// ------------------------------------------------------------------------------------------------
// Lagrangian products and the extragradient step (collocation transcriptions)
// ------------------------------------------------------------------------------------------------
enum { PRODOP_VJP = 0, PRODOP_JVP = 1, PRODOP_EXGD = 2 };
struct ProdArgs {
  int op, B;
  const double *z, *w, *params; int pstride;     // w: lam (vjp) or v (jvp)
  double* out; int add_gradf;
  double *zio, *lamio; const double *lb, *ub; double eta_x, eta_v; int nsteps;   // exgd
};

template <class Sys, int SCHEME>
static int launch_products(myr_handle h, const ProdArgs& a) {
  const int N = h->d.intervals;
  const double hstep = h->d.T / N;
  using P = CollocProducts<Sys, SCHEME>;
  KTimer& kt = h->kt[MYR_K_PROD];
  if (int rc_fill = stack_fill(h)) return rc_fill;
  HIPCHK(hipEventRecord(kt.a, h->stream));
  if (a.op == PRODOP_EXGD) {
    const size_t lds = colloc_exgd_lds_bytes<Sys, SCHEME>(N);
    if (lds > 160 * 1024) return fail(MYR_E_CAPACITY, "myr_exgd: intervals too large for the LDS-resident iterate");
    auto kern = colloc_exgd_kernel<Sys, SCHEME>;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)a.B), dim3(64), lds, h->stream, a.B, N, hstep, a.zio, a.lamio, a.lb, a.ub,
                       a.params, a.pstride, a.eta_x, a.eta_v, a.nsteps);
  } else {
    const long units = (long)a.B * (a.op == PRODOP_VJP ? P::points(N) : N);
    long blocks = (units + 255) / 256;
    if (blocks > 256L * 64) blocks = 256L * 64;       // grid-stride beyond 64 workgroups per CU
    if (a.op == PRODOP_VJP)
      hipLaunchKernelGGL((colloc_vjp_kernel<Sys, SCHEME>), dim3((unsigned)blocks), dim3(256), 0, h->stream, a.B, N, hstep, a.z, a.w,
                         a.params, a.pstride, a.out, a.add_gradf);
    else
      hipLaunchKernelGGL((colloc_jvp_kernel<Sys, SCHEME>), dim3((unsigned)blocks), dim3(256), 0, h->stream, a.B, N, hstep, a.z, a.w,
                         a.params, a.pstride, a.out);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms;
  kt.launches += 1;
  return MYR_OK;
}

// shooting: J^T lam / grad L by the reverse sweep of shoot_eval_kernel seeded with lam, J v by forward tangents, and the
// extragradient step as a launch sequence (the iterate of a shooting problem is small: no LDS-resident variant)
template <class Sys, int M>
static int launch_shoot_products_m(myr_handle h, const ProdArgs& a) {
  const int I = h->d.intervals, cpi = h->d.controls_per_interval, method = h->d.integration_method;
  const myr_dims& dm = h->dims;
  const size_t sweep = (size_t)a.B * (size_t)(cpi + 1) * Sys::NS * 8;
  const size_t extra = a.op == PRODOP_EXGD ? ((size_t)2 * a.B * dm.n + (size_t)a.B * dm.m) * 8 : 0;   // g, zbar, c
  if (sweep + extra > h->sbuf_bytes) {
    if (h->sbuf) HIPCHK(hipFree(h->sbuf));
    h->sbuf = nullptr; h->sbuf_bytes = 0;
    HIPCHK(hipMalloc(&h->sbuf, sweep + extra));
    h->sbuf_bytes = sweep + extra;
  }
  double* scr = (double*)h->sbuf;
  const dim3 grid((unsigned)((a.B + 63) / 64)), blk(64);
  KTimer& kt = h->kt[MYR_K_PROD];
  if (int rc_fill = stack_fill(h)) return rc_fill;
  HIPCHK(hipEventRecord(kt.a, h->stream));
  if (a.op == PRODOP_VJP) {
    hipLaunchKernelGGL((shoot_eval_kernel<Sys, M>), grid, blk, 0, h->stream, a.B, I, cpi, method, h->d.T, a.z, a.params, a.pstride,
                       (double*)nullptr, a.out, (double*)nullptr, (double*)nullptr, scr, a.w, a.add_gradf);
  } else if (a.op == PRODOP_JVP) {
    hipLaunchKernelGGL((shoot_jvp_kernel<Sys, M>), grid, blk, 0, h->stream, a.B, I, cpi, method, h->d.T, a.z, a.w, a.params, a.pstride, a.out);
  } else {
    double* g = scr + sweep / 8; double* zbar = g + (size_t)a.B * dm.n; double* cbuf = zbar + (size_t)a.B * dm.n;
    const long tz = (long)a.B * dm.n, tl = (long)a.B * dm.m;
    long ub_ = (tz + 255) / 256; if (ub_ > 16384) ub_ = 16384;
    for (int s = 0; s < a.nsteps; ++s) {
      hipLaunchKernelGGL((shoot_eval_kernel<Sys, M>), grid, blk, 0, h->stream, a.B, I, cpi, method, h->d.T, (const double*)a.zio, a.params, a.pstride,
                         (double*)nullptr, g, (double*)nullptr, (double*)nullptr, scr, (const double*)a.lamio, 1);
      hipLaunchKernelGGL(exgd_update_kernel, dim3((unsigned)ub_), dim3(256), 0, h->stream, tz, (const double*)a.zio, (const double*)g, a.lb, a.ub, a.eta_x, zbar);
      hipLaunchKernelGGL((shoot_eval_kernel<Sys, M>), grid, blk, 0, h->stream, a.B, I, cpi, method, h->d.T, (const double*)zbar, a.params, a.pstride,
                         (double*)nullptr, g, (double*)nullptr, (double*)nullptr, scr, (const double*)a.lamio, 1);
      hipLaunchKernelGGL(exgd_update_kernel, dim3((unsigned)ub_), dim3(256), 0, h->stream, tz, (const double*)a.zio, (const double*)g, a.lb, a.ub, a.eta_x, a.zio);
      hipLaunchKernelGGL((shoot_eval_kernel<Sys, M>), grid, blk, 0, h->stream, a.B, I, cpi, method, h->d.T, (const double*)a.zio, a.params, a.pstride,
                         (double*)nullptr, (double*)nullptr, cbuf, (double*)nullptr, scr, (const double*)nullptr, 1);
      hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, h->stream, tl, a.eta_v, (const double*)cbuf, a.lamio);
    }
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms;
  kt.launches += 1;
  return MYR_OK;
}

template <class Sys>
static int launch_shoot_products(myr_handle h, const ProdArgs& a) {
  return h->d.integration_method == MYR_INT_RK4 ? launch_shoot_products_m<Sys, 2>(h, a) : launch_shoot_products_m<Sys, 1>(h, a);
}

template <class Sys>
int products_for_system(myr_handle h, const ProdArgs& a) {
  if constexpr (Sys::PARAMS_BY_POINTER) {
    if (h->d.transcription != MYR_TR_HERMITE_SIMPSON) return fail(MYR_E_UNSUPPORTED, "products: NODE systems are built for HERMITE_SIMPSON");
    return launch_products<Sys, PROD_HS>(h, a);
  } else {
    switch (h->d.transcription) {
      case MYR_TR_HERMITE_SIMPSON: return launch_products<Sys, PROD_HS>(h, a);
      case MYR_TR_TRAPEZOIDAL: return launch_products<Sys, PROD_TRAP>(h, a);
      case MYR_TR_SHOOTING: return launch_shoot_products<Sys>(h, a);
      default: return fail(MYR_E_ARG, "products: unknown transcription");
    }
  }
}

// ------------------------------------------------------------------------------------------------
// solve
// ------------------------------------------------------------------------------------------------
// [rows][cols] row-major  ->  [cols][ld] (ld >= rows): instance-major <-> batch-minor, 32x32 LDS tiles
static __global__ __launch_bounds__(256) void transpose_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                        int rows, int cols, long ld) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int r = r0 + ty + i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + i][tx] = src[(long)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int c = c0 + ty + i, r = r0 + tx;
    if (r < rows && c < cols) dst[(long)c * ld + r] = tile[tx][ty + i];
  }
}
// [cols][ld] -> [rows][cols]
static __global__ __launch_bounds__(256) void transpose_back_kernel(const double* __restrict__ src, double* __restrict__ dst,
                                                             int rows, int cols, long ld) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int c = c0 + ty + i, r = r0 + tx;
    if (r < rows && c < cols) tile[ty + i][tx] = src[(long)c * ld + r];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int r = r0 + ty + i, c = c0 + tx;
    if (r < rows && c < cols) dst[(long)r * cols + c] = tile[tx][ty + i];
  }
}

// One trajectory per lane; every per-trajectory array is batch-minor (element i of trajectory b at a[i*Bp + b]),
// so the 64 lanes of a wavefront always touch 64 consecutive doubles (one 512-byte coalesced access).
template <class Core, class Sys>
__global__ __launch_bounds__(64, 1)
void lane_solve_kernel(int B, long Bp, int lanes_per_wave, HsSolveOpts o, VarScale vs, double* z, double* lb, double* ub, double* zL, double* zU,
                     double* lam, double* dz, double* st, const double* __restrict__ params, int params_stride,
                     double* cost, int32_t* status, int32_t* iters, double* kkt) {
  // The kernel is latency-bound (long dependent fp64 chains, one wave per SIMD at best), so a wavefront may be
  // launched partially populated: fewer trajectories per wave -> more waves in flight and a shorter wait for
  // the slowest trajectory of each wave.
  if ((int)threadIdx.x >= lanes_per_wave) return;
  const long b = (long)blockIdx.x * lanes_per_wave + threadIdx.x;
  if (b >= B) return;
  SysParams<Sys> pp;
  pp.load(params, b, params_stride);
  pp.set_scale(vs.s);
  const double* p = pp.get();
  HsWork w{{z + b, Bp}, {lb + b, Bp}, {ub + b, Bp}, {zL + b, Bp}, {zU + b, Bp}, {lam + b, Bp}, {dz + b, Bp}, {st + b, Bp}};
  HsSolveResult r;
  Core::solve(w, o, p, r);
  if (cost) cost[b] = r.cost;
  if (status) status[b] = r.status;
  if (iters) iters[b] = r.iters;
  if (kkt) { kkt[3 * b] = r.feas; kkt[3 * b + 1] = r.stat; kkt[3 * b + 2] = r.compl_; }
}

static HsSolveOpts make_opts(myr_handle h, const myr_solve_opts& so) {
  HsSolveOpts o;
  o.N = h->d.intervals; o.h = h->d.T / h->d.intervals; o.max_iter = so.max_iter; o.tol_feas = so.tol_feas;
  o.tol_stat = so.tol_stat; o.tol_compl = so.tol_compl; o.mu_init = so.mu_init;
  o.cpi = h->d.controls_per_interval; o.method = h->d.integration_method;
  o.restarts = so.restarts < 0 ? MYR_SHOOT_RESTARTS : (so.restarts > 4 ? 4 : so.restarts);
  // warm-started inertia correction: trades factorisation sweeps for (slightly more) iterations, which pays when a
  // sweep costs more than a linearisation -- collocation with closed-form dynamics; not shooting (the rollout
  // linearisation dominates, and the correction decays too slowly for its stragglers) nor network dynamics
  o.delta_warm = (h->d.transcription != MYR_TR_SHOOTING && h->d.system_id != MYR_SYS_NODE_CARTPOLE) ? 1 : 0;
  if (const char* e = getenv("MYRIAD_NONMONO")) o.nonmono = atoi(e);      // developer knobs (globalisation ablations)
  if (const char* e = getenv("MYRIAD_RECENTER")) o.recenter = atoi(e);
  if (const char* e = getenv("MYRIAD_DELTA_WARM")) o.delta_warm = atoi(e);
  if (const char* e = getenv("MYRIAD_DELTA_WARM_MIN")) o.delta_warm_min = atof(e);
  if (const char* e = getenv("MYRIAD_KAPPA_MU")) o.kappa_mu = atof(e);    // barrier schedule ablations
  if (const char* e = getenv("MYRIAD_THETA_MU")) o.theta_mu = atof(e);
  if (const char* e = getenv("MYRIAD_KAPPA_EPS")) o.kappa_eps = atof(e);
  if (const char* e = getenv("MYRIAD_MU_INIT")) o.mu_init = atof(e);
  if (o.nonmono < 0) o.nonmono = 0;
  if (o.nonmono > 8) o.nonmono = 8;
  return o;
}

template <class Core, class Sys>
static int launch_lane_solve(myr_handle h, int B, long nst, double* z, const double* lb, const double* ub, const double* params,
                             int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                             int32_t* iters, double* kkt);
#define MYR_SOLVE_ARGS myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params, int pstride, const myr_solve_opts& so, \
                       double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt
#define MYR_SOLVE_ARG_TYPES myr_handle, int, double*, const double*, const double*, const double*, int, const myr_solve_opts&, double*, double*, int32_t*, int32_t*, double*
// the lane kernels of the collocation solvers as an entry point of their own (objects of their own in a split build: MYR_TU_PART 5, 6)
template <class Sys, int SCHEME>
int solve_lane_colloc_for_system(MYR_SOLVE_ARGS);
#if defined(MYR_TU_SYSTEM) && defined(MYR_TU_PART)
#if MYR_TU_PART != 5
extern template int solve_lane_colloc_for_system<myriad::MYR_TU_SYSTEM, 0>(MYR_SOLVE_ARG_TYPES);
#endif
#if MYR_TU_PART != 6
extern template int solve_lane_colloc_for_system<myriad::MYR_TU_SYSTEM, 1>(MYR_SOLVE_ARG_TYPES);
#endif
#endif

// Resume order of a two-phase launch: parked trajectories by DESCENDING key (the geometric mean of the stationarity and the complementarity
// residual at the parking point: rank correlation 0.9 with the iterations still to go), bucketed by half powers of two.  Three small launches.
constexpr int PARK_BUCKETS = 256;
static __device__ inline int park_bucket(const double* kkt, int b) {
  const double st = kkt[3 * b + 1], cp = kkt[3 * b + 2];
  const double key = (cp > 0.0 && st > 0.0) ? sqrt(st * cp) : (st > cp ? st : cp);
  if (!(key > 0.0)) return 0;
  if (!(key < 1e300)) return PARK_BUCKETS - 1;
  int v = (int)floor(2.0 * log2(key)) + PARK_BUCKETS / 2 + 32;       // key 1 -> bucket 160; 1e-24 .. 1e14 spans the table
  return v < 0 ? 0 : (v > PARK_BUCKETS - 1 ? PARK_BUCKETS - 1 : v);
}
static __global__ void park_hist_kernel(int B, const int32_t* __restrict__ status, const double* __restrict__ kkt, int* __restrict__ cnt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && status[b] == myriad::MYR_STATUS_PARKED_) atomicAdd(&cnt[park_bucket(kkt, b)], 1);
}
static __global__ void park_scan_kernel(int* __restrict__ cnt) {        // one thread: 256 counters -> start offsets, largest bucket first
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int run = 0;
    for (int k = PARK_BUCKETS - 1; k >= 0; --k) { const int c = cnt[k]; cnt[k] = run; run += c; }
    cnt[PARK_BUCKETS] = run;
  }
}
static __global__ void park_scatter_kernel(int B, const int32_t* __restrict__ status, const double* __restrict__ kkt, int* __restrict__ cnt, int* __restrict__ perm) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && status[b] == myriad::MYR_STATUS_PARKED_) perm[atomicAdd(&cnt[park_bucket(kkt, b)], 1)] = b;
}

// fused-phase wavefront kernel (hs_solver_fused.h): Hermite-Simpson, closed-form systems with one control and <= 4 states.
// NWAVES wavefronts per trajectory: 1 for throughput (four trajectories per CU), 2 when the batch leaves CUs idle otherwise
// (B <= 2 trajectories per CU: the parallel phases of an iteration take half the time, the launch lasts as long as one solve).
template <class Sys, int NWAVES, int SCHEME>
static int launch_hs_fused_w(myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params,
                             int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                             int32_t* iters, double* kkt) {
  using W = HsFused<Sys, NWAVES, SCHEME>;
  const int N = h->d.intervals;
  const size_t lds = W::lds_bytes(N);
  auto kern = hs_solve_fused_kernel<Sys, NWAVES, SCHEME>;
  int per_cu = 0;       // (attributes and occupancy once per handle and configuration, not per call)
  if (int rc = kernel_blocks_per_cu(h, reinterpret_cast<const void*>(kern), 64 * NWAVES, lds, &per_cu)) return rc;
  int slots = h->solve_slots > 0 ? h->solve_slots : (per_cu > 0 ? per_cu : 4 / NWAVES) * device_cus(h);
  const int slots_full = slots;
  if (slots > B) slots = B;
  long stride = (W::scratch_doubles(N) + 31) / 32 * 32;
  if (((stride / 32) & 1) == 0) stride += 32;          // odd multiple of 256 B: rotate slots over HBM channels
  const size_t need = (size_t)slots * (size_t)stride * 8;
  if (need > h->sbuf_bytes) {
    if (h->sbuf) HIPCHK(hipFree(h->sbuf));
    h->sbuf = nullptr; h->sbuf_bytes = 0;
    HIPCHK(hipMalloc(&h->sbuf, need));
    h->sbuf_bytes = need;
  }
  if (!h->ticket) HIPCHK(hipMalloc(&h->ticket, sizeof(int)));
  HIPCHK(hipMemsetAsync(h->ticket, 0, sizeof(int), h->stream));
  HsSolveOpts o = make_opts(h, so);
  if (int rc = stack_fill(h)) return rc;
  KTimer& kt = h->kt[MYR_K_SOLVE];
  HIPCHK(hipEventRecord(kt.a, h->stream));
  h->last_solve_form = 1;
  if (getenv("MYRIAD_DEBUG_PTRS"))
    fprintf(stderr, "[myriad] fused W=%d N=%d B=%d slots=%d lds=%zu stride=%ld doubles: scratch [%p, %p) z %p lb %p ub %p lam %p ticket %p\n", NWAVES, N, B, slots, lds, stride,
            h->sbuf, (char*)h->sbuf + need, (void*)z, (const void*)lb, (const void*)ub, (void*)lam, (void*)h->ticket);
  // Two phases when a slot would take at least two whole solves (B >= 2 x resident wavefronts): k1 iterations for every trajectory, the
  // unfinished ones parked, then resumed longest-first (hs_solver_fused.h: ParkArgs).  Needs the per-instance status and residuals.
  int k1 = 0;
  if constexpr (NWAVES == 1 || W::MLP) {        // the one-wavefront kernels and the network kernel (four wavefronts, one trajectory per CU: B = 1024 is four rounds)
    k1 = so.park_iter != 0 ? so.park_iter : h->park_iter;
    if (k1 < 0 && so.park_iter < 0) k1 = 0;                     // opts.park_iter = -1: whole solves
    else if (k1 < 0) {
      // by the number of solves a resident wavefront gets (measured on the headline workload, tools/dev/exp/exp47.sh: B = 1536 +4 %, 2048 +7 %,
      // 3072 +16 %, 4096 +9 %, 8192 +7 %; below one and a half solves per wavefront whole solves are faster)
      const double per_slot = (double)B / (double)slots_full;
      k1 = (SCHEME != 0 || W::MLP) ? 0 : (per_slot >= 3.0 ? 12 : (per_slot >= 2.0 ? 10 : (per_slot >= 1.5 ? 8 : 0)));
    }       // a little more than half of a typical interior-point solve (20-25 iterations);
                                                                               // (trapezoidal solves are shorter and closer together: two phases cost them 2.5 %;
                                                                               //  the network system's longest solves -- 75 iterations against a median of 24 -- have SMALL
                                                                               //  residuals at the parking point and would come last: 41 -> 52 ms at B = 1024, exp42 / exp43)
    if (k1 >= o.max_iter || !status || !kkt) k1 = 0;
    if (W::ZLU_GLOBAL && W::MLP) k1 = 0;        // (the network kernel's throughput form: its helper protocol and activation store are not part of a parked record;
                                                //  the closed-form systems that keep the bound multipliers in the slot's scratch park them with the solver's LDS)
  }
  // Helper workgroups for the network passes (hs_solver_fused.h: NodeBoard): a batch of at most half the CUs (config 5's share of an 8-GPU node is 128
  // trajectories) gets nh = #CU / B - 1 <= 3 more workgroups per trajectory; whole solves with one shared weight set only.
  myriad::CoopArgs co{nullptr, nullptr, 0, 0, nullptr};
  unsigned grid = (unsigned)slots;
  if constexpr (W::MLP) {
    const int cus = device_cus(h);
    const int resident = per_cu * cus;                               // workgroups the device holds: 1 per CU (four wavefronts), 2 per CU (two)
    int maxh = NWAVES == 2 ? 5 : 3;                                  // 13 tiles over 4 (nh + 1) resp. 2 (nh + 1) wavefronts
    if (B <= resident / 2 && resident / B - 1 < maxh) maxh = resident / B - 1;      // a small batch: spread the idle workgroups evenly
    if (h->node_helpers >= 0 && h->node_helpers < maxh) maxh = h->node_helpers;
    // whole solves with one shared weight set; every workgroup of the launch must be resident: owners wait for their helpers
    if (pstride != 0 || k1 > 0 || per_cu < 1 || per_cu != 4 / NWAVES || h->solve_slots > 0) maxh = 0;
    if (maxh > 0) {
      grid = (unsigned)((long)B * (maxh + 1) < (long)resident ? B * (maxh + 1) : resident);
      const int nboards = B <= (int)grid ? B : (int)grid;      // workgroups that can own a trajectory (hs_solve_fused_kernel: `fixed`)
      const long pub = ((long)h->dims.n + (long)W::MLAM * N * W::NS + (long)W::npoints(N) * W::NS + 15) / 16 * 16;
      const size_t head = ((size_t)nboards * sizeof(myriad::NodeBoard) + 128 + 127) / 128 * 128;
      if (int rc = ensure_buf(&h->coop_buf, &h->coop_bytes, head + (size_t)nboards * (size_t)pub * 8)) return rc;
      HIPCHK(hipMemsetAsync(h->coop_buf, 0, head, h->stream));
      co.boards = (myriad::NodeBoard*)h->coop_buf;
      co.abort = (int*)((char*)h->coop_buf + (size_t)nboards * sizeof(myriad::NodeBoard));
      co.pub = (double*)((char*)h->coop_buf + head);
      co.pub_stride = pub; co.maxh = maxh;
    }
  }
  myriad::ParkArgs pk{0, 0, nullptr, nullptr, nullptr, 0};
  if (k1 > 0) {
    const long pstr = (W::park_doubles(N) + 31) / 32 * 32;
    const size_t pneed = (size_t)B * (size_t)pstr * 8;
    if (pneed > h->park_bytes) {
      if (h->park_state) HIPCHK(hipFree(h->park_state));
      h->park_state = nullptr; h->park_bytes = 0;
      if (hipMalloc(&h->park_state, pneed) == hipSuccess) h->park_bytes = pneed;
      else { (void)hipGetLastError(); h->park_state = nullptr; k1 = 0; }      // no room for the records (41 KB per instance): whole solves
    }
  }
  if (k1 > 0) {
    const long pstr = (W::park_doubles(N) + 31) / 32 * 32;
    if ((size_t)B > h->park_n) {
      if (h->park_perm) HIPCHK(hipFree(h->park_perm));
      h->park_perm = nullptr; h->park_n = 0;
      HIPCHK(hipMalloc(&h->park_perm, ((size_t)B + PARK_BUCKETS + 1) * sizeof(int)));
      h->park_n = (size_t)B;
    }
    int* cnt = h->park_perm + h->park_n;
    pk = myriad::ParkArgs{1, k1, h->park_perm, cnt + PARK_BUCKETS, (double*)h->park_state, pstr};
    hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3(64 * NWAVES), lds, h->stream, B, h->ticket, o, h->vscale, z, lb, ub, lam, (double*)h->sbuf, stride,
                       params, pstride, cost, status, iters, kkt, h->poison, pk, co);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(cnt, 0, (PARK_BUCKETS + 1) * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(h->ticket, 0, sizeof(int), h->stream));
    const unsigned gb = (unsigned)((B + 255) / 256);
    hipLaunchKernelGGL(park_hist_kernel, dim3(gb), dim3(256), 0, h->stream, B, status, kkt, cnt);
    hipLaunchKernelGGL(park_scan_kernel, dim3(1), dim3(64), 0, h->stream, cnt);
    hipLaunchKernelGGL(park_scatter_kernel, dim3(gb), dim3(256), 0, h->stream, B, status, kkt, cnt, h->park_perm);
    pk.mode = 2;
  }
#ifdef MYR_COOP_TRACE
  static int* trace_host = nullptr;
  if (!trace_host) { HIPCHK(hipHostMalloc((void**)&trace_host, 256 * 16 * sizeof(int), hipHostMallocMapped)); }
  for (int i = 0; i < 256 * 16; ++i) trace_host[i] = 0;
  if (co.maxh > 0) {
    int* dev = nullptr; HIPCHK(hipHostGetDevicePointer((void**)&dev, trace_host, 0));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(myriad::g_coop_trace), &dev, sizeof(dev)));
    static std::atomic<int> launch_no{0};
    const int my = ++launch_no; const int ng = (int)grid;
    std::thread([my, ng]() {
      std::this_thread::sleep_for(std::chrono::seconds(8));
      if (launch_no.load() != my) return;
      fprintf(stderr, "[coop trace] launch %d still running after 8 s; per workgroup: t0 seq target mode expect to hid lastcmd hstate ostate quit w1 w2 w3\n", my);
      for (int w = 0; w < ng; ++w) { fprintf(stderr, "  wg %3d:", w); for (int k = 0; k < 16; ++k) fprintf(stderr, " %d", trace_host[w * 16 + k]); fprintf(stderr, "\n"); }
      fflush(stderr);
    }).detach();
  }
#endif
  { const int32_t pl[8] = {1, NWAVES, k1, k1 > 0 ? 2 : 1, slots, co.maxh, 0, 0}; if (!h->plan_frozen) memcpy(h->plan, pl, sizeof(pl)); }      // (the plan of THIS handle's last launch; a restoration twin is a handle of its own)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NWAVES), lds, h->stream, B, h->ticket, o, h->vscale, z, lb, ub, lam, (double*)h->sbuf, stride,
                     params, pstride, cost, status, iters, kkt, h->poison, pk, co);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (co.maxh > 0) {
    int ab = 0;
    HIPCHK(hipMemcpy(&ab, co.abort, sizeof(int), hipMemcpyDeviceToHost));
    if (ab) return fail(MYR_E_HIP, "network kernel: a wait between a trajectory's workgroups ran into its bound (MYRIAD_NODE_HELPERS=0 runs without helper workgroups)");
  }
  if (k1 > 0) {
    if (const char* path = getenv("MYRIAD_PARK_DUMP")) {      // developer knob: the loop scalars of every record as parked ([B][NSCAL] doubles), for the study of resume orders
      std::vector<double> sc((size_t)B * W::NSCAL);
      HIPCHK(hipMemcpy2D(sc.data(), W::NSCAL * sizeof(double), h->park_state, (size_t)pk.stride * sizeof(double), W::NSCAL * sizeof(double), (size_t)B, hipMemcpyDeviceToHost));
      if (FILE* f = fopen(path, "wb")) { fwrite(sc.data(), sizeof(double), sc.size(), f); fclose(f); }
    }
  }
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms;
  kt.launches += 1;
  return MYR_OK;
}
template <class Sys, int SCHEME>
static int launch_hs_fused(myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params,
                           int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                           int32_t* iters, double* kkt) {
  if constexpr (NodeTraits<Sys>::mlp) {
    // network dynamics.  Up to one trajectory per CU: four wavefronts share a trajectory (the LATENCY form: a launch is one solve long, idle CUs attach as
    // helpers).  Beyond: two wavefronts per trajectory, two trajectories per CU (the THROUGHPUT form: one trajectory's sequential sweep and interval
    // passes overlap the other's matrix-core passes on the CU's other two SIMDs; bound multipliers in global scratch so that two workgroups fit the LDS).
    // MYRIAD_FUSED_WAVES=2|4 overrides.
    int waves = (B > device_cus(h)) ? 2 : 4;
    if (h->fused_waves == 2 || h->fused_waves == 4) waves = h->fused_waves;
    if (waves == 2 && pstride == 0 && 2 * HsFused<Sys, 2, 0>::lds_bytes(h->d.intervals) <= 160 * 1024)
      return launch_hs_fused_w<Sys, 2, 0>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
    return launch_hs_fused_w<Sys, 4, 0>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  } else {
    // W = 2: two wavefronts per trajectory for batches of at most two trajectories per CU (a launch then lasts as long as one solve,
    // and the parallel passes of an iteration take half the time: +15 % at B = 512).  Round 3 switched it off -- its build returned
    // results that differed from launch to launch; round 4 found the form of the sweep that does it
    // (hs_solver_fused.h: sweep) and gates every build with tests/test_gpu_poison.py.  MYRIAD_FUSED_WAVES=1|2 overrides the choice.
    int waves = (B <= 2 * device_cus(h)) ? 2 : 1;
    // (Four wavefronts per trajectory for batches of at most one trajectory per CU -- chunks of N / 4 stages -- were built and measured in round 6,
    // tools/dev/exp/exp84.sh: same iterates, and NO gain over two wavefronts -- B = 128 50.6 k against 50.4 k solves/s, B = 256 97.5 k against 96.6 k: what the
    // shorter chunks save, the three interface joins, one after the other on wavefront 0, cost.  Not instantiated: 160 KB of code per system.)
    // With -DMYR_TL_SPEC=1 four wavefronts run two chunks and TWO RUNGS of the inertia ladder at a time (HsFused::TLS) for batches of at most one trajectory per
    // CU: +3 % (exp90.sh), not built by default.
    if (B <= device_cus(h) && HsFused<Sys, 4, SCHEME>::TLS) waves = 4;
    // Systems whose solver LDS lets at most two one-wavefront workgroups onto a CU (ROCKETLANDING's Hermite-Simpson form: 64 KB; the wider twins) leave two
    // SIMDs of every CU idle in that form: when as many two-wavefront workgroups fit, those run at EVERY batch size (round 6, tools/dev/exp/exp96.sh, B = 4096:
    // ROCKETLANDING 94.5 -> 70.4 ms, CARTPOLE's twin 192 -> 146 ms, ROCKETLANDING's twin 397 -> 311 ms; with four one-wavefront workgroups per CU the
    // one-wavefront form wins as before -- BEARPOPULATIONS 9.6 against 14.9 ms).
    // The same holds for every system once the horizon is long enough (exp100.sh, the headline system, B = 4096: N = 150 -- two workgroups per CU in either form --
    // 34.1 -> 27.2 ms; N = 300 -- one -- 130.9 -> 88.2 ms; N = 200, where two one-wavefront workgroups fit but only one of two wavefronts, stays: 46.3 against 57.7 ms).
    if constexpr (HsFused<Sys, 2, SCHEME>::SUPPORTED) {
      const int N = h->d.intervals;
      if (HsFused<Sys, 2, SCHEME>::lds_bytes(N) <= 160 * 1024) {
        int pc1 = 0, pc2 = 0;
        if (int rc = kernel_blocks_per_cu(h, reinterpret_cast<const void*>(hs_solve_fused_kernel<Sys, 1, SCHEME>), 64, HsFused<Sys, 1, SCHEME>::lds_bytes(N), &pc1)) return rc;
        if (int rc = kernel_blocks_per_cu(h, reinterpret_cast<const void*>(hs_solve_fused_kernel<Sys, 2, SCHEME>), 128, HsFused<Sys, 2, SCHEME>::lds_bytes(N), &pc2)) return rc;
        if (pc1 >= 1 && pc1 <= 2 && pc2 >= pc1) waves = 2;
      }
    }
    if (so.park_iter > 0) waves = 1;                  // an explicit two-phase launch: only the one-wavefront form parks (include/myriad_hip.h: park_iter)
    if (h->fused_waves > 0) waves = h->fused_waves;
    if constexpr (HsFused<Sys, 4, SCHEME>::TLS) {
      if (waves == 4 && HsFused<Sys, 4, SCHEME>::lds_bytes(h->d.intervals) <= 160 * 1024)
        return launch_hs_fused_w<Sys, 4, SCHEME>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
    }
    if (waves > 2) waves = 2;
    {
      if (waves == 2 && HsFused<Sys, 2, SCHEME>::lds_bytes(h->d.intervals) <= 160 * 1024)
        return launch_hs_fused_w<Sys, 2, SCHEME>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
    }
    return launch_hs_fused_w<Sys, 1, SCHEME>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  }
}

template <class Sys, int SCHEME = 0>
static int launch_hs_solve(myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params,
                           int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                           int32_t* iters, double* kkt) {
  const int N = h->d.intervals;
  const myr_dims& dm = h->dims;
  // one trajectory per wavefront while its LDS working set fits a CU (N <= ~480 for CARTPOLE); beyond that the
  // lane-per-trajectory form, which keeps everything in global scratch, takes over
  if constexpr (HsFused<Sys, 1, SCHEME>::SUPPORTED && !(SCHEME == 1 && NodeTraits<Sys>::mlp)) {
    if (h->solve_mode == 1 && h->solve_fused && HsFused<Sys, (NodeTraits<Sys>::mlp ? 4 : 1), SCHEME>::lds_bytes(N) <= 160 * 1024)
      return launch_hs_fused<Sys, SCHEME>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  }
  // (Round 3 kept the trapezoidal scheme with more than one control or more than four states off the wavefront kernel: its general
  // sweep read the end point's Hessian record at the Hermite-Simpson stride -- point 2k + 2 instead of k + 1 -- and BEARPOPULATIONS
  // ended in NaN.  Fixed in HsWave::riccati (H_PTS); MYRIAD_TRAP_GENERAL_WAVE=0 sends those systems to the lane form again.)
  static const bool trap_general_wave = [] { const char* e = getenv("MYRIAD_TRAP_GENERAL_WAVE"); return !(e && atoi(e) == 0); }();
  const bool wave_ok = !(SCHEME == 1 && !(Sys::NU == 1 && Sys::NS <= 4)) || trap_general_wave;
  // (ROCKETLANDING's twin, 14 variables per point: round 2's kernel -- minutes of build time, a 16 x 16 factor per lane and stage -- is not built any more
  //  now that the fused kernel serves it; beyond the fused kernel's LDS limit it runs on the lane kernel)
  constexpr bool round2_built = !(Sys::ID >= 100 && Sys::NW > 9 && HsFused<Sys, 1, SCHEME>::SUPPORTED);
  if constexpr (round2_built)
  if (wave_ok)
  if (h->solve_mode == 1 && HsWave<Sys, SCHEME>::lds_bytes(N) <= 160 * 1024) {
    using W = HsWave<Sys, SCHEME>;
    // wavefronts per workgroup: 1, except network systems -- independent solves that share the 40 KB of weights in LDS, four to a
    // workgroup (4 x 24 + 40 KB), or, for batches smaller than the number of CUs, ONE solve whose network passes the four share
    int wpb = 1, coop = 0;
    if (W::MLP) {
      int dev = 0, cus = 256;
      HIPCHK(hipGetDevice(&dev));
      HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      // Small batches: the wavefronts of a workgroup share one trajectory's network passes (cooperative mode) -- four per
      // trajectory up to one trajectory per CU, two up to two per CU (two such workgroups fit a CU), four again up to four per
      // CU (the workgroup takes its trajectories one after the other); beyond that independent solves, four per workgroup.
      // Measured on one box (B: independent / coop 2 / coop 4 ms): 512: 50.8 / 34.5 / 42.3, 1024: 91.3 / 84.6 / 82.5,
      // 2048: 104.4 / 113.5 / 144.8.
      coop = (B <= 4 * cus) ? 1 : 0;
      int coop_wpb = (B > cus && B <= 2 * cus) ? 2 : W::WPB_MAX;
      if (const char* e = getenv("MYRIAD_NODE_COOP")) coop = atoi(e) != 0;
      if (coop || pstride == 0) {
        wpb = coop ? coop_wpb : W::WPB_MAX;
        if (const char* e = getenv("MYRIAD_NODE_WPB")) { wpb = atoi(e); if (wpb < 1) wpb = 1; if (wpb > W::WPB_MAX) wpb = W::WPB_MAX; }
        while (!coop && wpb > 1 && ((size_t)wpb * W::lds_solver_doubles(N) + NodeTraits<Sys>::lds_doubles) * 8 > 160 * 1024) --wpb;
      }
      if (wpb == 1) coop = 0;
    }
    const int lwaves = coop ? 1 : wpb;                          // solves per workgroup
    const size_t lds = ((size_t)lwaves * W::lds_solver_doubles(N) + NodeTraits<Sys>::lds_doubles + (coop ? W::COOP_CMD_DOUBLES : 0)) * 8;
    auto kern = hs_solve_wave_kernel<Sys, SCHEME>;
    int per_cu = 0;
    if (int rc = kernel_blocks_per_cu(h, reinterpret_cast<const void*>(kern), 64 * wpb, lds, &per_cu)) return rc;
    // Persistent form: as many workgroups as the device keeps resident (registers and LDS allow 4 wavefronts per CU), each
    // pulling trajectories from a ticket counter.  Scratch belongs to the SLOT, not to the trajectory: the working set of
    // a launch is slots x 273 KB (280 MB for CARTPOLE N=100) instead of B x 273 KB (1.1 GB at B = 4096) and is re-used
    // trajectory after trajectory, i.e. it stays in the 256 MB Infinity Cache instead of streaming through HBM.
    int slots = h->solve_slots;              // wavefronts
    if (slots <= 0) slots = (per_cu > 0 ? per_cu : 4) * device_cus(h) * lwaves;
    if (slots > B) slots = B;
    slots = (slots + lwaves - 1) / lwaves * lwaves;              // whole workgroups (surplus wavefronts find the ticket counter exhausted)
    long stride = (W::scratch_doubles(N) + 31) / 32 * 32;
    if (((stride / 32) & 1) == 0) stride += 32;          // odd multiple of 256 B: rotate slots over HBM channels
    const size_t need = (size_t)slots * (size_t)stride * 8;
    if (getenv("MYRIAD_DEBUG_PTRS"))
      fprintf(stderr, "[myriad] wave kernel N=%d B=%d slots=%d wpb=%d coop=%d per_cu=%d lds=%zu stride=%ld doubles need=%zu\n", N, B, slots, wpb, coop, per_cu, lds, stride, need);
    if (need > h->sbuf_bytes) {
      if (h->sbuf) HIPCHK(hipFree(h->sbuf));
      h->sbuf = nullptr; h->sbuf_bytes = 0;
      HIPCHK(hipMalloc(&h->sbuf, need));
      h->sbuf_bytes = need;
    }
    if (!h->ticket) HIPCHK(hipMalloc(&h->ticket, sizeof(int)));
    HIPCHK(hipMemsetAsync(h->ticket, 0, sizeof(int), h->stream));
    HsSolveOpts o = make_opts(h, so);
    KTimer& kt = h->kt[MYR_K_SOLVE];
    if (int rc_fill = stack_fill(h)) return rc_fill;
    HIPCHK(hipEventRecord(kt.a, h->stream));
    h->last_solve_form = 1;
    { const int32_t pl[8] = {2, coop ? wpb : 1, 0, 1, slots, 0, 0, 0}; if (!h->plan_frozen) memcpy(h->plan, pl, sizeof(pl)); }
    hipLaunchKernelGGL(kern, dim3((unsigned)(slots / lwaves)), dim3(64 * wpb), lds, h->stream, B, h->ticket, o, h->vscale, z, lb, ub, lam, (double*)h->sbuf, stride,
                       params, pstride, cost, status, iters, kkt, coop, h->poison);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(kt.b, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
    kt.sum_ms += ms;
    kt.launches += 1;
    return MYR_OK;
  }
  return solve_lane_colloc_for_system<Sys, SCHEME>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
}

// lane-per-trajectory path, any sweep core (Hermite-Simpson, trapezoidal, shooting)
template <class Core, class Sys>
static int launch_lane_solve(myr_handle h, int B, long nst, double* z, const double* lb, const double* ub, const double* params,
                             int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                             int32_t* iters, double* kkt) {
  const myr_dims& dm = h->dims;
  // batch-minor leading dimension: a multiple of 64 lanes, but an ODD multiple so that consecutive elements of a
  // trajectory (stride Bp*8 bytes) rotate over HBM channels / L2 sets instead of camping on one (power-of-two stride)
  long Bp = ((long)B + 63) / 64 * 64;
  if (((Bp / 64) & 1) == 0) Bp += 64;
  const long n = dm.n, m = dm.m;
  const size_t need = (size_t)Bp * (size_t)(6 * n + m + nst) * 8;
  if (need > h->sbuf_bytes) {
    if (h->sbuf) HIPCHK(hipFree(h->sbuf));
    h->sbuf = nullptr; h->sbuf_bytes = 0;
    HIPCHK(hipMalloc(&h->sbuf, need));
    h->sbuf_bytes = need;
  }
  double* sz = (double*)h->sbuf;
  double* slb = sz + n * Bp;
  double* sub = slb + n * Bp;
  double* szL = sub + n * Bp;
  double* szU = szL + n * Bp;
  double* sdz = szU + n * Bp;
  double* slam = sdz + n * Bp;
  double* sst = slam + m * Bp;
  // padded lanes read garbage-free memory
  HIPCHK(hipMemsetAsync(h->sbuf, 0, (size_t)Bp * (size_t)(3 * n) * 8, h->stream));
  dim3 tb(256), tg((unsigned)((n + 31) / 32), (unsigned)((B + 31) / 32));
  hipLaunchKernelGGL(transpose_kernel, tg, tb, 0, h->stream, (const double*)z, sz, B, (int)n, Bp);
  hipLaunchKernelGGL(transpose_kernel, tg, tb, 0, h->stream, lb, slb, B, (int)n, Bp);
  hipLaunchKernelGGL(transpose_kernel, tg, tb, 0, h->stream, ub, sub, B, (int)n, Bp);
  HsSolveOpts o = make_opts(h, so);
  KTimer& kt = h->kt[MYR_K_SOLVE];
  if (int rc_fill = stack_fill(h)) return rc_fill;
  HIPCHK(hipEventRecord(kt.a, h->stream));
  const int lpw = h->solve_lpw;
  h->last_solve_form = 0;
  { const int32_t pl[8] = {0, 0, 0, 1, B, 0, 0, 0}; if (!h->plan_frozen) memcpy(h->plan, pl, sizeof(pl)); }
  hipLaunchKernelGGL((lane_solve_kernel<Core, Sys>), dim3((unsigned)((B + lpw - 1) / lpw)), dim3(64), 0, h->stream, B, Bp, lpw, o, h->vscale, sz, slb, sub, szL,
                     szU, slam, sdz, sst, params, pstride, cost, status, iters, kkt);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  hipLaunchKernelGGL(transpose_back_kernel, tg, tb, 0, h->stream, (const double*)sz, z, B, (int)n, Bp);
  if (lam) {
    dim3 tgl((unsigned)((m + 31) / 32), (unsigned)((B + 31) / 32));
    hipLaunchKernelGGL(transpose_back_kernel, tgl, tb, 0, h->stream, (const double*)slam, lam, B, (int)m, Bp);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms;
  kt.launches += 1;
  return MYR_OK;
}

// shooting, one trajectory per wavefront (shoot_solver_wave.h) while the iterate fits the LDS of a CU; else the lane form
template <class Sys, int M>
static int launch_shoot_solve(myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params,
                              int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                              int32_t* iters, double* kkt) {
  const int N = h->d.intervals, cpi = h->d.controls_per_interval;
  using W = ShootWave<Sys, M>;
  const size_t lds = W::lds_bytes(N, cpi);
  if (h->solve_mode != 1 || lds > 160 * 1024)
    return launch_lane_solve<ShootCore<Sys, M>, Sys>(h, B, ShootCore<Sys, M>::stage_doubles(N, cpi), z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  auto kern = shoot_solve_wave_kernel<Sys, M>;
  int per_cu = 0;
  if (int rc = kernel_blocks_per_cu(h, reinterpret_cast<const void*>(kern), 64, lds, &per_cu)) return rc;
  int slots = h->solve_slots;
  if (slots <= 0) slots = (per_cu > 0 ? per_cu : 4) * device_cus(h);
  if (slots > B) slots = B;
  if (!h->ticket) HIPCHK(hipMalloc(&h->ticket, sizeof(int)));
  HIPCHK(hipMemsetAsync(h->ticket, 0, sizeof(int), h->stream));
  HsSolveOpts o = make_opts(h, so);
  // network systems: records, activations and tangents of the matrix-core passes, per resident workgroup (shoot_solver_wave.h: mlp_scratch_doubles)
  long sstride = (W::mlp_scratch_doubles(N, cpi) + 31) / 32 * 32;
  if (sstride > 0) {
    if (((sstride / 32) & 1) == 0) sstride += 32;
    const size_t need = (size_t)slots * (size_t)sstride * 8;
    if (need > h->sbuf_bytes) {
      if (h->sbuf) HIPCHK(hipFree(h->sbuf));
      h->sbuf = nullptr; h->sbuf_bytes = 0;
      HIPCHK(hipMalloc(&h->sbuf, need));
      h->sbuf_bytes = need;
    }
  }
  KTimer& kt = h->kt[MYR_K_SOLVE];
  if (int rc_fill = stack_fill(h)) return rc_fill;
  HIPCHK(hipEventRecord(kt.a, h->stream));
  h->last_solve_form = 1;
  { const int32_t pl[8] = {3, 1, 0, 1, slots, 0, 0, 0}; if (!h->plan_frozen) memcpy(h->plan, pl, sizeof(pl)); }
  hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3(64), lds, h->stream, B, h->ticket, o, h->vscale, z, lb, ub, lam, params, pstride,
                     cost, status, iters, kkt, h->poison, sstride > 0 ? (double*)h->sbuf : (double*)nullptr, sstride);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms;
  kt.launches += 1;
  return MYR_OK;
}

// One function template per transcription, so that the build can give each its own object (MYR_TU_PART below: the solver kernels of a wide system
// are minutes of compile time per scheme).
template <class Sys, int SCHEME>
int solve_lane_colloc_for_system(MYR_SOLVE_ARGS) {
  const int N = h->d.intervals;
  if constexpr (SCHEME == 1) {
    if constexpr (Sys::PARAMS_BY_POINTER) return fail(MYR_E_UNSUPPORTED, "myr_solve: NODE systems are built for HERMITE_SIMPSON");
    else return launch_lane_solve<TrapCore<Sys>, Sys>(h, B, TrapCore<Sys>::stage_doubles(N), z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  } else
    return launch_lane_solve<HsSolver<Sys>, Sys>(h, B, HsSol<Sys>::stage_doubles(N), z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
}
template <class Sys>
int solve_hs_for_system(MYR_SOLVE_ARGS) {
  return launch_hs_solve<Sys>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
}
template <class Sys>
int solve_trap_for_system(MYR_SOLVE_ARGS) {      // wavefront form (falls back to the lane form for MYRIAD_SOLVE_MODE=lane / very large N)
  if constexpr (Sys::PARAMS_BY_POINTER) return fail(MYR_E_UNSUPPORTED, "myr_solve: NODE systems are built for HERMITE_SIMPSON");
  // (a twin too wide for the fused kernel's block sweep would cost minutes of build time per solver on round 2's kernel: refused; no twin is, since round 5)
  else if constexpr (Sys::ID >= 100 && Sys::NW > 9 && !HsFused<Sys, 1, 1>::SUPPORTED) return fail(MYR_E_UNSUPPORTED, "myr_solve: the trapezoidal solver of this elastic twin is not built");
  else return launch_hs_solve<Sys, 1>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
}
template <class Sys>
int solve_shoot_for_system(MYR_SOLVE_ARGS) {
  // elastic twins (id >= 100) exist for the collocation solvers, whose restoration device they are; their shooting solver is not built
  if constexpr (Sys::ID >= 100) return fail(MYR_E_UNSUPPORTED, "myr_solve: elastic twins are built for the collocation transcriptions");
  else {
    if (h->d.integration_method == MYR_INT_RK4)
      return launch_shoot_solve<Sys, 2>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
    return launch_shoot_solve<Sys, 1>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  }
}
#if defined(MYR_TU_SYSTEM) && defined(MYR_TU_PART)      // the scheme solvers this object does not hold: no implicit instantiation
#if MYR_TU_PART != 1
extern template int solve_hs_for_system<myriad::MYR_TU_SYSTEM>(MYR_SOLVE_ARG_TYPES);
#endif
#if MYR_TU_PART != 3
extern template int solve_trap_for_system<myriad::MYR_TU_SYSTEM>(MYR_SOLVE_ARG_TYPES);
#endif
#if MYR_TU_PART != 4
extern template int solve_shoot_for_system<myriad::MYR_TU_SYSTEM>(MYR_SOLVE_ARG_TYPES);
#endif
#endif
template <class Sys>
int solve_for_system(MYR_SOLVE_ARGS) {
  switch (h->d.transcription) {
    case MYR_TR_HERMITE_SIMPSON: return solve_hs_for_system<Sys>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
    case MYR_TR_TRAPEZOIDAL: return solve_trap_for_system<Sys>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
    case MYR_TR_SHOOTING: return solve_shoot_for_system<Sys>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  }
  return fail(MYR_E_ARG, "solve: unknown transcription");
}

// ------------------------------------------------------------------------------------------------
// rollout
// ------------------------------------------------------------------------------------------------
template <class Sys>
__global__ __launch_bounds__(64)
void rollout_kernel(int B, int method, int num_steps, double h, int u_rows, const double* __restrict__ x0,
                    const double* __restrict__ us, const double* __restrict__ params, int params_stride,
                    double* __restrict__ xs, double* __restrict__ cost) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  SysParams<Sys> pp;
  pp.load(params, b, params_stride);
  const double* p = pp.get();
  const double c = Rollout<Sys>::run(method, num_steps, h, u_rows, x0 + b * Sys::NS, us + b * (long)u_rows * Sys::NU, p,
                                     xs ? xs + b * (long)(num_steps + 1) * Sys::NS : nullptr);
  if (cost) cost[b] = c;
}

// ------------------------------------------------------------------------------------------------
// batched Forward-Backward Sweep (indirect method)
// ------------------------------------------------------------------------------------------------
template <class Sys>
int launch_fbsm(myr_handle h, int B, long Bp, int N, const double* x0, const double* adjT, const double* params, int pstride,
                       const VarScale& lo, const VarScale& hi, double bang, double delta, int max_sweeps, double* X, double* U, double* A, int32_t* sweeps) {
  if constexpr (!Indirect<Sys>::SUPPORTED) {
    return fail(MYR_E_UNSUPPORTED, "myr_fbsm: this system has no adjoint dynamics (not an IndirectFHCS on the path)");
  } else {
    KTimer& kt = h->kt[MYR_K_FBSM];
    if (int rc_fill = stack_fill(h)) return rc_fill;
    HIPCHK(hipEventRecord(kt.a, h->stream));
    hipLaunchKernelGGL(fbsm_kernel<Sys>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, B, Bp, N, h->d.T, x0, adjT, params,
                       pstride, lo, hi, bang, delta, max_sweeps, X, U, A, sweeps);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(kt.b, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
    kt.sum_ms += ms;
    kt.launches += 1;
    return MYR_OK;
  }
}

template <class Sys>
int rollout_for_system(myr_handle h, int B, int num_steps, int u_rows, const double* x0, const double* us, const double* params,
                       int pstride, double* xs, double* cost) {
  hipLaunchKernelGGL(rollout_kernel<Sys>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, B, (int)h->d.integration_method, num_steps,
                     h->d.T / num_steps, u_rows, x0, us, params, pstride, xs, cost);
  return MYR_OK;
}

// ---- per-system entry points: explicit instantiation (system objects) / extern declaration (main object) -----------------
#define MYR_SYSTEM_ENTRY_POINTS(LINK, S)                                                                                          \
  LINK template int eval_for_system<S>(myr_handle, int, const double*, const double*, int, double*, double*, double*, double*);   \
  LINK template int products_for_system<S>(myr_handle, const ProdArgs&);                                                          \
  LINK template int solve_for_system<S>(myr_handle, int, double*, const double*, const double*, const double*, int,               \
                                        const myr_solve_opts&, double*, double*, int32_t*, int32_t*, double*);                    \
  LINK template int rollout_for_system<S>(myr_handle, int, int, int, const double*, const double*, const double*, int, double*, double*); \
  LINK template int launch_fbsm<S>(myr_handle, int, long, int, const double*, const double*, const double*, int, const VarScale&,  \
                                   const VarScale&, double, double, int, double*, double*, double*, int32_t*);
#if defined(MYR_TU_SYSTEM) && defined(MYR_TU_PART)
// a system's object in parts (the build splits the slow ones): -DMYR_TU_PART=1 the Hermite-Simpson wavefront solvers and the dispatch, 3 the trapezoidal
// wavefront solvers, 4 the shooting solvers, 5 / 6 the lane kernels of the two collocation solvers, 2 everything else
#if MYR_TU_PART == 1
template int solve_hs_for_system<myriad::MYR_TU_SYSTEM>(MYR_SOLVE_ARG_TYPES);
template int solve_for_system<myriad::MYR_TU_SYSTEM>(MYR_SOLVE_ARG_TYPES);
#elif MYR_TU_PART == 3
template int solve_trap_for_system<myriad::MYR_TU_SYSTEM>(MYR_SOLVE_ARG_TYPES);
#elif MYR_TU_PART == 4
template int solve_shoot_for_system<myriad::MYR_TU_SYSTEM>(MYR_SOLVE_ARG_TYPES);
#elif MYR_TU_PART == 5
template int solve_lane_colloc_for_system<myriad::MYR_TU_SYSTEM, 0>(MYR_SOLVE_ARG_TYPES);
#elif MYR_TU_PART == 6
template int solve_lane_colloc_for_system<myriad::MYR_TU_SYSTEM, 1>(MYR_SOLVE_ARG_TYPES);
#else
#define MYR_SYSTEM_ENTRY_POINTS_REST(S)                                                                                            \
  template int eval_for_system<S>(myr_handle, int, const double*, const double*, int, double*, double*, double*, double*);         \
  template int products_for_system<S>(myr_handle, const ProdArgs&);                                                                \
  template int rollout_for_system<S>(myr_handle, int, int, int, const double*, const double*, const double*, int, double*, double*); \
  template int launch_fbsm<S>(myr_handle, int, long, int, const double*, const double*, const double*, int, const VarScale&,        \
                              const VarScale&, double, double, int, double*, double*, double*, int32_t*);
MYR_SYSTEM_ENTRY_POINTS_REST(myriad::MYR_TU_SYSTEM)
#endif
#elif defined(MYR_TU_SYSTEM)
MYR_SYSTEM_ENTRY_POINTS(, myriad::MYR_TU_SYSTEM)
#elif defined(MYR_TU_MAIN)
#define X(N) MYR_SYSTEM_ENTRY_POINTS(extern, Sys##N)
MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
MYR_SYSTEM_ENTRY_POINTS(extern, SysNODE_CARTPOLE)
#endif

#if !defined(MYR_TU_SYSTEM)   // ===== C-ABI and dispatch: the main object only ============================================
extern "C" const char* myr_last_error(void) { return g_err.c_str(); }
extern "C" const char* myr_version(void) { return "myriad_hip 0.2 (gfx950)"; }
extern "C" int32_t myr_abi_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(myr_solve_opts);
    case 1: return (int32_t)sizeof(myr_problem_desc);
    case 2: return (int32_t)sizeof(myr_dims);
  }
  return -1;
}
extern "C" int myr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n;
}

extern "C" void myr_default_solve_opts(myr_solve_opts* o) {
  if (!o) return;
  o->max_iter = 1000;   // config.py:70
  o->restarts = -1;     // library default, see the header
  o->tol_feas = 1e-8;
  o->tol_stat = 1e-6;
  o->tol_compl = 1e-7;
  o->mu_init = 0.1;
  o->restoration = -1;  // library default: elastic phase + second starts, see the header
  o->park_iter = 0;     // the library decides, see the header
}

extern "C" int myr_create(const myr_problem_desc* desc, myr_handle* out) {
  if (!desc || !out) return fail(MYR_E_ARG, "myr_create: null argument");
  SysInfo si;
  if (!sys_info(desc->system_id, &si)) return fail(MYR_E_ARG, "myr_create: unknown system_id");
  if (desc->intervals < 1 || desc->controls_per_interval < 1 || !(desc->T > 0.0))
    return fail(MYR_E_ARG, "myr_create: intervals, controls_per_interval and T must be positive");
  myr_dims dm;
  memset(&dm, 0, sizeof(dm));
  dm.ns = si.ns; dm.nu = si.nu; dm.np = si.np;
  const int N = desc->intervals;
  switch (desc->transcription) {
    case MYR_TR_HERMITE_SIMPSON: {
      const int K = 2 * N + 1;
      dm.x_rows = K; dm.u_rows = K;
      dm.n = K * (si.ns + si.nu);
      dm.m = 2 * N * si.ns;
      dm.jblk = N * (5 * si.ns * si.ns + 5 * si.ns * si.nu);
      dm.ngrad = si.cost_dep_x ? dm.n : K * si.nu;
      break;
    }
    case MYR_TR_TRAPEZOIDAL: {
      dm.x_rows = N + 1; dm.u_rows = N + 1;
      dm.n = (N + 1) * (si.ns + si.nu);
      dm.m = N * si.ns;
      dm.jblk = N * (2 * si.ns * si.ns + 2 * si.ns * si.nu);
      dm.ngrad = si.cost_dep_x ? dm.n : (N + 1) * si.nu;
      break;
    }
    case MYR_TR_SHOOTING: {
      const int mc = desc->integration_method == MYR_INT_RK4 ? 2 : 1;
      dm.x_rows = N + 1; dm.u_rows = mc * N * desc->controls_per_interval + 1;
      dm.n = dm.x_rows * si.ns + dm.u_rows * si.nu;
      dm.m = N * si.ns;
      dm.jblk = N * (si.ns * si.ns + si.ns * (mc * desc->controls_per_interval + 1) * si.nu);   // Ju spans the interval's mc*cpi+1 control rows
      dm.ngrad = dm.n;
      break;
    }
    default:
      return fail(MYR_E_ARG, "myr_create: unknown transcription");
  }
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (desc->device < 0 || desc->device >= ndev) return fail(MYR_E_ARG, "myr_create: bad device ordinal");
  HIPCHK(hipSetDevice(desc->device));
  myr_handle h = new myr_handle_s();
  h->d = *desc;
  h->dims = dm;
  h->si = si;
  HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  for (int i = 0; i < MYR_K_COUNT; ++i) {
    HIPCHK(hipEventCreate(&h->kt[i].a));
    HIPCHK(hipEventCreate(&h->kt[i].b));
  }
  const char* w = getenv("MYRIAD_EVAL_WPT");
  if (w) { int v = atoi(w); if (v == 1 || v == 4 || v == 8) h->eval_wpt = v; }
  if (const char* e = getenv("MYRIAD_EVAL_NT")) h->eval_nt = atoi(e);
  const char* md = getenv("MYRIAD_SOLVE_MODE");
  if (md) { h->solve_mode = (strcmp(md, "lane") == 0) ? 0 : 1; h->solve_fused = (strcmp(md, "wave1") == 0) ? 0 : 1; }
  const char* l = getenv("MYRIAD_SOLVE_LPW");
  if (l) { int v = atoi(l); if (v >= 1 && v <= 64) h->solve_lpw = v; }
  if (const char* e = getenv("MYRIAD_REG_FILL")) h->reg_fill = !strcmp(e, "nan") ? 1 : (!strcmp(e, "random") ? 2 : (!strcmp(e, "big") ? 3 : (!strcmp(e, "zero") ? 4 : 0)));
  if (const char* e = getenv("MYRIAD_STACK_FILL")) h->stack_fill = !strcmp(e, "nan") ? 1 : (!strcmp(e, "random") ? 2 : (!strcmp(e, "big") ? 3 : (!strcmp(e, "zero") ? 4 : 0)));
  if (const char* e = getenv("MYRIAD_NODE_HELPERS")) h->node_helpers = atoi(e);   // developer knob: helper workgroups per trajectory of the network kernel (0 = none)
  if (const char* e = getenv("MYRIAD_PARK_ITER")) h->park_iter = atoi(e);         // developer knob: iterations of phase 1 of the two-phase launch (0 = off)
  if (const char* e = getenv("MYRIAD_FUSED_WAVES")) h->fused_waves = atoi(e);     // developer knob: wavefronts per trajectory of the fused kernel
  if (const char* e = getenv("MYRIAD_SOLVE_SLOTS")) h->solve_slots = atoi(e);   // developer knob: resident wavefronts of the solve kernel
  if (const char* e = getenv("MYRIAD_POISON")) {      // test knob: "nan" (signalling NaN), "big", "random", or a 64-bit pattern in hex
    if (strcmp(e, "nan") == 0) h->poison = 0x7ff4dead0000beefULL;
    else if (strcmp(e, "big") == 0) h->poison = 0x4415af1d78b58c40ULL;      // 1e20
    else if (strcmp(e, "random") == 0) h->poison = MYR_POISON_RANDOM;       // finite values of order one, different in every word
    else h->poison = strtoull(e, nullptr, 16);
  }
  *out = h;
  return MYR_OK;
}

extern "C" int myr_destroy(myr_handle h) {
  if (!h) return MYR_OK;
  (void)hipSetDevice(h->d.device);
  if (h->dbuf) (void)hipFree(h->dbuf);
  if (h->sbuf) (void)hipFree(h->sbuf);
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->park_state) (void)hipFree(h->park_state);
  if (h->park_perm) (void)hipFree(h->park_perm);
  if (h->vbuf) (void)hipFree(h->vbuf);
  if (h->rbuf) (void)hipFree(h->rbuf);
  if (h->fbuf) (void)hipFree(h->fbuf);
  if (h->nfail_host) (void)hipHostFree(h->nfail_host);
  if (h->nfail_dev) (void)hipFree(h->nfail_dev);
  if (h->coop_buf) (void)hipFree(h->coop_buf);
  if (h->twin) { (void)myr_destroy(h->twin); h->twin = nullptr; }
  for (int i = 0; i < MYR_K_COUNT; ++i) {
    if (h->kt[i].a) (void)hipEventDestroy(h->kt[i].a);
    if (h->kt[i].b) (void)hipEventDestroy(h->kt[i].b);
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return MYR_OK;
}

extern "C" int myr_get_dims(myr_handle h, myr_dims* out) {
  if (!h || !out) return fail(MYR_E_ARG, "myr_get_dims: null argument");
  *out = h->dims;
  return MYR_OK;
}

extern "C" int myr_kernel_time(myr_handle h, int32_t kernel_id, double* avg_ms, int32_t* launches) {
  if (!h || kernel_id < 0 || kernel_id >= MYR_K_COUNT) return fail(MYR_E_ARG, "myr_kernel_time: bad argument");
  const KTimer& k = h->kt[kernel_id];
  if (avg_ms) *avg_ms = k.launches ? k.sum_ms / k.launches : 0.0;
  if (launches) *launches = k.launches;
  return MYR_OK;
}

extern "C" int myr_kernel_time_reset(myr_handle h) {
  if (!h) return fail(MYR_E_ARG, "myr_kernel_time_reset: null handle");
  for (int i = 0; i < MYR_K_COUNT; ++i) { h->kt[i].sum_ms = 0.0; h->kt[i].launches = 0; }
  return MYR_OK;
}

// systems without a direct-transcription path: INVASIVEPLANT is discrete-time (the reference refuses it in its direct
// optimisers too, trajectory_optimizers/base.py:66-67) and only has the discrete FBSM
static int no_such_path(myr_handle h, const char* who) {
  if (h->d.system_id == MYR_SYS_INVASIVEPLANT)
    return fail(MYR_E_UNSUPPORTED, std::string(who) + ": INVASIVEPLANT is a discrete-time system; only myr_fbsm is available for it");
  return fail(MYR_E_ARG, std::string(who) + ": unknown system");
}

static int dispatch_eval(myr_handle h, int B, const double* z, const double* params, int pstride,
                         double* f, double* g, double* c, double* j) {
  switch (h->d.system_id) {
#define X(N) case MYR_SYS_##N: return eval_for_system<Sys##N>(h, B, z, params, pstride, f, g, c, j);
    MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
    case MYR_SYS_NODE_CARTPOLE: return eval_for_system<SysNODE_CARTPOLE>(h, B, z, params, pstride, f, g, c, j);
  }
  return no_such_path(h, "eval");
}

extern "C" int myr_eval(myr_handle h, int32_t B, const double* z, const double* params, int32_t params_stride,
                        double* f, double* gradf, double* c, double* jblk, int32_t mem) {
  if (!h || !z) return fail(MYR_E_ARG, "myr_eval: null handle or z");
  if (h->d.system_id == MYR_SYS_INVASIVEPLANT) return no_such_path(h, "myr_eval");
  if (B < 0) return fail(MYR_E_ARG, "myr_eval: negative batch");
  if (B == 0) return MYR_OK;
  if (params && params_stride != 0 && params_stride != h->dims.np)
    return fail(MYR_E_ARG, "myr_eval: params_stride must be 0 (shared) or np");
  if (!params && h->d.system_id == MYR_SYS_NODE_CARTPOLE) return fail(MYR_E_ARG, "myr_eval: a NODE system needs its weights in `params`");
  HIPCHK(hipSetDevice(h->d.device));
  const myr_dims& dm = h->dims;
  if (mem == MYR_MEM_DEVICE) return dispatch_eval(h, B, z, params, params_stride, f, gradf, c, jblk);
  if (mem != MYR_MEM_HOST) return fail(MYR_E_ARG, "myr_eval: bad mem kind");
  // host pointers: stage through device scratch
  const size_t nz = (size_t)B * dm.n, npar = params ? (params_stride ? (size_t)B * dm.np : (size_t)dm.np) : 0;
  const size_t nf = f ? (size_t)B : 0, ng = gradf ? (size_t)B * dm.ngrad : 0;
  const size_t nc = c ? (size_t)B * dm.m : 0, nj = jblk ? (size_t)B * dm.jblk : 0;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };   // keep 16-byte alignment of every carve
  const size_t total = al(nz) + al(npar) + al(nf) + al(ng) + al(nc) + al(nj);
  int rc = ensure_dbuf(h, total * 8);
  if (rc) return rc;
  double* dz = (double*)h->dbuf;
  double* dp = dz + al(nz);
  double* df = dp + al(npar);
  double* dg = df + al(nf);
  double* dc = dg + al(ng);
  double* dj = dc + al(nc);
  HIPCHK(hipMemcpyAsync(dz, z, nz * 8, hipMemcpyHostToDevice, h->stream));
  if (npar) HIPCHK(hipMemcpyAsync(dp, params, npar * 8, hipMemcpyHostToDevice, h->stream));
  rc = dispatch_eval(h, B, dz, npar ? dp : nullptr, params_stride, nf ? df : nullptr, ng ? dg : nullptr,
                     nc ? dc : nullptr, nj ? dj : nullptr);
  if (rc) return rc;
  if (nf) HIPCHK(hipMemcpyAsync(f, df, nf * 8, hipMemcpyDeviceToHost, h->stream));
  if (ng) HIPCHK(hipMemcpyAsync(gradf, dg, ng * 8, hipMemcpyDeviceToHost, h->stream));
  if (nc) HIPCHK(hipMemcpyAsync(c, dc, nc * 8, hipMemcpyDeviceToHost, h->stream));
  if (nj) HIPCHK(hipMemcpyAsync(jblk, dj, nj * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MYR_OK;
}

static int dispatch_products(myr_handle h, const ProdArgs& a) {
  switch (h->d.system_id) {
#define X(N) case MYR_SYS_##N: return products_for_system<Sys##N>(h, a);
    MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
    case MYR_SYS_NODE_CARTPOLE: return products_for_system<SysNODE_CARTPOLE>(h, a);
  }
  return no_such_path(h, "products");
}

static int check_products_args(myr_handle h, const char* who, int32_t B, const void* p0, const void* p1, const void* p2,
                               const double* params, int32_t params_stride) {
  if (!h || !p0 || !p1 || !p2) return fail(MYR_E_ARG, std::string(who) + ": null handle or array");
  if (h->d.system_id == MYR_SYS_INVASIVEPLANT) return no_such_path(h, who);
  if (B < 0) return fail(MYR_E_ARG, std::string(who) + ": negative batch");
  if (params && params_stride != 0 && params_stride != h->dims.np)
    return fail(MYR_E_ARG, std::string(who) + ": params_stride must be 0 (shared) or np");
  if (!params && h->d.system_id == MYR_SYS_NODE_CARTPOLE) return fail(MYR_E_ARG, std::string(who) + ": a NODE system needs its weights in `params`");
  return MYR_OK;
}

// one implementation for J^T lam / grad L (in_w = m, out = n) and J v (in_w = n, out = m)
static int products_call(myr_handle h, int op, const char* who, int32_t B, const double* z, const double* w, const double* params,
                         int32_t params_stride, double* out, int32_t add_gradf, int32_t mem) {
  int rc = check_products_args(h, who, B, z, w, out, params, params_stride);
  if (rc) return rc;
  if (B == 0) return MYR_OK;
  HIPCHK(hipSetDevice(h->d.device));
  const myr_dims& dm = h->dims;
  ProdArgs a{};
  a.op = op; a.B = B; a.pstride = params_stride; a.add_gradf = add_gradf;
  if (mem == MYR_MEM_DEVICE) { a.z = z; a.w = w; a.params = params; a.out = out; return dispatch_products(h, a); }
  if (mem != MYR_MEM_HOST) return fail(MYR_E_ARG, std::string(who) + ": bad mem kind");
  const size_t nz = (size_t)B * dm.n, nw = (size_t)B * (op == PRODOP_VJP ? dm.m : dm.n), no = (size_t)B * (op == PRODOP_VJP ? dm.n : dm.m);
  const size_t npar = params ? (params_stride ? (size_t)B * dm.np : (size_t)dm.np) : 0;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };
  rc = ensure_dbuf(h, (al(nz) + al(nw) + al(no) + al(npar)) * 8);
  if (rc) return rc;
  double* dz = (double*)h->dbuf; double* dw = dz + al(nz); double* dout = dw + al(nw); double* dp = dout + al(no);
  HIPCHK(hipMemcpyAsync(dz, z, nz * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dw, w, nw * 8, hipMemcpyHostToDevice, h->stream));
  if (npar) HIPCHK(hipMemcpyAsync(dp, params, npar * 8, hipMemcpyHostToDevice, h->stream));
  a.z = dz; a.w = dw; a.params = npar ? dp : nullptr; a.out = dout;
  rc = dispatch_products(h, a);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(out, dout, no * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MYR_OK;
}

extern "C" int myr_vjp(myr_handle h, int32_t B, const double* z, const double* lam, const double* params, int32_t params_stride,
                       double* out, int32_t add_gradf, int32_t mem) {
  return products_call(h, PRODOP_VJP, "myr_vjp", B, z, lam, params, params_stride, out, add_gradf, mem);
}

extern "C" int myr_jvp(myr_handle h, int32_t B, const double* z, const double* v, const double* params, int32_t params_stride,
                       double* out, int32_t mem) {
  return products_call(h, PRODOP_JVP, "myr_jvp", B, z, v, params, params_stride, out, 0, mem);
}

extern "C" int myr_exgd(myr_handle h, int32_t B, double* z, double* lam, const double* lb, const double* ub, const double* params,
                        int32_t params_stride, double eta_x, double eta_v, int32_t nsteps, int32_t mem) {
  int rc = check_products_args(h, "myr_exgd", B, z, lam, lb, params, params_stride);
  if (rc) return rc;
  if (!ub) return fail(MYR_E_ARG, "myr_exgd: null ub");
  if (nsteps < 0) return fail(MYR_E_ARG, "myr_exgd: negative nsteps");
  if (B == 0 || nsteps == 0) return MYR_OK;
  HIPCHK(hipSetDevice(h->d.device));
  const myr_dims& dm = h->dims;
  ProdArgs a{};
  a.op = PRODOP_EXGD; a.B = B; a.pstride = params_stride; a.eta_x = eta_x; a.eta_v = eta_v; a.nsteps = nsteps;
  if (mem == MYR_MEM_DEVICE) { a.zio = z; a.lamio = lam; a.lb = lb; a.ub = ub; a.params = params; return dispatch_products(h, a); }
  if (mem != MYR_MEM_HOST) return fail(MYR_E_ARG, "myr_exgd: bad mem kind");
  const size_t nz = (size_t)B * dm.n, nl = (size_t)B * dm.m;
  const size_t npar = params ? (params_stride ? (size_t)B * dm.np : (size_t)dm.np) : 0;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };
  rc = ensure_dbuf(h, (3 * al(nz) + al(nl) + al(npar)) * 8);
  if (rc) return rc;
  double* dz = (double*)h->dbuf; double* dlb = dz + al(nz); double* dub = dlb + al(nz); double* dl = dub + al(nz); double* dp = dl + al(nl);
  HIPCHK(hipMemcpyAsync(dz, z, nz * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dlb, lb, nz * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dub, ub, nz * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dl, lam, nl * 8, hipMemcpyHostToDevice, h->stream));
  if (npar) HIPCHK(hipMemcpyAsync(dp, params, npar * 8, hipMemcpyHostToDevice, h->stream));
  a.zio = dz; a.lamio = dl; a.lb = dlb; a.ub = dub; a.params = npar ? dp : nullptr;
  rc = dispatch_products(h, a);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(z, dz, nz * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(lam, dl, nl * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MYR_OK;
}

static int dispatch_solve(myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params,
                          int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                          int32_t* iters, double* kkt) {
  switch (h->d.system_id) {
#define X(N) case MYR_SYS_##N: return solve_for_system<Sys##N>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
    MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
    case MYR_SYS_NODE_CARTPOLE: return solve_for_system<SysNODE_CARTPOLE>(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  }
  return no_such_path(h, "solve");
}

// ---- variable scaling of the solve path ----------------------------------------------------------------------
// z / lb / ub -> scaled variables (z in place, bounds into handle scratch); afterwards z back and lam / s_state
__global__ __launch_bounds__(256)
void scale_in_kernel(long total, int n, int xcount, int ns, int nu, VarScale vs, double* z, const double* __restrict__ lb,
                     const double* __restrict__ ub, double* lbs, double* ubs) {
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    const int i = (int)(g % n);
    const double inv = 1.0 / vs.s[i < xcount ? i % ns : ns + (i - xcount) % nu];
    z[g] *= inv; lbs[g] = lb[g] * inv; ubs[g] = ub[g] * inv;
  }
}
__global__ __launch_bounds__(256)
void scale_out_kernel(long total_z, long total_l, int n, int xcount, int ns, int nu, int m, VarScale vs, double* z, double* lam) {
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total_z + total_l; g += (long)gridDim.x * blockDim.x) {
    if (g < total_z) {
      const int i = (int)(g % n);
      z[g] *= vs.s[i < xcount ? i % ns : ns + (i - xcount) % nu];
    } else if (lam) {
      const long q = g - total_z;
      lam[q] /= vs.s[(int)((q % m) % ns)];       // every constraint row is a state-component row (defect / interpolation)
    }
  }
}

static int dispatch_solve_scaled(myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params,
                                 int pstride, const myr_solve_opts& so, double* lam, double* cost, int32_t* status,
                                 int32_t* iters, double* kkt) {
  if (!h->vscale_on) return dispatch_solve(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  const myr_dims& dm = h->dims;
  const size_t need = 2 * (size_t)B * dm.n * 8;
  if (need > h->vbuf_bytes) {
    if (h->vbuf) HIPCHK(hipFree(h->vbuf));
    h->vbuf = nullptr; h->vbuf_bytes = 0;
    HIPCHK(hipMalloc(&h->vbuf, need));
    h->vbuf_bytes = need;
  }
  double* lbs = (double*)h->vbuf;
  double* ubs = lbs + (size_t)B * dm.n;
  const long tz = (long)B * dm.n, tl = lam ? (long)B * dm.m : 0;
  const int xcount = dm.x_rows * dm.ns;
  long blocks = (tz + 255) / 256; if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(scale_in_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, tz, dm.n, xcount, dm.ns, dm.nu, h->vscale, z, lb, ub, lbs, ubs);
  HIPCHK(hipGetLastError());
  int rc = dispatch_solve(h, B, z, lbs, ubs, params, pstride, so, lam, cost, status, iters, kkt);
  // unscale even after a failed launch sequence is pointless: return the error as is
  if (rc) return rc;
  hipLaunchKernelGGL(scale_out_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, tz, tl, dm.n, xcount, dm.ns, dm.nu, dm.m, h->vscale, z, lam);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));
  return MYR_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Restoration inside the solve call (myr_solve_opts.restoration): what the reference gets from inside its one minimize_ipopt call
// (/root/reference/myriad/nlp_solvers/__init__.py:57-58 -- IPOPT falls back to a feasibility-restoration phase when its line search
// fails) happens here for the instances the first attempt leaves without a KKT point, for EVERY binding of the C-ABI:
//   1. elastic phase  -- the system's elastic twin (x' = f(x,u) + s, cost g + rho/2 |s|^2) solved for rho = 1, 1e2, 1e4, each from the
//                        previous solution; its trajectory starts the problem itself;
//   2. second starts  -- excitation guesses (oscillating controls, states by a rollout of the true dynamics), c = 2, 3, 5 cycles.
// Host logic over device arrays: the failed rows are gathered into a working set, solved, and scattered back where they converged.
// (Rounds 2-3 had this in the Python host only: myriad_amd/trajectory_optimizers/__init__.py, removed in round 4.)
// ------------------------------------------------------------------------------------------------------------------------------
static int dispatch_rollout(myr_handle h, int B, int num_steps, int u_rows, const double* x0, const double* us,
                            const double* params, int pstride, double* xs, double* cost);
extern "C" int myr_set_var_scale(myr_handle h, const double* scale);

__global__ void gather_rows_kernel(long total, int w, const int32_t* __restrict__ idx, const double* __restrict__ src, double* __restrict__ dst) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long r = t / w; const int c = (int)(t - r * w);
    dst[t] = src[(long)idx[r] * w + c];
  }
}
__global__ void scatter_rows_kernel(long total, int w, const int32_t* __restrict__ idx, const int32_t* __restrict__ take, const double* __restrict__ src,
                                    double* __restrict__ dst) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long r = t / w; const int c = (int)(t - r * w);
    if (take[r]) dst[(long)idx[r] * w + c] = src[t];
  }
}
// z of the problem -> z of its twin: the state rows as they are, every control row widened by ns slack controls (= fill)
__global__ void twin_widen_kernel(long total, int nx, int nu, int ns, double fill, const double* __restrict__ src, double* __restrict__ dst, int n, int nt) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long b = t / nt; const int i = (int)(t - b * nt);
    double v;
    if (i < nx) v = src[b * n + i];
    else { const int row = (i - nx) / (nu + ns), c = (i - nx) % (nu + ns); v = c < nu ? src[b * n + nx + row * nu + c] : fill; }
    dst[t] = v;
  }
}
// the twin's states and controls -> a start of the problem itself; zt is first clipped into its bounds (non-finite entries -> 0 / +-1e6),
// as a start must be; slack[b] = max |s| of the clipped twin solution
__global__ void twin_clip_kernel(long total, const double* __restrict__ lb, const double* __restrict__ ub, double* __restrict__ z) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    double v = z[t];
    if (v != v) v = 0.0; else if (v > 1e300) v = 1e6; else if (v < -1e300) v = -1e6;
    v = v < lb[t] ? lb[t] : v; v = v > ub[t] ? ub[t] : v;
    z[t] = v;
  }
}
__global__ __launch_bounds__(256) void twin_slack_kernel(int nx, int rows_u, int nu, int ns, int nt, const double* __restrict__ zt, double* __restrict__ slack) {
  __shared__ double red[256];
  const long b = blockIdx.x;
  double m = 0.0;
  for (int i = threadIdx.x; i < rows_u * ns; i += blockDim.x) {
    const int row = i / ns, c = i % ns;
    const double v = fabs(zt[b * nt + nx + row * (nu + ns) + nu + c]);
    m = v > m ? v : m;
  }
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + o] ? red[threadIdx.x] : red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) slack[b] = red[0];
}
__global__ void twin_narrow_kernel(long total, int nx, int nu, int ns, const double* __restrict__ zt, double* __restrict__ z, int n, int nt) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long b = t / n; const int i = (int)(t - b * n);
    z[t] = i < nx ? zt[b * nt + i] : zt[b * nt + nx + ((i - nx) / nu) * (nu + ns) + (i - nx) % nu];
  }
}
__global__ void twin_params_kernel(int B, int np, const double* __restrict__ params, int pstride, const double* __restrict__ defaults, double rho, double* __restrict__ out) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)B * (np + 1); t += (long)gridDim.x * blockDim.x) {
    const long b = t / (np + 1); const int i = (int)(t - b * (np + 1));
    out[t] = i == np ? rho : (params ? params[b * (long)pstride + i] : defaults[i]);
  }
}
// excitation guess, controls: us[b][j][a] = centre_a + 0.95 amp_a sin(2 pi c t_j / T), t_j = T j / (rr - 1); centre / amplitude from control a's own
// bounds.  The reference writes control bounds control-major (quirk Q1: hermite_simpson.py:71-74, trapezoidal.py:58-61, shooting.py:264-267 -- flat
// entries [a rr, (a + 1) rr) of the control part hold control a's bounds) while the variables are time-major, so control a's bounds are read at flat
// index a * u_rows (u_rows = control rows of the decision vector), NOT at the variable's own entry (for nu > 1 the first row holds control 0's bounds in every component).  x0[b] = the pinned start
// state, else the guess's
__global__ void excitation_controls_kernel(int B, int rr, int u_rows, int nu, int ns, int nx, int n, double cycles, const double* __restrict__ lb, const double* __restrict__ ub,
                                           const double* __restrict__ z0, double* __restrict__ us, double* __restrict__ x0) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)B * rr * nu; t += (long)gridDim.x * blockDim.x) {
    const long b = t / ((long)rr * nu); const int j = (int)((t / nu) % rr), a = (int)(t % nu);
    const double lo = lb[b * n + nx + (long)a * u_rows], hi = ub[b * n + nx + (long)a * u_rows];
    const bool fin = lo > -1e300 && hi < 1e300;
    const double centre = fin ? 0.5 * (lo + hi) : 0.0, amp = fin ? 0.5 * (hi - lo) : 1.0;
    us[t] = centre + 0.95 * amp * sin(2.0 * 3.14159265358979323846 * cycles * ((double)j / (double)(rr - 1)));
    if (j == 0 && a == 0) {
      for (int q = 0; q < ns; ++q) { const double l = lb[b * n + q], u = ub[b * n + q]; x0[b * ns + q] = (l == u) ? l : z0[b * n + q]; }
    }
  }
}
// ... and the guess itself: states = every `xstride`-th state of the rollout, controls = every `ustride`-th row, clipped into the bounds
__global__ void excitation_pack_kernel(long total, int n, int nx, int ns, int nu, int steps, int xstride, int rr, int ustride, const double* __restrict__ xs,
                                       const double* __restrict__ us, const double* __restrict__ lb, const double* __restrict__ ub, double* __restrict__ z) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long b = t / n; const int i = (int)(t - b * n);
    double v;
    if (i < nx) { const int row = i / ns, c = i % ns; v = xs[(b * (steps + 1) + (long)row * xstride) * ns + c]; }
    else { const int row = (i - nx) / nu, c = (i - nx) % nu; v = us[(b * rr + (long)row * ustride) * nu + c]; }
    if (v != v) v = 0.0; else if (v > 1e300) v = 1e6; else if (v < -1e300) v = -1e6;
    v = v < lb[t] ? lb[t] : v; v = v > ub[t] ? ub[t] : v;
    z[t] = v;
  }
}

// how many instances ended without a KKT point -> a device word, copied to a pinned host word (the common answer, none, costs no status download)
__global__ void count_failed_kernel(int B, const int32_t* __restrict__ status, int32_t* __restrict__ out) {
  int n = 0;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) n += status[b] != MYR_STATUS_CONVERGED ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if ((threadIdx.x & 63) == 0 && n) atomicAdd(out, n);
}

static bool twin_defaults(int twin_id, double* buf) {
  switch (twin_id) {
#define X(N) case MYR_SYS_##N: Sys##N::default_params(buf); return true;
    MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
  }
  return false;
}

static unsigned grid_for(long total) { long b = (total + 255) / 256; if (b > 16384) b = 16384; if (b < 1) b = 1; return (unsigned)b; }


struct RestoreCfg { bool elastic, starts; std::vector<int> cycles; };
static RestoreCfg restore_cfg(const myr_solve_opts& so) {
  RestoreCfg c;
  int mode = so.restoration < 0 ? 3 : so.restoration;
  c.elastic = (mode & 1) != 0; c.starts = (mode & 2) != 0;
  c.cycles = {2, 3, 5};
  if (so.restoration < 0) {      // the library default listens to the environment (the explicit values do not)
    if (const char* e = getenv("MYRIAD_ELASTIC")) { if (atoi(e) == 0) c.elastic = false; }
    if (const char* e = getenv("MYRIAD_SECOND_STARTS")) {
      c.cycles.clear();
      for (const char* q = e; *q;) { char* end = nullptr; long v = strtol(q, &end, 10); if (end == q) { ++q; continue; } if (v > 0) c.cycles.push_back((int)v); q = end; }
      if (c.cycles.empty()) c.starts = false;
    }
  }
  return c;
}

// the twin handle of a collocation problem whose system has one (lazily; nullptr when there is none or its solver is not built)
static myr_handle twin_of(myr_handle h) {
  if (h->twin_unavailable) return nullptr;
  if (h->twin) return h->twin;
  SysInfo si;
  if (h->d.transcription == MYR_TR_SHOOTING || h->d.system_id >= 100 || !sys_info(h->d.system_id + 100, &si)) { h->twin_unavailable = true; return nullptr; }
  myr_problem_desc d = h->d;
  d.system_id += 100;
  myr_handle t = nullptr;
  if (myr_create(&d, &t) != MYR_OK) { h->twin_unavailable = true; return nullptr; }
  if (h->vscale_on) {          // a slack is a rate of its state: it takes the state's scale
    double sc[16];
    const int nw = h->dims.ns + h->dims.nu;
    for (int i = 0; i < nw; ++i) sc[i] = h->vscale.s[i];
    for (int i = 0; i < h->dims.ns && nw + i < 16; ++i) sc[nw + i] = h->vscale.s[i];
    if (myr_set_var_scale(t, sc) != MYR_OK) { (void)myr_destroy(t); h->twin_unavailable = true; return nullptr; }
  }
  t->solve_mode = h->solve_mode; t->solve_fused = h->solve_fused;
  h->twin = t;
  return t;
}

static int solve_restored(myr_handle h, int B, double* z, const double* lb, const double* ub, const double* params, int pstride,
                          const myr_solve_opts& so, double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt) {
  h->info_start.assign(B, 0); h->info_attempts.assign(B, 1); h->info_restored.assign(B, 0);
  h->plan_frozen = false;
  struct Thaw { myr_handle h; ~Thaw() { h->plan_frozen = false; } } thaw{h};
  const RestoreCfg cfg = restore_cfg(so);
  if (!cfg.elastic && !cfg.starts) return dispatch_solve_scaled(h, B, z, lb, ub, params, pstride, so, lam, cost, status, iters, kkt);
  const myr_dims& dm = h->dims;
  const int n = dm.n, m = dm.m, ns = dm.ns, nu = dm.nu, np = dm.np;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };
  // the caller's guess is overwritten by the first attempt: keep it
  if (int rc = ensure_buf(&h->rbuf, &h->rbuf_bytes, (al((size_t)B * n) + 2 * al((size_t)B)) * 8)) return rc;
  double* z0c = (double*)h->rbuf;
  int32_t* dstat = status ? status : (int32_t*)(z0c + al((size_t)B * n));
  int32_t* dit = iters ? iters : (int32_t*)(z0c + al((size_t)B * n) + al((size_t)B));
  HIPCHK(hipMemcpyAsync(z0c, z, (size_t)B * n * 8, hipMemcpyDeviceToDevice, h->stream));
  if (int rc = dispatch_solve_scaled(h, B, z, lb, ub, params, pstride, so, lam, cost, dstat, dit, kkt)) return rc;
  h->plan_frozen = true;
  if (!h->nfail_host) HIPCHK(hipHostMalloc((void**)&h->nfail_host, 64, hipHostMallocDefault));
  if (!h->nfail_dev) HIPCHK(hipMalloc((void**)&h->nfail_dev, 64));
  *h->nfail_host = -1;
  HIPCHK(hipMemsetAsync(h->nfail_dev, 0, 4, h->stream));
  hipLaunchKernelGGL(count_failed_kernel, dim3((unsigned)((B + 255) / 256 > 64 ? 64 : (B + 255) / 256)), dim3(256), 0, h->stream, B, dstat, h->nfail_dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h->nfail_host, h->nfail_dev, 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (*h->nfail_host < 0) return fail(MYR_E_HIP, "restoration: the count of failed instances did not arrive");
  if (*h->nfail_host == 0) return MYR_OK;
  std::vector<int32_t> hstat(B), hit(B);
  HIPCHK(hipMemcpyAsync(hstat.data(), dstat, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<int32_t> fail_;
  for (int b = 0; b < B; ++b) if (hstat[b] != MYR_STATUS_CONVERGED) fail_.push_back(b);
  if (fail_.empty()) return MYR_OK;
  myr_handle twin = cfg.elastic ? twin_of(h) : nullptr;
  if (!twin && !cfg.starts) return MYR_OK;
  HIPCHK(hipMemcpyAsync(hit.data(), dit, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));

  // working set of the failed rows (device): idx | take | z0f zf lbf ubf | pf | lamf costf kktf statf itf | twin: zt lbt ubt pt slack | excitation: us x0 xs
  const int nf0 = (int)fail_.size();
  const int nt = twin ? twin->dims.n : 0;
  const int mc = (h->d.transcription == MYR_TR_SHOOTING && h->d.integration_method == MYR_INT_RK4) ? 2 : 1;
  const int steps = (dm.u_rows - 1) / mc;
  const int rr = (h->d.integration_method == MYR_INT_RK4) ? 2 * steps + 1 : steps + 1;
  size_t words = 2 * al((size_t)nf0) + 4 * al((size_t)nf0 * n) + al((size_t)nf0 * (np > 0 ? np : 1)) + al((size_t)nf0 * m) + al((size_t)nf0) + al((size_t)nf0 * 3) + 2 * al((size_t)nf0);
  if (twin) words += 3 * al((size_t)nf0 * nt) + al((size_t)nf0 * (np + 1)) + al((size_t)nf0) + al(64);
  if (cfg.starts) words += al((size_t)nf0 * rr * nu) + al((size_t)nf0 * ns) + al((size_t)nf0 * (steps + 1) * ns);
  if (int rc = ensure_buf(&h->fbuf, &h->fbuf_bytes, words * 8)) return rc;
  double* q = (double*)h->fbuf;
  int32_t* didx = (int32_t*)q; q += al((size_t)nf0);
  int32_t* dtake = (int32_t*)q; q += al((size_t)nf0);
  double* z0f = q; q += al((size_t)nf0 * n);
  double* zf = q; q += al((size_t)nf0 * n);
  double* lbf = q; q += al((size_t)nf0 * n);
  double* ubf = q; q += al((size_t)nf0 * n);
  double* pf = q; q += al((size_t)nf0 * (np > 0 ? np : 1));
  double* lamf = q; q += al((size_t)nf0 * m);
  double* costf = q; q += al((size_t)nf0);
  double* kktf = q; q += al((size_t)nf0 * 3);
  int32_t* statf = (int32_t*)q; q += al((size_t)nf0);
  int32_t* itf = (int32_t*)q; q += al((size_t)nf0);
  double *zt = nullptr, *lbt = nullptr, *ubt = nullptr, *pt = nullptr, *dslack = nullptr, *ddef = nullptr;
  if (twin) { zt = q; q += al((size_t)nf0 * nt); lbt = q; q += al((size_t)nf0 * nt); ubt = q; q += al((size_t)nf0 * nt); pt = q; q += al((size_t)nf0 * (np + 1));
              dslack = q; q += al((size_t)nf0); ddef = q; q += al(64); }
  double *dus = nullptr, *dx0 = nullptr, *dxs = nullptr;
  if (cfg.starts) { dus = q; q += al((size_t)nf0 * rr * nu); dx0 = q; q += al((size_t)nf0 * ns); dxs = q; q += al((size_t)nf0 * (steps + 1) * ns); }
  const bool per_row_params = params && pstride != 0;
  const int nx = dm.x_rows * ns;

  // rows `rows` (indices into the batch) -> the working set
  auto gather = [&](const std::vector<int32_t>& rows) -> int {
    const int nf = (int)rows.size();
    HIPCHK(hipMemcpyAsync(didx, rows.data(), (size_t)nf * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)nf * n)), dim3(256), 0, h->stream, (long)nf * n, n, didx, z0c, z0f);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)nf * n)), dim3(256), 0, h->stream, (long)nf * n, n, didx, lb, lbf);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)nf * n)), dim3(256), 0, h->stream, (long)nf * n, n, didx, ub, ubf);
    if (per_row_params) hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)nf * np)), dim3(256), 0, h->stream, (long)nf * np, np, didx, params, pf);
    HIPCHK(hipGetLastError());
    return MYR_OK;
  };
  const double* pfp = per_row_params ? pf : params;
  // rows of the working set with take[r] != 0 -> the caller's arrays
  auto scatter = [&](const std::vector<int32_t>& take, int nf) -> int {
    HIPCHK(hipMemcpyAsync(dtake, take.data(), (size_t)nf * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long)nf * n)), dim3(256), 0, h->stream, (long)nf * n, n, didx, dtake, zf, z);
    if (lam) hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long)nf * m)), dim3(256), 0, h->stream, (long)nf * m, m, didx, dtake, lamf, lam);
    if (cost) hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long)nf)), dim3(256), 0, h->stream, (long)nf, 1, didx, dtake, costf, cost);
    if (kkt) hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((long)nf * 3)), dim3(256), 0, h->stream, (long)nf * 3, 3, didx, dtake, kktf, kkt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));     // (`take` is a host vector of the caller's frame)
    return MYR_OK;
  };
  std::vector<int32_t> st2, it2;
  auto read_back = [&](int nf) -> int {
    st2.resize(nf); it2.resize(nf);
    HIPCHK(hipMemcpyAsync(st2.data(), statf, (size_t)nf * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(it2.data(), itf, (size_t)nf * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return MYR_OK;
  };
  myr_solve_opts plain = so;
  plain.restoration = 0;

  if (twin) {      // ---- elastic phase -----------------------------------------------------------------------------------------
    const int nf = (int)fail_.size();
    if (int rc = gather(fail_)) return rc;
    double defs[64];
    for (int i = 0; i < 64; ++i) defs[i] = 0.0;
    if (!twin_defaults(twin->d.system_id - 100, defs)) return fail(MYR_E_ARG, "restoration: no default parameters for the system");
    HIPCHK(hipMemcpyAsync(ddef, defs, 64 * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(twin_widen_kernel, dim3(grid_for((long)nf * nt)), dim3(256), 0, h->stream, (long)nf * nt, nx, nu, ns, 0.0, z0f, zt, n, nt);
    hipLaunchKernelGGL(twin_widen_kernel, dim3(grid_for((long)nf * nt)), dim3(256), 0, h->stream, (long)nf * nt, nx, nu, ns, -INFINITY, lbf, lbt, n, nt);
    hipLaunchKernelGGL(twin_widen_kernel, dim3(grid_for((long)nf * nt)), dim3(256), 0, h->stream, (long)nf * nt, nx, nu, ns, INFINITY, ubf, ubt, n, nt);
    HIPCHK(hipGetLastError());
    myr_solve_opts topt = plain;
    if (topt.max_iter > 500) topt.max_iter = 500;      // per twin solve (the ones that help take 30-300 iterations)
    const double rhos[3] = {1.0, 1e2, 1e4};
    std::vector<double> slack_prev(nf, 0.0), slack_last(nf, 0.0);
    std::vector<int64_t> it_acc(nf, 0);
    std::vector<int32_t> twin_stat(nf, 1);
    bool twin_ok = true;
    for (int k = 0; k < 3 && twin_ok; ++k) {
      hipLaunchKernelGGL(twin_params_kernel, dim3(grid_for((long)nf * (np + 1))), dim3(256), 0, h->stream, nf, np, pfp, per_row_params ? np : 0, ddef, rhos[k], pt);
      HIPCHK(hipGetLastError());
      HIPCHK(hipStreamSynchronize(h->stream));      // the twin runs on a stream of its own
      const int rc = dispatch_solve_scaled(twin, nf, zt, lbt, ubt, pt, np + 1, topt, nullptr, nullptr, statf, itf, nullptr);
      if (rc == MYR_E_UNSUPPORTED) {      // (a twin whose solver is not built for this scheme): drop the handle -- its stream and buffers -- for good, and do not
        twin_ok = false;                  // leave "... not built" behind as the last error of a myr_solve that returns MYR_OK
        h->twin_unavailable = true;
        (void)myr_destroy(h->twin); h->twin = nullptr; twin = nullptr;
        g_err.clear();
        break;
      }
      if (rc) return rc;
      hipLaunchKernelGGL(twin_clip_kernel, dim3(grid_for((long)nf * nt)), dim3(256), 0, h->stream, (long)nf * nt, lbt, ubt, zt);
      hipLaunchKernelGGL(twin_slack_kernel, dim3((unsigned)nf), dim3(256), 0, h->stream, nx, dm.u_rows, nu, ns, nt, zt, dslack);
      HIPCHK(hipGetLastError());
      if (int rc2 = read_back(nf)) return rc2;
      slack_prev = slack_last;
      HIPCHK(hipMemcpy(slack_last.data(), dslack, (size_t)nf * 8, hipMemcpyDeviceToHost));
      for (int r = 0; r < nf; ++r) { it_acc[r] += it2[r]; twin_stat[r] = st2[r]; }
      if (getenv("MYRIAD_DEBUG_ELASTIC"))
        for (int r = 0; r < nf && r < 8; ++r)
          fprintf(stderr, "[myriad] elastic phase: instance %d, rho %g: twin status %d after %d iterations, slack %.3e\n", (int)fail_[r], rhos[k], (int)st2[r], (int)it2[r], slack_last[r]);
    }
    if (twin_ok) {
      hipLaunchKernelGGL(twin_narrow_kernel, dim3(grid_for((long)nf * n)), dim3(256), 0, h->stream, (long)nf * n, nx, nu, ns, zt, zf, n, nt);
      HIPCHK(hipGetLastError());
      if (int rc = dispatch_solve_scaled(h, nf, zf, lbf, ubf, pfp, per_row_params ? np : pstride, plain, lamf, costf, statf, itf, kktf)) return rc;
      if (int rc = read_back(nf)) return rc;
      std::vector<int32_t> take(nf, 0), left;
      for (int r = 0; r < nf; ++r) {
        const int b = fail_[r];
        const bool ok = st2[r] == MYR_STATUS_CONVERGED;
        take[r] = ok ? 1 : 0;
        hit[b] += (int32_t)(it2[r] + it_acc[r]);
        h->info_attempts[b] += 4;
        if (ok) { hstat[b] = MYR_STATUS_CONVERGED; h->info_restored[b] = 1; }
        else {
          // stationary point of the infeasibility: the twin converged for the largest rho and its slack neither vanished nor shrank.
          // (A twin that ran on the lane kernel gives no verdict: its instantiations with many variables per point are not covered by the
          // wave / lane agreement tests -- and ROCKETLANDING's own lane kernel was once miscompiled)
          if (twin->last_solve_form == 1 && twin_stat[r] == MYR_STATUS_CONVERGED && slack_last[r] > 1e-3 && slack_last[r] > 0.1 * slack_prev[r])
            hstat[b] = MYR_STATUS_INFEASIBLE;
          left.push_back(b);
        }
      }
      if (int rc = scatter(take, nf)) return rc;
      fail_.swap(left);
    }
  }
  if (cfg.starts) {      // ---- second starts: excitation guesses --------------------------------------------------------------
    const int xstride = steps / (dm.x_rows - 1 > 0 ? dm.x_rows - 1 : 1), ustride = (rr - 1) / (dm.u_rows - 1 > 0 ? dm.u_rows - 1 : 1);
    for (size_t ci = 0; ci < cfg.cycles.size() && !fail_.empty(); ++ci) {
      const int nf = (int)fail_.size(), c = cfg.cycles[ci];
      if (int rc = gather(fail_)) return rc;
      hipLaunchKernelGGL(excitation_controls_kernel, dim3(grid_for((long)nf * rr * nu)), dim3(256), 0, h->stream, nf, rr, dm.u_rows, nu, ns, nx, n, (double)c, lbf, ubf, z0f, dus, dx0);
      HIPCHK(hipGetLastError());
      if (int rc = dispatch_rollout(h, nf, steps, rr, dx0, dus, pfp, per_row_params ? np : pstride, dxs, nullptr)) return rc;
      hipLaunchKernelGGL(excitation_pack_kernel, dim3(grid_for((long)nf * n)), dim3(256), 0, h->stream, (long)nf * n, n, nx, ns, nu, steps, xstride, rr, ustride, dxs, dus, lbf, ubf, zf);
      HIPCHK(hipGetLastError());
      if (int rc = dispatch_solve_scaled(h, nf, zf, lbf, ubf, pfp, per_row_params ? np : pstride, plain, lamf, costf, statf, itf, kktf)) return rc;
      if (int rc = read_back(nf)) return rc;
      std::vector<int32_t> take(nf, 0), left;
      for (int r = 0; r < nf; ++r) {
        const int b = fail_[r];
        const bool ok = st2[r] == MYR_STATUS_CONVERGED;
        take[r] = ok ? 1 : 0;
        hit[b] += it2[r];
        h->info_attempts[b] += 1;
        if (ok) { hstat[b] = MYR_STATUS_CONVERGED; h->info_restored[b] = 0; h->info_start[b] = c; }
        else left.push_back(b);
      }
      if (int rc = scatter(take, nf)) return rc;
      fail_.swap(left);
    }
  }
  HIPCHK(hipMemcpyAsync(dstat, hstat.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dit, hit.data(), (size_t)B * 4, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MYR_OK;
}

extern "C" int myr_solve_info(myr_handle h, int32_t B, int32_t* start, int32_t* attempts, int32_t* restored) {
  if (!h) return fail(MYR_E_ARG, "myr_solve_info: null handle");
  if (B != (int32_t)h->info_start.size()) return fail(MYR_E_ARG, "myr_solve_info: B is not the batch size of the last solve on this handle");
  if (start) memcpy(start, h->info_start.data(), (size_t)B * 4);
  if (attempts) memcpy(attempts, h->info_attempts.data(), (size_t)B * 4);
  if (restored) memcpy(restored, h->info_restored.data(), (size_t)B * 4);
  return MYR_OK;
}

extern "C" int myr_solve_plan(myr_handle h, int32_t* plan) {
  if (!h || !plan) return fail(MYR_E_ARG, "myr_solve_plan: null argument");
  memcpy(plan, h->plan, sizeof(h->plan));
  return MYR_OK;
}

extern "C" int myr_set_var_scale(myr_handle h, const double* scale) {
  if (!h) return fail(MYR_E_ARG, "myr_set_var_scale: null handle");
  if (h->d.system_id == MYR_SYS_NODE_CARTPOLE && scale) return fail(MYR_E_UNSUPPORTED, "myr_set_var_scale: not available for NODE systems");
  const int nw = h->dims.ns + h->dims.nu;
  if (nw > 16) return fail(MYR_E_CAPACITY, "myr_set_var_scale: more than 16 variables per point");
  if (h->twin) { (void)myr_destroy(h->twin); h->twin = nullptr; }      // (the twin of the restoration phase takes its scales from here: made again on demand)
  bool on = false;
  for (int i = 0; i < 16; ++i) h->vscale.s[i] = 1.0;
  if (scale) {
    for (int i = 0; i < nw; ++i) {
      if (!(scale[i] > 0.0) || !(scale[i] < 1e300)) return fail(MYR_E_ARG, "myr_set_var_scale: scales must be positive and finite");
      h->vscale.s[i] = scale[i];
      on = on || scale[i] != 1.0;
    }
  }
  h->vscale_on = on;
  return MYR_OK;
}

extern "C" int myr_solve(myr_handle h, int32_t B, double* z, const double* lb, const double* ub,
                         const double* params, int32_t params_stride, const myr_solve_opts* opts,
                         double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt, int32_t mem) {
  if (!h || !z || !lb || !ub) return fail(MYR_E_ARG, "myr_solve: null handle, z, lb or ub");
  if (h->d.system_id == MYR_SYS_INVASIVEPLANT) return no_such_path(h, "myr_solve");
  if (B < 0) return fail(MYR_E_ARG, "myr_solve: negative batch");
  if (B == 0) { h->info_start.clear(); h->info_attempts.clear(); h->info_restored.clear(); return MYR_OK; }
  if (params && params_stride != 0 && params_stride != h->dims.np)
    return fail(MYR_E_ARG, "myr_solve: params_stride must be 0 (shared) or np");
  if (!params && h->d.system_id == MYR_SYS_NODE_CARTPOLE) return fail(MYR_E_ARG, "myr_solve: a NODE system needs its weights in `params`");
  myr_solve_opts so;
  if (opts) so = *opts; else myr_default_solve_opts(&so);
  if (so.max_iter < 0 || !(so.tol_feas > 0) || !(so.tol_stat > 0) || !(so.tol_compl > 0) || !(so.mu_init > 0))
    return fail(MYR_E_ARG, "myr_solve: bad options");
  HIPCHK(hipSetDevice(h->d.device));
  const myr_dims& dm = h->dims;
  if (mem == MYR_MEM_DEVICE)
    return solve_restored(h, B, z, lb, ub, params, params_stride, so, lam, cost, status, iters, kkt);
  if (mem != MYR_MEM_HOST) return fail(MYR_E_ARG, "myr_solve: bad mem kind");
  const size_t nz = (size_t)B * dm.n, nl = lam ? (size_t)B * dm.m : 0;
  const size_t npar = params ? (params_stride ? (size_t)B * dm.np : (size_t)dm.np) : 0;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };
  const size_t total = 3 * al(nz) + al(nl) + al(npar) + al(B) /*cost*/ + al(B) /*status+iters as int32 pairs*/ + al(3 * (size_t)B);
  int rc = ensure_dbuf(h, total * 8);
  if (rc) return rc;
  double* dz = (double*)h->dbuf;
  double* dlb = dz + al(nz);
  double* dub = dlb + al(nz);
  double* dlam = dub + al(nz);
  double* dp = dlam + al(nl);
  double* dcost = dp + al(npar);
  int32_t* dstat = (int32_t*)(dcost + al(B));
  int32_t* dit = dstat + B;
  double* dkkt = (double*)(dstat) + al(B);
  HIPCHK(hipMemcpyAsync(dz, z, nz * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dlb, lb, nz * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dub, ub, nz * 8, hipMemcpyHostToDevice, h->stream));
  if (npar) HIPCHK(hipMemcpyAsync(dp, params, npar * 8, hipMemcpyHostToDevice, h->stream));
  rc = solve_restored(h, B, dz, dlb, dub, npar ? dp : nullptr, params_stride, so, nl ? dlam : nullptr, dcost, dstat, dit, dkkt);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(z, dz, nz * 8, hipMemcpyDeviceToHost, h->stream));
  if (nl) HIPCHK(hipMemcpyAsync(lam, dlam, nl * 8, hipMemcpyDeviceToHost, h->stream));
  if (cost) HIPCHK(hipMemcpyAsync(cost, dcost, (size_t)B * 8, hipMemcpyDeviceToHost, h->stream));
  if (status) HIPCHK(hipMemcpyAsync(status, dstat, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
  if (iters) HIPCHK(hipMemcpyAsync(iters, dit, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
  if (kkt) HIPCHK(hipMemcpyAsync(kkt, dkkt, (size_t)B * 24, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MYR_OK;
}

// ---- myr_solve_x0: B instances that differ in their START STATE only ------------------------------------------------
// The reference builds guess and bounds of an instance from system.x_0 in the optimiser's constructor
// (hermite_simpson.py:37-48 guess, :55-81 bounds; trapezoidal.py:36-50, 55-77; shooting.py:56-74, 247-275).  For a batch
// of start states that is B x 3 n doubles of host packing and PCIe traffic carrying B x ns doubles of information: the
// expansion runs on the device instead.  z0[b][i] = g0[i] + g1[i] * x0s[b][i mod ns] on the state rows (two roundings, the
// product and the sum: bit for bit what numpy's x0 * (1 - lin) + x_T * lin gives with g1 = 1 - lin, g0 = x_T * lin), g0[i]
// elsewhere; lb / ub = the templates with the first point's state rows replaced by x0s[b].
__global__ void pack_x0_kernel(long total, int n, int ns, int xcount, const double* __restrict__ x0s, const double* __restrict__ g0,
                               const double* __restrict__ g1, const double* __restrict__ lbt, const double* __restrict__ ubt,
                               double* __restrict__ z, double* __restrict__ lb, double* __restrict__ ub) {
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long b = t / n;
    const int i = (int)(t - b * n);
    double zi = g0[i], l = lbt[i], u = ubt[i];
    if (i < xcount) {
      const double x = x0s[b * ns + i % ns];
      double prod = x * g1[i];
      asm volatile("" : "+v"(prod));     // a rounded product of its own: the compiler must not contract it into an fma with the sum
      zi = prod + zi;
      if (i < ns) { l = x; u = x; }
    }
    z[t] = zi; lb[t] = l; ub[t] = u;
  }
}

extern "C" int myr_solve_x0(myr_handle h, int32_t B, const double* x0s, const double* g0, const double* g1, const double* lb,
                            const double* ub, const double* params, int32_t params_stride, const myr_solve_opts* opts,
                            double* z, double* lam, double* cost, int32_t* status, int32_t* iters, double* kkt, int32_t mem) {
  if (!h || !x0s || !g0 || !g1 || !lb || !ub || !z) return fail(MYR_E_ARG, "myr_solve_x0: null handle, x0s, g0, g1, lb, ub or z");
  if (h->d.system_id == MYR_SYS_INVASIVEPLANT) return no_such_path(h, "myr_solve_x0");
  if (B < 0) return fail(MYR_E_ARG, "myr_solve_x0: negative batch");
  if (B == 0) { h->info_start.clear(); h->info_attempts.clear(); h->info_restored.clear(); return MYR_OK; }
  if (params && params_stride != 0 && params_stride != h->dims.np)
    return fail(MYR_E_ARG, "myr_solve_x0: params_stride must be 0 (shared) or np");
  if (!params && h->d.system_id == MYR_SYS_NODE_CARTPOLE) return fail(MYR_E_ARG, "myr_solve_x0: a NODE system needs its weights in `params`");
  if (mem != MYR_MEM_HOST && mem != MYR_MEM_DEVICE) return fail(MYR_E_ARG, "myr_solve_x0: bad mem kind");
  myr_solve_opts so;
  if (opts) so = *opts; else myr_default_solve_opts(&so);
  if (so.max_iter < 0 || !(so.tol_feas > 0) || !(so.tol_stat > 0) || !(so.tol_compl > 0) || !(so.mu_init > 0))
    return fail(MYR_E_ARG, "myr_solve_x0: bad options");
  HIPCHK(hipSetDevice(h->d.device));
  const myr_dims& dm = h->dims;
  const bool host = mem == MYR_MEM_HOST;
  const size_t nz = (size_t)B * dm.n, nl = lam ? (size_t)B * dm.m : 0, nx0 = (size_t)B * dm.ns;
  const size_t npar = params ? (params_stride ? (size_t)B * dm.np : (size_t)dm.np) : 0;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };
  // device scratch: the expanded bounds always; for host callers also the iterate, the inputs and the results
  size_t total = 2 * al(nz);
  if (host) total += al(nz) + al(nx0) + 4 * al((size_t)dm.n) + al(nl) + al(npar) + al(B) + al(B) + al(3 * (size_t)B);
  int rc = ensure_dbuf(h, total * 8);
  if (rc) return rc;
  double* dlb = (double*)h->dbuf;
  double* dub = dlb + al(nz);
  double* dz = z; const double* dx0 = x0s; const double* dg0 = g0; const double* dg1 = g1; const double* dlt = lb; const double* dut = ub;
  const double* dpar = params;
  double* dlam = lam; double* dcost = cost; int32_t* dstat = status; int32_t* dit = iters; double* dkkt = kkt;
  if (host) {
    double* q = dub + al(nz);
    dz = q; q += al(nz);
    double* hx0 = q; q += al(nx0);
    double* tpl = q; q += 4 * al((size_t)dm.n);
    dlam = nl ? q : nullptr; q += al(nl);
    double* hp = q; q += al(npar);
    dcost = q; q += al(B);
    dstat = (int32_t*)q; dit = dstat + B; q += al(B);
    dkkt = q;
    HIPCHK(hipMemcpyAsync(hx0, x0s, nx0 * 8, hipMemcpyHostToDevice, h->stream));
    const double* src[4] = {g0, g1, lb, ub};
    for (int k = 0; k < 4; ++k) HIPCHK(hipMemcpyAsync(tpl + k * al((size_t)dm.n), src[k], (size_t)dm.n * 8, hipMemcpyHostToDevice, h->stream));
    if (npar) HIPCHK(hipMemcpyAsync(hp, params, npar * 8, hipMemcpyHostToDevice, h->stream));
    dx0 = hx0; dg0 = tpl; dg1 = tpl + al((size_t)dm.n); dlt = tpl + 2 * al((size_t)dm.n); dut = tpl + 3 * al((size_t)dm.n);
    dpar = npar ? hp : nullptr;
  }
  {
    const long tz = (long)nz;
    long blocks = (tz + 255) / 256; if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pack_x0_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, tz, dm.n, dm.ns, dm.x_rows * dm.ns, dx0, dg0, dg1, dlt, dut, dz, dlb, dub);
    HIPCHK(hipGetLastError());
  }
  rc = solve_restored(h, B, dz, dlb, dub, dpar, params_stride, so, dlam, dcost, dstat, dit, dkkt);
  if (rc) return rc;
  if (host) {
    HIPCHK(hipMemcpyAsync(z, dz, nz * 8, hipMemcpyDeviceToHost, h->stream));
    if (nl) HIPCHK(hipMemcpyAsync(lam, dlam, nl * 8, hipMemcpyDeviceToHost, h->stream));
    if (cost) HIPCHK(hipMemcpyAsync(cost, dcost, (size_t)B * 8, hipMemcpyDeviceToHost, h->stream));
    if (status) HIPCHK(hipMemcpyAsync(status, dstat, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (iters) HIPCHK(hipMemcpyAsync(iters, dit, (size_t)B * 4, hipMemcpyDeviceToHost, h->stream));
    if (kkt) HIPCHK(hipMemcpyAsync(kkt, dkkt, (size_t)B * 24, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return MYR_OK;
}

static int dispatch_rollout(myr_handle h, int B, int num_steps, int u_rows, const double* x0, const double* us,
                            const double* params, int pstride, double* xs, double* cost) {
  KTimer& kt = h->kt[MYR_K_ROLLOUT];
  if (int rc_fill = stack_fill(h)) return rc_fill;
  HIPCHK(hipEventRecord(kt.a, h->stream));
  int rc = MYR_E_ARG;
  switch (h->d.system_id) {
#define X(N) case MYR_SYS_##N: rc = rollout_for_system<Sys##N>(h, B, num_steps, u_rows, x0, us, params, pstride, xs, cost); break;
    MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
    case MYR_SYS_NODE_CARTPOLE: rc = rollout_for_system<SysNODE_CARTPOLE>(h, B, num_steps, u_rows, x0, us, params, pstride, xs, cost); break;
    default: return no_such_path(h, "rollout");
  }
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(kt.b, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
  kt.sum_ms += ms; kt.launches += 1;
  return MYR_OK;
}

extern "C" int myr_rollout(myr_handle h, int32_t B, int32_t num_steps, int32_t u_rows, const double* x0, const double* us,
                           const double* params, int32_t params_stride, double* xs, double* cost, int32_t mem) {
  if (!h || !x0 || !us) return fail(MYR_E_ARG, "myr_rollout: null handle, x0 or us");
  if (h->d.system_id == MYR_SYS_INVASIVEPLANT) return no_such_path(h, "myr_rollout");
  if (B < 0 || num_steps < 1 || u_rows < 1) return fail(MYR_E_ARG, "myr_rollout: bad sizes");
  if (B == 0) return MYR_OK;
  if (params && params_stride != 0 && params_stride != h->dims.np)
    return fail(MYR_E_ARG, "myr_rollout: params_stride must be 0 (shared) or np");
  if (!params && h->d.system_id == MYR_SYS_NODE_CARTPOLE) return fail(MYR_E_ARG, "myr_rollout: a NODE system needs its weights in `params`");
  HIPCHK(hipSetDevice(h->d.device));
  const myr_dims& dm = h->dims;
  if (mem == MYR_MEM_DEVICE) return dispatch_rollout(h, B, num_steps, u_rows, x0, us, params, params_stride, xs, cost);
  if (mem != MYR_MEM_HOST) return fail(MYR_E_ARG, "myr_rollout: bad mem kind");
  const size_t nx0 = (size_t)B * dm.ns, nus = (size_t)B * u_rows * dm.nu;
  const size_t npar = params ? (params_stride ? (size_t)B * dm.np : (size_t)dm.np) : 0;
  const size_t nxs = xs ? (size_t)B * (num_steps + 1) * dm.ns : 0, nc = cost ? (size_t)B : 0;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };
  int rc = ensure_dbuf(h, (al(nx0) + al(nus) + al(npar) + al(nxs) + al(nc)) * 8);
  if (rc) return rc;
  double* dx0 = (double*)h->dbuf;
  double* dus = dx0 + al(nx0);
  double* dp = dus + al(nus);
  double* dxs = dp + al(npar);
  double* dc = dxs + al(nxs);
  HIPCHK(hipMemcpyAsync(dx0, x0, nx0 * 8, hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(dus, us, nus * 8, hipMemcpyHostToDevice, h->stream));
  if (npar) HIPCHK(hipMemcpyAsync(dp, params, npar * 8, hipMemcpyHostToDevice, h->stream));
  rc = dispatch_rollout(h, B, num_steps, u_rows, dx0, dus, npar ? dp : nullptr, params_stride, nxs ? dxs : nullptr, nc ? dc : nullptr);
  if (rc) return rc;
  if (nxs) HIPCHK(hipMemcpyAsync(xs, dxs, nxs * 8, hipMemcpyDeviceToHost, h->stream));
  if (nc) HIPCHK(hipMemcpyAsync(cost, dc, nc * 8, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  return MYR_OK;
}

extern "C" int myr_fbsm(myr_handle h, int32_t B, int32_t N, const double* x0, const double* adj_T, const double* params,
                        int32_t params_stride, const double* clip_lo, const double* clip_hi, double bang, double delta,
                        int32_t max_sweeps, double* xs, double* us, double* adjs, int32_t* sweeps, int32_t mem) {
  if (!h || !x0 || !xs || !us || !adjs || !clip_lo || !clip_hi) return fail(MYR_E_ARG, "myr_fbsm: null handle or array");
  if (h->dims.nu > 8) return fail(MYR_E_CAPACITY, "myr_fbsm: more than 8 controls");
  VarScale vlo{}, vhi{};
  for (int c = 0; c < h->dims.nu; ++c) { vlo.s[c] = clip_lo[c]; vhi.s[c] = clip_hi[c]; }
  if (B < 0 || N < 1 || max_sweeps < 1) return fail(MYR_E_ARG, "myr_fbsm: bad sizes");
  if (mem != MYR_MEM_HOST) return fail(MYR_E_ARG, "myr_fbsm: host arrays only");
  if (params && params_stride != 0 && params_stride != h->dims.np)
    return fail(MYR_E_ARG, "myr_fbsm: params_stride must be 0 (shared) or np");
  if (B == 0) return MYR_OK;
  HIPCHK(hipSetDevice(h->d.device));
  const myr_dims& dm = h->dims;
  long Bp = ((long)B + 63) / 64 * 64;
  if (((Bp / 64) & 1) == 0) Bp += 64;                 // odd multiple of 64 lanes: rotate points over HBM channels
  const bool discrete = h->d.system_id == MYR_SYS_INVASIVEPLANT;      // u has one row per step, not per point
  if (discrete && !params) return fail(MYR_E_ARG, "myr_fbsm: a discrete system needs `params`");
  const size_t rows_x = (size_t)(N + 1) * dm.ns, rows_u = (size_t)(N + (discrete ? 0 : 1)) * dm.nu;
  const size_t nx0 = (size_t)B * dm.ns, npar = params ? (params_stride ? (size_t)B * dm.np : (size_t)dm.np) : 0;
  auto al = [](size_t v) { return (v + 1) & ~(size_t)1; };
  // batch-minor working arrays + instance-major staging for the transposes + sweeps
  const size_t work = (2 * rows_x + rows_u) * (size_t)Bp, stage = (size_t)B * (rows_x > rows_u ? rows_x : rows_u);
  int rc = ensure_dbuf(h, (al(nx0) + al(npar) + al((size_t)dm.ns) + al(work) + al(stage) + al((size_t)B)) * 8);
  if (rc) return rc;
  double* dx0 = (double*)h->dbuf;
  double* dp = dx0 + al(nx0);
  double* dadj = dp + al(npar);
  double* X = dadj + al((size_t)dm.ns);
  double* A = X + rows_x * Bp;
  double* U = A + rows_x * Bp;
  double* st = X + al(work);
  int32_t* dsw = reinterpret_cast<int32_t*>(st + al(stage));
  HIPCHK(hipMemcpyAsync(dx0, x0, nx0 * 8, hipMemcpyHostToDevice, h->stream));
  if (npar) HIPCHK(hipMemcpyAsync(dp, params, npar * 8, hipMemcpyHostToDevice, h->stream));
  if (adj_T) HIPCHK(hipMemcpyAsync(dadj, adj_T, (size_t)dm.ns * 8, hipMemcpyHostToDevice, h->stream));
#define MYR_FBSM(S) rc = launch_fbsm<S>(h, B, Bp, N, dx0, adj_T ? dadj : nullptr, npar ? dp : nullptr, params_stride, vlo, vhi, bang, delta, max_sweeps, X, U, A, dsw)
  switch (h->d.system_id) {
#define X(N) case MYR_SYS_##N: MYR_FBSM(Sys##N); break;
    MYR_CLOSED_FORM_SYSTEMS(X)
#undef X
    case MYR_SYS_INVASIVEPLANT: {
      KTimer& kt = h->kt[MYR_K_FBSM];
      if (int rc_fill = stack_fill(h)) return rc_fill;
      HIPCHK(hipEventRecord(kt.a, h->stream));
      hipLaunchKernelGGL(fbsm_discrete_kernel<DiscINVASIVEPLANT>, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, B, Bp, N, dx0,
                         adj_T ? dadj : nullptr, dp, params_stride, vlo, vhi, delta, max_sweeps, X, U, A, dsw);
      HIPCHK(hipGetLastError());
      HIPCHK(hipEventRecord(kt.b, h->stream));
      HIPCHK(hipStreamSynchronize(h->stream));
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, kt.a, kt.b));
      kt.sum_ms += ms;
      kt.launches += 1;
      rc = MYR_OK;
      break;
    }
    default: rc = fail(MYR_E_UNSUPPORTED, "myr_fbsm: this system has no adjoint dynamics");
  }
#undef MYR_FBSM
  if (rc) return rc;
  struct { double* src; double* dst; size_t cols; } outs[3] = {{X, xs, rows_x}, {U, us, rows_u}, {A, adjs, rows_x}};
  for (auto& o : outs) {
    dim3 grid((unsigned)((o.cols + 31) / 32), (unsigned)((B + 31) / 32));
    hipLaunchKernelGGL(transpose_back_kernel, grid, dim3(256), 0, h->stream, o.src, st, B, (int)o.cols, Bp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(o.dst, st, (size_t)B * o.cols * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  if (sweeps) HIPCHK(hipMemcpy(sweeps, dsw, (size_t)B * 4, hipMemcpyDeviceToHost));
  return MYR_OK;
}
#endif  // !MYR_TU_SYSTEM
