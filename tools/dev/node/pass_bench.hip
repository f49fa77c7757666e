// dev tool: the matrix-core passes of node_mfma.h alone -- cycles per pass (wavefront 0 of workgroup 0, all workgroups running the same
// pass on their own records) and the round-5 modes (3, 4, activations stored by MODE 0) against the round-2 modes (1, 2) on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/dev/node/pass_bench.hip -o build/pass_bench && build/pass_bench [blocks] [waves]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "../../../myriad_amd/csrc/node_mfma.h"
using namespace myriad;
using NM = NodeMfma64;
constexpr int N = 100, K = 201, NSV = 4, PTN = 4 + 16 + 4 + 15, NT = (K + 15) / 16;

struct Bufs { const double *params, *z, *dz, *lam; double *pt, *hb, *tb, *sF; long long* cyc; };

template <int MODE>
__global__ __launch_bounds__(256, 1) void bench(Bufs b, int reps, int h_valid, double alpha, int use_store) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* z = lds; double* dz = z + K * 5; double* lam = dz + K * 5; double* sF = lam + 2 * N * NSV; double* wl = sF + K * NSV;
  const int tid = threadIdx.x, nt = blockDim.x, W = nt / 64;
  for (int i = tid; i < K * 5; i += nt) { z[i] = b.z[i]; dz[i] = b.dz[i]; }
  for (int i = tid; i < 2 * N * NSV; i += nt) lam[i] = b.lam[i];
  NM::load_weights(b.params, wl, tid, nt);
  __syncthreads();
  NM::ArgsT<nd_lds> a;
  a.z = (const nd_lds*)z; a.dz = (const nd_lds*)dz; a.lam = (const nd_lds*)lam;
  a.pt = (nd_glb*)(b.pt + (long)blockIdx.x * PTN * K); a.sF = (nd_lds*)sF;
  a.alpha = alpha; a.h6 = 0.02 / 6; a.h8 = 0.02 / 8; a.K = K; a.N = N;
  a.pf_f = 0; a.pf_a = 4; a.pf_b = 20; a.pf_d2 = 24;
  a.t0 = __builtin_amdgcn_readfirstlane(tid >> 6); a.ts = W;
  if (use_store) { a.hb = (nd_glb*)(b.hb + (long)blockIdx.x * NT * NM::HB_TILE); a.mb = (nd_glb*)(b.tb + (long)blockIdx.x * NT * NM::MB_TILE); }
  a.h_valid = h_valid;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) { NM::pass<MODE, nd_lds>((const nd_lds*)wl, a, tid & 63); __syncthreads(); }
  const long long t1 = clock64();
  if (blockIdx.x == 0 && tid == 0) b.cyc[0] = (t1 - t0) / reps;
  if (MODE == 0 && blockIdx.x == 0) for (int i = tid; i < K * NSV; i += nt) b.sF[i] = sF[i];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int MODE> long long run(Bufs b, int blocks, int waves, int reps, int h_valid, double alpha, int use_store) {
  const size_t lds = (size_t)(2 * K * 5 + 2 * N * NSV + K * NSV + NM::L_N) * 8;
  CK(hipFuncSetAttribute((const void*)bench<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  bench<MODE><<<blocks, 64 * waves, lds>>>(b, reps, h_valid, alpha, use_store);
  CK(hipDeviceSynchronize());
  long long c; CK(hipMemcpy(&c, b.cyc, 8, hipMemcpyDeviceToHost)); return c;
}
static double maxdiff(const std::vector<double>& a, const std::vector<double>& b, int lo, int hi, double* scale) {
  double d = 0, s = 0; for (int f = lo; f < hi; ++f) for (int j = 0; j < K; ++j) { d = fmax(d, fabs(a[f * K + j] - b[f * K + j])); s = fmax(s, fabs(a[f * K + j])); }
  *scale = s; return d;
}
int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 256, waves = argc > 2 ? atoi(argv[2]) : 4, reps = 20;
  std::mt19937_64 rng(1); std::normal_distribution<double> nd(0.0, 1.0);
  const int NP = 4804; std::vector<double> hp(NP), hz(K * 5), hdz(K * 5), hl(2 * N * NSV);
  for (auto& v : hp) v = 0.3 * nd(rng); for (auto& v : hz) v = nd(rng); for (auto& v : hdz) v = 0.1 * nd(rng); for (auto& v : hl) v = nd(rng);
  Bufs b; double *dp, *dz_, *ddz, *dl;
  CK(hipMalloc(&dp, NP * 8)); CK(hipMalloc(&dz_, K * 5 * 8)); CK(hipMalloc(&ddz, K * 5 * 8)); CK(hipMalloc(&dl, 2 * N * NSV * 8));
  CK(hipMemcpy(dp, hp.data(), NP * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dz_, hz.data(), K * 5 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(ddz, hdz.data(), K * 5 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dl, hl.data(), 2 * N * NSV * 8, hipMemcpyHostToDevice));
  b.params = dp; b.z = dz_; b.dz = ddz; b.lam = dl;
  CK(hipMalloc(&b.pt, (size_t)blocks * PTN * K * 8)); CK(hipMalloc(&b.hb, (size_t)blocks * NT * NM::HB_TILE * 8)); CK(hipMalloc(&b.tb, (size_t)blocks * NT * NM::MB_TILE * 8));
  CK(hipMalloc(&b.sF, K * NSV * 8)); CK(hipMalloc(&b.cyc, 8));
  std::vector<double> p_old(PTN * K), p_new(PTN * K), f0(K * NSV);
  auto get = [&](std::vector<double>& v) { CK(hipMemcpy(v.data(), b.pt, PTN * K * 8, hipMemcpyDeviceToHost)); };
  CK(hipMemset(b.pt, 0, (size_t)blocks * PTN * K * 8));
  const long long c1 = run<1>(b, blocks, waves, reps, 0, 0.0, 0), c2 = run<2>(b, blocks, waves, reps, 0, 0.0, 0); get(p_old);
  CK(hipMemset(b.pt, 0, (size_t)blocks * PTN * K * 8));
  const long long c3c = run<3>(b, blocks, waves, reps, 0, 0.0, 1), c4 = run<4>(b, blocks, waves, reps, 0, 0.0, 1); get(p_new);
  double s; double d;
  d = maxdiff(p_old, p_new, 0, 4, &s); printf("MODE 3 (activations computed) vs MODE 1: F  max diff %.3e (scale %.3e)\n", d, s);
  d = maxdiff(p_old, p_new, 4, 24, &s); printf("                                         A,B max diff %.3e (scale %.3e)\n", d, s);
  d = maxdiff(p_old, p_new, 24, 39, &s); printf("MODE 4 vs MODE 2:                        D2 max diff %.3e (scale %.3e)\n", d, s);
  // activations stored by MODE 0 at alpha = 0 (the trial point IS the iterate), then MODE 3 / 4 from the store
  const long long c0 = run<0>(b, blocks, waves, reps, 0, 0.0, 0), c0s = run<0>(b, blocks, waves, reps, 0, 0.0, 1);
  CK(hipMemcpy(f0.data(), b.sF, K * NSV * 8, hipMemcpyDeviceToHost));
  CK(hipMemset(b.pt, 0, (size_t)blocks * PTN * K * 8));
  const long long c3s = run<3>(b, blocks, waves, reps, 1, 0.0, 1), c4s = run<4>(b, blocks, waves, reps, 0, 0.0, 1);
  std::vector<double> p_st(PTN * K); get(p_st);
  d = maxdiff(p_new, p_st, 0, 39, &s); printf("MODE 3 / 4 from MODE 0's stored activations vs computed: max diff %.3e (bitwise equal expected)\n", d);
  double dF = 0; for (int j = 0; j < K; ++j) for (int r = 0; r < NSV; ++r) dF = fmax(dF, fabs(f0[j * NSV + r] - p_new[r * K + j]));
  printf("MODE 0 values vs MODE 3 values: max diff %.3e\n", dF);
  const int tiles_w0 = (NT + waves - 1) / waves;
  printf("blocks %d, waves %d (wavefront 0: %d tiles of 16 points); cycles per pass / per tile:\n", blocks, waves, tiles_w0);
  printf("  MODE 0 %lld / %lld   MODE 0 + store %lld / %lld\n", c0, c0 / tiles_w0, c0s, c0s / tiles_w0);
  printf("  MODE 1 %lld / %lld   MODE 2 %lld / %lld   (round 2-4)\n", c1, c1 / tiles_w0, c2, c2 / tiles_w0);
  printf("  MODE 3 computing activations %lld / %lld   from the store %lld / %lld\n", c3c, c3c / tiles_w0, c3s, c3s / tiles_w0);
  printf("  MODE 4 %lld / %lld\n", c4s, c4s / tiles_w0);
  return 0;
}
