// dev probe: latency of the pivot / gain chain pieces of the matrix-core Riccati stage (one wave, gfx950)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned long long now() { return __builtin_amdgcn_s_memtime(); }
__device__ inline double rdlane(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x); double e = fma(-x, r, 1.0); r = fma(r, e, r); e = fma(-x, r, 1.0); return fma(r, e, r);
}
__global__ void probe(double* out, long long* cyc, double seed) {
  const int l = threadIdx.x;
  double x = seed + l * 1e-3;
  unsigned long long t[8];
  t[0] = now();
#pragma unroll
  for (int i = 0; i < 32; ++i) { x = __builtin_amdgcn_rcp(x) + 1.5; asm volatile("" : "+v"(x)); }       // rcp + add
  t[1] = now();
#pragma unroll
  for (int i = 0; i < 32; ++i) { x = fast_rcp(x) + 1.5; asm volatile("" : "+v"(x)); }                    // rcp + 4 fma + add
  t[2] = now();
#pragma unroll
  for (int i = 0; i < 32; ++i) { x = rdlane(x, 12) + x * 1e-9; asm volatile("" : "+v"(x)); }             // readlane + mul + add (uses SGPR operand)
  t[3] = now();
  d4 c = {x, x, x, x};
#pragma unroll
  for (int i = 0; i < 16; ++i) {   // MFMA -> readlane of its result -> VALU -> next MFMA's operand
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 0.5, c, 0, 0, 0);
    x = rdlane(c[3], 12) * 1e-3; asm volatile("" : "+v"(x));
  }
  t[4] = now();
#pragma unroll
  for (int i = 0; i < 16; ++i) {   // MFMA -> plain VALU use of its result -> next MFMA's operand
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(x, 0.5, c, 0, 0, 0);
    x = c[3] * 1e-3; asm volatile("" : "+v"(x));
  }
  t[5] = now();
  if (l == 0) for (int i = 0; i < 5; ++i) cyc[i] = t[i + 1] - t[i];
  out[l] = x + c[0];
}
int main() {
  double* d; long long* c; (void)hipMalloc(&d, 8 * 64); (void)hipMalloc(&c, 64);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c, 1.25);
  long long hc[5]; (void)hipMemcpy(hc, c, 40, hipMemcpyDeviceToHost);
  printf("per step: rcp+add %.1f | fast_rcp+add %.1f | readlane+mul+add %.1f | mfma->readlane->mul->mfma %.1f | mfma->mul->mfma %.1f\n",
         hc[0] / 32.0, hc[1] / 32.0, hc[2] / 32.0, hc[3] / 16.0, hc[4] / 16.0);
  return 0;
}
