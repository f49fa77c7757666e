// dev probe: issue / dependent latency of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on gfx950 (one wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__global__ void probe(double* out, long long* cyc) {
  const int l = threadIdx.x;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = 1.0 + l, b = 0.5;
  unsigned long long t0 = now();
#pragma unroll
  for (int i = 0; i < 64; ++i) { c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0); asm volatile("" : "+v"(c0)); }
  unsigned long long t1 = now();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
  }
  unsigned long long t2 = now();
  double x = a;
#pragma unroll
  for (int i = 0; i < 64; ++i) { x = fma(x, b, a); asm volatile("" : "+v"(x)); }
  unsigned long long t3 = now();
  // MFMA whose B operand is the previous result's reg 0 (the chained form the solver uses)
  d4 d = c0;
#pragma unroll
  for (int i = 0; i < 32; ++i) { d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, d[0], c1, 0, 0, 0); asm volatile("" : "+v"(d)); }
  unsigned long long t4 = now();
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
  out[l] = c0[0] + c1[1] + c2[2] + c3[3] + x + d[1];
}
int main() {
  double* d; long long* c; (void)hipMalloc(&d, 8 * 64); (void)hipMalloc(&c, 32);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c);
  long long hc[4]; (void)hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
  printf("s_memtime ticks (100 MHz?) -- dependent-accumulator MFMA x64: %lld; 4 independent accumulators x64: %lld; dependent v_fma_f64 x64: %lld; B-chained MFMA x32: %lld\n", hc[0], hc[1], hc[2], hc[3]);
  return 0;
}
