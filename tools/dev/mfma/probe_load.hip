// dev probe (round 6): does the fp64 matrix pipe slow down when the whole chip runs it?  One wavefront per workgroup issues N x 4 independent
// v_mfma_f64_16x16x4_f64; launched on 1, 256 and 1024 workgroups (one per CU / four per CU = one per SIMD).  Reported per launch: wall time per matrix
// instruction of a wavefront (hipEvents), s_memtime ticks per instruction, s_memrealtime (100 MHz) ticks -> the rate of the s_memtime counter.
//   hipcc --offload-arch=gfx950 -O3 tools/dev/mfma/probe_load.hip -o build/probe_load && build/probe_load
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned long long now() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ inline unsigned long long real() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
template <int VALU>
__global__ void load(double* out, long long* cyc, int reps) {
  const int l = threadIdx.x;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = 1.0 + l * 1e-3, b = 0.5, x = a, y = b;
  const unsigned long long t0 = now(), r0 = real();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (VALU) {      // the same time in dependent-free v_fma_f64 instead
#pragma unroll
        for (int q = 0; q < 8; ++q) { x = fma(x, b, a); y = fma(y, b, a); }
        asm volatile("" : "+v"(x), "+v"(y));
      } else {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
      }
    }
  }
  const unsigned long long t1 = now(), r1 = real();
  if (l == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
  out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3] + x + y;
}
int main() {
  double* d; long long* c; (void)hipMalloc(&d, 8 * 64 * 4096); (void)hipMalloc(&c, 32);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int reps = 2000;      // 2000 x 64 matrix instructions per wavefront
  for (int valu = 0; valu < 2; ++valu)
    for (int grid : {1, 256, 512, 1024, 2048}) {
      for (int w = 0; w < 2; ++w) {      // (the first launch of a shape warms up)
        (void)hipEventRecord(e0, 0);
        if (valu) hipLaunchKernelGGL(load<1>, dim3(grid), dim3(64), 0, 0, d, c, reps); else hipLaunchKernelGGL(load<0>, dim3(grid), dim3(64), 0, 0, d, c, reps);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      long long hc[2]; (void)hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
      const double n = (double)reps * 64;
      printf("%s grid %4d: launch %.3f ms; wavefront 0: %.1f s_memtime ticks and %.2f ns (s_memrealtime) per %s; s_memtime runs at %.1f MHz\n", valu ? "v_fma_f64 x16 " : "v_mfma_f64    ",
             grid, ms, hc[0] / n, hc[1] * 10.0 / n, valu ? "16 fma" : "matrix instruction", hc[0] / (hc[1] * 0.01));
    }
  return 0;
}
