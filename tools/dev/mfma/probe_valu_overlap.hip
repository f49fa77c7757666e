// dev tool: does fp64 vector work overlap with fp64 matrix instructions in ONE wavefront on gfx950?  (On this part the fp64 matrix rate equals the
// fp64 vector rate: if v_mfma_f64_16x16x4_f64 runs on the vector pipe's multipliers, independent v_fma_f64 between matrix instructions ADD to the time.)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/dev/mfma/probe_valu_overlap.hip -o build/probe_valu_overlap && build/probe_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NV, int KIND>     // NV vector instructions after every matrix instruction; KIND 0: v_fma_f64, 1: v_fma_f32, 2: v_mul_f64, 3: v_add_u32
__global__ void k(double* out, long long* cyc, int reps) {
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + 1e-9 * threadIdx.x;
  double v[8]; for (int i = 0; i < 8; ++i) v[i] = x + i;
  float fv[8]; for (int i = 0; i < 8; ++i) fv[i] = (float)x + i;
  unsigned uv[8]; for (int i = 0; i < 8; ++i) uv[i] = threadIdx.x + i;
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      d4& acc = u == 0 ? a0 : (u == 1 ? a1 : (u == 2 ? a2 : a3));
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[i & 7]) : "v"(y));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fv[i & 7]) : "v"((float)y));
        if (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i & 7]) : "v"(y));
        if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(uv[i & 7]) : "v"(uv[(i + 1) & 7]));
      }
    }
  }
  const long long t1 = clock64();
  double s = a0[0] + a1[1] + a2[2] + a3[3]; for (int i = 0; i < 8; ++i) s += v[i] + fv[i] + uv[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV, int KIND> void run(const char* name, double* out, long long* cyc) {
  const int reps = 2000;
  k<NV, KIND><<<1, 64>>>(out, cyc, reps); hipDeviceSynchronize();
  k<NV, KIND><<<1, 64>>>(out, cyc, reps); hipDeviceSynchronize();
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-12s %2d per matrix instruction: %7.1f cycles per matrix instruction (+ its %d vector instructions)\n", name, NV, (double)c / (reps * 4), NV);
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  run<0, 0>("none", out, cyc);
  run<4, 0>("v_fma_f64", out, cyc); run<8, 0>("v_fma_f64", out, cyc); run<16, 0>("v_fma_f64", out, cyc); run<32, 0>("v_fma_f64", out, cyc);
  run<8, 2>("v_mul_f64", out, cyc); run<16, 2>("v_mul_f64", out, cyc);
  run<8, 1>("v_fma_f32", out, cyc); run<16, 1>("v_fma_f32", out, cyc); run<32, 1>("v_fma_f32", out, cyc);
  run<8, 3>("v_add_u32", out, cyc); run<16, 3>("v_add_u32", out, cyc); run<32, 3>("v_add_u32", out, cyc);
  return 0;
}
