// dev tool: leave a bit pattern in the LDS of every CU and in freed device memory, so that a kernel launched afterwards finds it wherever
// it reads LDS / scratch it has not written (LDS is not cleared between kernels; a freed allocation is handed out again as it is).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC ldsfill.hip -o ../libldsfill.so
#include <hip/hip_runtime.h>
__global__ void lds_fill_kernel(unsigned long long pat, int ndoubles) {
  extern __shared__ double s[];
  const double v = __longlong_as_double((long long)pat);
  for (int i = threadIdx.x; i < ndoubles; i += blockDim.x) s[i] = v;
  __syncthreads();
  // keep the workgroup resident long enough that the grid spreads over every CU
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 20000) {}
  if (s[threadIdx.x] != v) asm volatile("s_nop 0");
}
__global__ void mem_fill_kernel(unsigned long long pat, unsigned long long* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pat;
}
extern "C" int lds_fill(unsigned long long pat) {
  const int bytes = 160 * 1024;
  if (hipFuncSetAttribute((const void*)lds_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
  hipLaunchKernelGGL(lds_fill_kernel, dim3(2048), dim3(256), bytes, 0, pat, bytes / 8);
  return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
extern "C" int mem_fill(unsigned long long pat, size_t bytes) {
  unsigned long long* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) return 1;
  hipLaunchKernelGGL(mem_fill_kernel, dim3(4096), dim3(256), 0, 0, pat, p, bytes / 8);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  return hipFree(p) == hipSuccess ? 0 : 3;
}
