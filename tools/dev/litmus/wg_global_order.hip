// Litmus test of an assumption the compiler makes for gfx942 / gfx950 (LLVM AMDGPUUsage, memory model, "fence release - workgroup":
// "if not TgSplit execution mode, omit vmcnt(0)"): global stores of one wavefront, a workgroup barrier WITHOUT s_waitcnt vmcnt(0),
// loads of the same addresses by ANOTHER wavefront of the workgroup -- are the stores always seen?  The fused solver's W > 1 forms hand
// their per-point / per-stage records from wavefront to wavefront exactly this way (hs_solver_fused.h: wsync()).
//   hipcc --offload-arch=gfx950 -O3 wg_global_order.hip -o wg_global_order ; ./wg_global_order [MB per block] [rounds] [blocks] [strong]
// `strong` = 1 inserts s_waitcnt vmcnt(0) in front of the barrier (the control experiment).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

template <bool STRONG>
__global__ __launch_bounds__(128, 1) void litmus(double* buf, long doubles_per_block, long round_stride, int rounds, int rec, unsigned long long* bad, long* where) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double* p = buf + (long)blockIdx.x * doubles_per_block;
  for (int r = 0; r < rounds; ++r) {
    double* q = p + (long)r * round_stride;          // memory no earlier round has touched
    const int writer = r & 1;
    if (wave == writer) {
      // an AoS record of `rec` doubles per lane (what the hessian / backward passes write)
      for (int i = 0; i < rec; ++i) q[(long)lane * rec + i] = (double)(r + 1) * 4096.0 + lane * 64 + i;
    }
    if (STRONG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave != writer) {
      // the reader takes the records with another lane mapping (lanes over fields, like the sweep's prefetch ring)
      for (int i = 0; i < rec; ++i) {
        const long idx = (long)i * 64 + lane;         // covers [0, 64 * rec)
        const int l = (int)(idx / rec), f = (int)(idx % rec);
        const double want = (double)(r + 1) * 4096.0 + l * 64 + f;
        const double got = q[idx];
        if (got != want) {
          const unsigned long long n = atomicAdd(bad, 1ULL);
          if (n < 8) { where[3 * n] = blockIdx.x; where[3 * n + 1] = r; where[3 * n + 2] = idx; }
        }
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const long mb = argc > 1 ? atol(argv[1]) : 64;
  const int rounds = argc > 2 ? atoi(argv[2]) : 256;
  const int blocks = argc > 3 ? atoi(argv[3]) : 256;
  const int strong = argc > 4 ? atoi(argv[4]) : 0;
  const int rec = 25;
  const long per_block = mb * 1024 * 1024 / 8;
  const long round_stride = per_block / rounds;
  if (round_stride < 64 * rec) { printf("too many rounds for the block size\n"); return 1; }
  double* buf; unsigned long long* bad; long* where;
  CHK(hipMalloc(&buf, (size_t)per_block * 8 * blocks));          // fresh: the kernel's stores are the first touch
  CHK(hipMalloc(&bad, 8)); CHK(hipMalloc(&where, 24 * 8));
  CHK(hipMemset(bad, 0, 8)); CHK(hipMemset(where, 0, 24 * 8));
  if (strong) hipLaunchKernelGGL(litmus<true>, dim3(blocks), dim3(128), 0, 0, buf, per_block, round_stride, rounds, rec, bad, where);
  else hipLaunchKernelGGL(litmus<false>, dim3(blocks), dim3(128), 0, 0, buf, per_block, round_stride, rounds, rec, bad, where);
  CHK(hipDeviceSynchronize());
  unsigned long long nb = 0; long w[24];
  CHK(hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(w, where, 24 * 8, hipMemcpyDeviceToHost));
  printf("%s barrier, %d blocks x %d rounds x %d values, %ld MB per block (round stride %ld KB): %llu stale reads", strong ? "vmcnt(0) +" : "plain",
         blocks, rounds, 64 * rec, mb, round_stride * 8 / 1024, nb);
  for (unsigned long long i = 0; i < nb && i < 4; ++i) printf(" [block %ld round %ld index %ld]", w[3 * i], w[3 * i + 1], w[3 * i + 2]);
  printf("\n");
  return nb ? 1 : 0;
}
