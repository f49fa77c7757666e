// Litmus test: does v_mfma_f64_16x16x4_f64 honour the EXEC mask on gfx950?  (the W > 1 kernels of round 3 ran
// their matrix-core sweep inside an EXEC-masked region.)  The accumulator is preset to 7; the instruction runs with EXEC = `mask`;
// lanes outside the mask must keep 7 if the instruction is masked like any vector instruction.
//   hipcc --offload-arch=gfx950 -O2 mfma_exec.hip -o mfma_exec ; ./mfma_exec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned long long mask, double* out) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane, b = 2.0;
  d4 c = {7.0, 7.0, 7.0, 7.0};
  asm volatile(
      "s_mov_b64 s[10:11], exec\n\t"
      "s_mov_b64 exec, %3\n\t"
      "s_nop 4\n\t"
      "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\t"
      "s_nop 15\n\t"
      "s_nop 15\n\t"
      "s_mov_b64 exec, s[10:11]\n\t"
      : "+v"(c) : "v"(a), "v"(b), "s"(mask) : "s10", "s11", "memory");
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}
int main() {
  double* d; hipMalloc(&d, 64 * 4 * 8);
  const unsigned long long masks[4] = {~0ULL, 0ULL, 1ULL, 0x00000000FFFFFFFFULL};
  for (int m = 0; m < 4; ++m) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, masks[m], d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int changed_in = 0, changed_out = 0, n_in = 0, n_out = 0;
    for (int l = 0; l < 64; ++l) {
      const bool in = (masks[m] >> l) & 1;
      bool ch = false; for (int i = 0; i < 4; ++i) ch = ch || h[l * 4 + i] != 7.0;
      if (in) { ++n_in; changed_in += ch; } else { ++n_out; changed_out += ch; }
    }
    printf("EXEC = %016llx: accumulator written in %d of %d enabled lanes, in %d of %d DISABLED lanes\n", masks[m], changed_in, n_in, changed_out, n_out);
  }
  return 0;
}
