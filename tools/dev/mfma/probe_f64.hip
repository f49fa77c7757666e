// dev probe: register layout of v_mfma_f64_16x16x4_f64 on gfx950 (D = A[16x4] * B[4x16] + C)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, long long* cyc) {
  const int l = threadIdx.x;
  // hypothesis: A[i][k] in lane 16k+i, B[k][j] in lane 16k+j; encode A[i][k] = 1 + i + 100k, B[k][j] = (k == K0) ? (j == J0) : 0
  for (int K0 = 0; K0 < 4; ++K0) {
    d4 c = {0, 0, 0, 0};
    double a = 1.0 + (l % 16) + 100.0 * (l / 16);
    double b = ((l / 16) == K0 && (l % 16) == 3) ? 1.0 : 0.0;     // picks column 3 of D = A[:, K0]
    d4 d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(K0 * 64 + l) * 4 + r] = d[r];
  }
  // latency / issue probe: dependent chain of 64 MFMAs, then 64 independent ones (4 accumulators)
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = 1.0 + l, b = 0.5;
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  long long t1 = clock64();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  long long t2 = clock64();
  if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
  out[4 * 64 * 4 + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  double* d; long long* c; hipMalloc(&d, 8 * (4 * 64 * 4 + 64)); hipMalloc(&c, 16);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c);
  static double h[4 * 64 * 4 + 64]; long long hc[2];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  for (int K0 = 0; K0 < 4; ++K0) {
    printf("K0=%d: nonzero D entries (lane, reg, value): ", K0);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { double v = h[(K0 * 64 + l) * 4 + r]; if (v != 0) printf("(%d,%d,%g) ", l, r, v); }
    printf("\n");
  }
  printf("64 dependent MFMA f64 16x16x4: %lld cycles (%.1f each); 64 independent (4 acc): %lld cycles (%.1f each)\n", hc[0], hc[0] / 64.0, hc[1], hc[1] / 64.0);
  return 0;
}
