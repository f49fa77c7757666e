// Reproducer of the select-on-stale-SCC lowering (see tools/dev/scan_scc.py): a wave-uniform f64 compare of two
// LDS values inside a one-lane region, selecting between two constants that is then stored to LDS and also branched on.
//   hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only scc_select.hip -o scc_select.s ; python tools/dev/scan_scc.py scc_select.s
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) double lds_d;
__global__ void k(const double* in, double* out, int n, const double* nu) {
  extern __shared__ char smem[];
  lds_d* s = (lds_d*)reinterpret_cast<double*>(smem);
  for (int i = threadIdx.x; i < n; i += 64) s[i] = in[i];
  __syncthreads();
  long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    s[200] += (double)(wall_clock64() - t0);
    double pi[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool pinned = !(s[59 + c] < s[62 + c]);
      s[20 + c] = pinned ? 1.0 : 0.0;
      pi[c] = pinned ? nu[c] : s[71 + c];
    }
    s[100] = pi[0] + pi[1];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 64) out[i] = s[i];
}
