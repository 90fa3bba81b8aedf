// Microbenchmark: the one-wavefront 16 x 16 Cholesky + inverse of mfma_blocks.h (chol16_wave), the serial piece of the Riccati sweep's
// stage KKT system and of the stage kernel's contact solve: ticks per call, alone on its SIMD and with the other wavefronts busy on MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mpc_benchmark_amd/csrc/mfma_blocks.h"

template <bool BUSY>
__global__ void __launch_bounds__(512) k_chol(double* out, long long* cyc, int reps) {
  __shared__ double D[16 * 33], LI[272], A0[16 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 16 * 33; i += blockDim.x) { const int r = i / 33, c = i % 33; A0[i] = (r == c) ? 20.0 + r : 1.0 / (1.0 + r + c); }
  __syncthreads();
  d4_t acc = d4_t{0, 0, 0, 0};
  long long t0 = clock64(), t1 = t0;
  bool ok = true;
  if (wv == 0) {
    for (int r = 0; r < reps; ++r) {
      for (int i = lane; i < 16 * 33; i += 64) D[i] = A0[i];
      ok = chol16_wave(D, 33, LI, lane) && ok;
    }
    t1 = clock64();
  } else if (BUSY) {
    const double a = 1e-3 * lane, b = 1.0 + 1e-4 * lane;
    for (int r = 0; r < reps * 40; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + tid] = acc[0] + (ok ? LI[lane] : 0.0);
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
  const int reps = 200;
  for (int busy = 0; busy < 2; ++busy) {
    for (int it = 0; it < 2; ++it) {
      if (busy) hipLaunchKernelGGL(k_chol<true>, dim3(1), dim3(512), 0, 0, out, cyc, reps);
      else hipLaunchKernelGGL(k_chol<false>, dim3(1), dim3(512), 0, 0, out, cyc, reps);
      hipDeviceSynchronize();
    }
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("chol16_wave (load 16 x 16 from LDS, factor, invert, store)%s: %.0f ticks per call\n", busy ? ", seven wavefronts on MFMA beside it" : "", (double)h / reps);
  }
  return 0;
}
