// Microbenchmark: 16x16 fp64 MFMA tile products on LDS operands exactly as the Riccati sweep issues them (mma_tile of
// mfma_blocks.h, K = 80), 4 / 8 wavefronts per workgroup, against the same MFMA count on register operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mpc_benchmark_amd/csrc/mfma_blocks.h"

// MODE 0: A = PT columns (1, ldp), B = AB rows (ldab, 1)      [G = Pt [A B]]
// MODE 1: A = PT rows (ldp, 1),    B = PT2 rows (ldp, 1)      [series products]
// MODE 2: A = AB columns (1, ldab), B = PT rows (ldp, 1)      [Hh = H + [A B]^T G]
// MODE 3: register operands only
template <int MODE>
__global__ void __launch_bounds__(512) k_tiles(double* out, long long* cyc, int reps, int ldab) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int np = 80, ldp = 81;
  double* PT = sm;
  double* P2 = PT;  // the series multiplies two matrices of the same layout
  double* AB = PT + np * ldp;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  for (int i = tid; i < np * ldp; i += blockDim.x) { PT[i] = 1e-3 * (i % 97); P2[i] = 1e-3 * (i % 89); }
  for (int i = tid; i < np * ldab; i += blockDim.x) AB[i] = 1e-3 * (i % 83);
  __syncthreads();
  d4_t acc = d4_t{0, 0, 0, 0};
  const long long t0 = clock64();
  for (int r = 0; r < reps; ++r) {
    const int ti = (wv + r) % 5, tj = (wv + 2 * r) % 5;
    if (MODE == 0) mma_tile<false>(acc, PT + ti * 16, 1, ldp, AB + tj * 16, ldab, 1, np, lane);
    if (MODE == 1) mma_tile<false>(acc, PT + (ti * 16) * ldp, ldp, 1, P2 + tj * 16, ldp, 1, np, lane);
    if (MODE == 2) mma_tile<false>(acc, AB + ti * 16, 1, ldab, PT + tj * 16, ldp, 1, np, lane);
    if (MODE == 3) {
      const double a = 1e-3 * lane, b = 1.0 + 1e-4 * lane;
      for (int k = 0; k < 20; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  (void)nw;
}

template <int MODE>
static void run(const char* what, int waves, int ldab, double* out, long long* cyc) {
  const int reps = 400;
  const size_t lds = (80 * 81 + 80 * (size_t)ldab) * 8;
  hipFuncSetAttribute((const void*)k_tiles<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_tiles<MODE>, dim3(1), dim3(64 * waves), lds, 0, out, cyc, reps, ldab);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_tiles<MODE>, dim3(1), dim3(64 * waves), lds, 0, out, cyc, reps, ldab);
  hipEventRecord(e1); hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) printf("launch failed\n");
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-44s waves %d ld(AB) %3d : %7.3f us per tile per wave (event), %6.0f ticks ; all waves: %.3f us per tile\n", what, waves, ldab, ms * 1e3 / reps,
         (double)h / reps, ms * 1e3 / reps / waves);
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
  for (int waves : {4, 8}) {
    run<3>("register operands (20 dependent MFMA)", waves, 112, out, cyc);
    run<0>("G: A = Pt cols, B = [A B] rows", waves, 112, out, cyc);
    run<0>("G: A = Pt cols, B = [A B] rows", waves, 113, out, cyc);
    run<0>("G: A = Pt cols, B = [A B] rows", waves, 116, out, cyc);
    run<1>("series: A = T rows, B = Ph rows", waves, 112, out, cyc);
    run<2>("Hh: A = [A B] cols, B = G rows", waves, 112, out, cyc);
    run<2>("Hh: A = [A B] cols, B = G rows", waves, 113, out, cyc);
    run<2>("Hh: A = [A B] cols, B = G rows", waves, 116, out, cyc);
  }
  return 0;
}
