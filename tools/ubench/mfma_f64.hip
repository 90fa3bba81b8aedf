// Microbenchmark: issue rate / dependent latency of v_mfma_f64_16x16x4_f64 on gfx950, 1..8 wavefronts per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k_mfma(double* out, long long* cyc, int iters) {
  d4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = d4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void k_fma(double* out, long long* cyc, int iters) {
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i;
  const double a = 1.0000001, b = 1e-9;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
  }
  const long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
  const int iters = 2000;
  long long h;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {1, 4, 8}) {
    for (int nacc : {1, 2, 4}) {
      float ms = 0;
      hipEventRecord(e0);
      if (nacc == 1) hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      if (nacc == 2) hipLaunchKernelGGL(k_mfma<2>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      if (nacc == 4) hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("mfma_f64_16x16x4: waves/CU=%d independent acc=%d : %.1f clock64 ticks per MFMA per wave, %.3f us per MFMA per wave (event)\n", waves, nacc,
             (double)h / (iters * nacc), ms * 1e3 / (iters * nacc));
    }
  }
  for (int waves : {1, 4, 8}) {
    float ms = 0;
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fma, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("v_fma_f64: waves/CU=%d 8 independent chains : %.2f ticks per FMA instr per wave, %.4f us (event)\n", waves, (double)h / (iters * 8), ms * 1e3 / (iters * 8));
  }
  return 0;
}
