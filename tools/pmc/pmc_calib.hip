// Calibration kernel for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md §HBM: the
// counters are only calibrated for 16 B/lane streaming reads).  Streams `n` doubles with the 8 B/lane
// coalesced access pattern the solver kernels use: reads n*8 bytes, writes n*8 bytes, past the 256 MiB L3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_calib_copy8(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i] + 1.0;
}

int main(int argc, char** argv) {
  const size_t n = (argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)1 << 27);  // 1 GiB of doubles
  double *a = nullptr, *b = nullptr;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
  hipMemset(a, 0, n * 8);
  hipMemset(b, 0, n * 8);
  for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(k_calib_copy8, dim3(256 * 16), dim3(256), 0, 0, a, b, n);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
  printf("calib bytes_read_per_launch=%zu bytes_written_per_launch=%zu\n", n * 8, n * 8);
  hipFree(a);
  hipFree(b);
  return 0;
}
