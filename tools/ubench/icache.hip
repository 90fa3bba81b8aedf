// Microbenchmark: instruction fetch.  A loop whose straight-line body is BODY_KB of 8-byte VALU instructions, walked by 8
// wavefronts of one workgroup: ticks per instruction when the body fits the instruction cache and when it does not.
#include <hip/hip_runtime.h>
#include <cstdio>
#define I1 asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
#define I4 I1 I1 I1 I1
#define I16 I4 I4 I4 I4
#define I64 I16 I16 I16 I16
#define I256 I64 I64 I64 I64
#define I1K I256 I256 I256 I256
#define I4K I1K I1K I1K I1K

template <int KB>
__global__ void __launch_bounds__(512) k_body(float* out, long long* cyc, int iters) {
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (KB >= 8) { I1K }
    if (KB >= 32) { I1K I1K I1K }
    if (KB >= 64) { I4K }
    if (KB >= 128) { I4K I4K }
    if (KB >= 256) { I4K I4K I4K I4K }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KB>
static void run(int waves, int blocks, float* out, long long* cyc) {
  const int ninstr = KB * 1024 / 8, iters = 4096 / KB * 8;
  hipLaunchKernelGGL(k_body<KB>, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k_body<KB>, dim3(blocks), dim3(64 * waves), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("body %3d KB, %d waves x %3d workgroups: %.2f ticks per instruction per wave\n", KB, waves, blocks, (double)h / ((double)ninstr * iters));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8192);
  for (int blocks : {1, 256}) for (int waves : {1, 8}) {
    run<8>(waves, blocks, out, cyc); run<32>(waves, blocks, out, cyc); run<64>(waves, blocks, out, cyc); run<128>(waves, blocks, out, cyc); run<256>(waves, blocks, out, cyc);
  }
  return 0;
}
