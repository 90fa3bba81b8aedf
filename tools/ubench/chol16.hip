// Microbenchmark: the one-wavefront 16 x 16 Cholesky + inverse of mfma_blocks.h — chol16_wave (a row per lane, DPP broadcast-fmacs) and
// chol16_wave_mfma (accumulator layout, four panels, two MFMAs each: round 5) — the serial piece of the Riccati sweep's stage KKT system and
// of the stage kernel's contact solve: clocks per call, alone on its SIMD and with the other wavefronts busy on MFMA; and the two against
// each other (L, L^-1) and against the identity (L L^-1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../../mpc_benchmark_amd/csrc/mfma_blocks.h"

template <bool BUSY, bool MFMA>
__global__ void __launch_bounds__(512) k_chol(double* out, long long* cyc, int reps, double* dump) {
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
      if (MFMA) ok = chol16_wave_mfma(D, 33, LI, lane) && ok;
      else ok = chol16_wave(D, 33, LI, lane) && ok;
    }
    t1 = clock64();
    if (dump) {
      for (int i = lane; i < 256; i += 64) { const int r = i / 16, c = i % 16; dump[i] = c <= r ? D[r * 33 + c] : 0.0; dump[256 + i] = LI[r * 17 + c]; }
      if (lane == 0) dump[512] = ok ? 1.0 : 0.0;
    }
  } else if (BUSY) {
    const double a = 1e-3 * lane, b = 1.0 + 1e-4 * lane;
    for (int r = 0; r < reps * 40; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + tid] = acc[0] + (ok ? LI[lane] : 0.0);
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// in place (LIb == D, ld 17): the inverse replaces the block (the blocked routines' use)
template <bool MFMA>
__global__ void __launch_bounds__(64) k_inplace(double* dump) {
  __shared__ double D[272];
  const int lane = threadIdx.x;
  for (int i = lane; i < 272; i += 64) { const int r = i / 17, c = i % 17; D[i] = (r == c) ? 20.0 + r : 1.0 / (1.0 + r + c); }
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
  const bool ok = MFMA ? chol16_wave_mfma(D, 17, D, lane) : chol16_wave(D, 17, D, lane);
  __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < 256; i += 64) dump[i] = D[(i / 16) * 17 + i % 16];
  if (lane == 0) dump[256] = ok ? 1.0 : 0.0;
}

int main() {
  double *out, *dump; long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024); hipMalloc(&dump, 4 * 520 * 8);
  const int reps = 200;
  for (int mf = 0; mf < 2; ++mf)
    for (int busy = 0; busy < 2; ++busy) {
      for (int it = 0; it < 2; ++it) {
        double* dp = dump + mf * 520;
        if (mf && busy) hipLaunchKernelGGL((k_chol<true, true>), dim3(1), dim3(512), 0, 0, out, cyc, reps, dp);
        else if (mf) hipLaunchKernelGGL((k_chol<false, true>), dim3(1), dim3(512), 0, 0, out, cyc, reps, dp);
        else if (busy) hipLaunchKernelGGL((k_chol<true, false>), dim3(1), dim3(512), 0, 0, out, cyc, reps, dp);
        else hipLaunchKernelGGL((k_chol<false, false>), dim3(1), dim3(512), 0, 0, out, cyc, reps, dp);
        hipDeviceSynchronize();
      }
      long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("%s (load 16 x 16 from LDS, factor, invert, store)%s: %.0f clocks per call\n", mf ? "chol16_wave_mfma" : "chol16_wave     ",
             busy ? ", seven wavefronts on MFMA beside it" : "", (double)h / reps);
    }
  hipLaunchKernelGGL((k_inplace<false>), dim3(1), dim3(64), 0, 0, dump + 2 * 520);
  hipLaunchKernelGGL((k_inplace<true>), dim3(1), dim3(64), 0, 0, dump + 3 * 520);
  hipDeviceSynchronize();
  static double h[4 * 520];
  hipMemcpy(h, dump, sizeof(h), hipMemcpyDeviceToHost);
  double dl = 0, di = 0, eye[2] = {0, 0}, dip = 0;
  for (int i = 0; i < 256; ++i) { dl = fmax(dl, fabs(h[i] - h[520 + i])); di = fmax(di, fabs(h[256 + i] - h[520 + 256 + i])); dip = fmax(dip, fabs(h[2 * 520 + i] - h[3 * 520 + i])); }
  for (int v = 0; v < 2; ++v)
    for (int r = 0; r < 16; ++r)
      for (int c = 0; c < 16; ++c) {
        double s = 0;
        for (int k = 0; k < 16; ++k) s += h[v * 520 + r * 16 + k] * h[v * 520 + 256 + k * 16 + c];
        eye[v] = fmax(eye[v], fabs(s - (r == c ? 1.0 : 0.0)));
      }
  printf("ok flags %.0f %.0f (in place %.0f %.0f) ; max |L - L'| %.3e ; max |Linv - Linv'| %.3e (in place %.3e) ; max |L Linv - I|: DPP form %.3e, MFMA form %.3e\n",
         h[512], h[520 + 512], h[2 * 520 + 256], h[3 * 520 + 256], dl, di, dip, eye[0], eye[1]);
  return 0;
}
