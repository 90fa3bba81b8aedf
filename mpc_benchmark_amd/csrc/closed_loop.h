// closed_loop.h — the forward sweep split into a knot-PARALLEL part and a short sequential part.
//
// The linear rollout of one ProxDDP iteration is, per knot
//     du = K dx + k ;  y = A dx + B du + mx ;  dx' = T (y - mu_d Pt y)
// i.e. an affine recurrence  dx' = Phi dx + phi  with
//     Phi = T (I - mu_d Pt)(A + B K) ,   phi = T (I - mu_d Pt)(B k + mx) .
// None of K, [A B], Pt, T depends on the recursion, so k_closed_loop builds Phi / phi for every knot of every instance in
// parallel on the matrix cores (one workgroup per knot), and the sequential sweep k_forward_phi is left with ONE mat-vec
// and ONE barrier per knot (rows of Phi for dx', rows of K for du, all reading the same dx).
#pragma once
#include "mfma_blocks.h"
#include "solver_kernels.h"

struct ClLds {
  int np, mp, nzp, lda, ldp, nb, nbm;
  int R0, KM, ACL, Z0, vec, total_bytes;  // R0: [A B] (stage 1) then Pt (stage 2)
};

static inline ClLds make_cl_lds(int n, int m) {
  ClLds s;
  s.np = (n + 15) & ~15; s.mp = (m + 15) & ~15; s.nzp = s.np + s.mp; s.lda = s.np + 1; s.ldp = s.np + 1;
  s.nb = s.np / 16; s.nbm = s.mp / 16;
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
  const int r0 = s.np * s.nzp > s.np * s.ldp ? s.np * s.nzp : s.np * s.ldp;
  s.R0 = take(r0); s.KM = take(s.mp * s.np); s.ACL = take(s.np * s.lda); s.Z0 = take(16 * s.np); s.vec = take(4 * s.np + s.mp + 40);
  s.total_bytes = o * 8;
  return s;
}

#define CL_THREADS 512
#define CL_U 20  // global loads in flight per thread in the LDS fills: one batch covers [A B] for np <= 96, nzp <= 128 (x 512 threads)
#define CL_PT 18  // Pt elements per thread parked in registers during stage 1 (np <= 96)
#define CL_TILES ((36 * 64 + CL_THREADS - 1) / CL_THREADS)  // output tiles per wavefront, nb <= 6

// grid (N, B): knot k of instance b.  Reads K, k (gain record), [A B] (knot record), Pt, mx, T6 (gain record);
// writes Phi (n x n) and phi (n) into the gain record.
__global__ void __launch_bounds__(CL_THREADS) k_closed_loop(SolverArgs a, ClLds S) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, nw = nthr >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, nz = L.nz, np = S.np, mp = S.mp, nzp = S.nzp, lda = S.lda, ldp = S.ldp, nb = S.nb;
  const double* kn = knot_ptr(a, b, k);
  double* g = gain_ptr(a, b, k);
  const int m = (int)kn[L.oMISC + MISC_M];
  const double mud = st.mu * a.opt.dyn_al_scale;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *AB = sm + S.R0, *PT = sm + S.R0, *KM = sm + S.KM, *ACL = sm + S.ACL, *Z0 = sm + S.Z0, *vec = sm + S.vec;
  double *kf = vec, *y0 = vec + mp, *z0 = y0 + np, *t6 = z0 + np;  // k (mp), B k + mx (np), (I - mu Pt) y0 (np), T6 (36)

  // ---- stage 1: [A B] and K into LDS (zero padded; u-columns of [A B] start at np) ; A_cl = A + B K ; y0 = B k + mx ----
  // (copies are unrolled by hand, CL_U loads in flight per thread: a plain strided loop serialises on the HBM latency)
  for (int base = tid; base < np * nzp; base += nthr * CL_U) {
    double v[CL_U];
#pragma unroll
    for (int u = 0; u < CL_U; ++u) {
      const int idx = base + u * nthr, i = idx / nzp, zp = idx % nzp;
      const int z = (zp < n) ? zp : ((zp >= np && zp - np < m) ? n + zp - np : -1);
      v[u] = (idx < np * nzp && i < n && z >= 0) ? kn[L.oAB + (size_t)i * nz + z] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < CL_U; ++u) { const int idx = base + u * nthr; if (idx < np * nzp) AB[idx] = v[u]; }
  }
  for (int base = tid; base < mp * np; base += nthr * CL_U) {
    double v[CL_U];
#pragma unroll
    for (int u = 0; u < CL_U; ++u) {
      const int idx = base + u * nthr, l = idx / np, j = idx % np;
      v[u] = (idx < mp * np && l < m && j < n) ? g[L.oK + l * n + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < CL_U; ++u) { const int idx = base + u * nthr; if (idx < mp * np) KM[idx] = v[u]; }
  }
  for (int l = tid; l < mp; l += nthr) kf[l] = (l < m) ? g[L.ok + l] : 0.0;
  if (tid < 36) t6[tid] = g[L.oT6 + tid];
  // Pt is requested now and parked in registers: its HBM latency hides behind stage 1
  double ptv[CL_PT];
#pragma unroll
  for (int u = 0; u < CL_PT; ++u) {
    const int idx = tid + u * nthr, i = idx / np, j = idx % np;
    ptv[u] = (idx < np * np && i < n && j < n) ? g[L.oMx + i * n + j] : 0.0;
  }
  __syncthreads();
  const int ntile = nb * nb;
#pragma unroll
  for (int sidx = 0; sidx < CL_TILES; ++sidx) {
    const int t = wv + sidx * nw;
    if (t < ntile) {
      const int ri = t / nb, cj = t % nb;
      d4_t acc = tile_load(AB + (ri * 16) * nzp + cj * 16, nzp, lane);                              // A tile
      mma_tile<false>(acc, AB + (ri * 16) * nzp + np, nzp, 1, KM + cj * 16, np, 1, mp, lane);        // + B K
      tile_store(ACL + (ri * 16) * lda + cj * 16, lda, acc, lane);
    }
  }
  for (int i = tid; i < np; i += nthr) {
    double s = (i < n) ? g[L.omx + i] : 0.0;
    for (int l = 0; l < m; ++l) s += AB[i * nzp + np + l] * kf[l];
    y0[i] = s;
  }
  __syncthreads();
  // ---- stage 2: Pt into LDS (over [A B]) ; M = A_cl - mu_d Pt A_cl, tiles straight to the gain record except the row
  // tile that holds the base rows (T6 couples rows 0..5) ; z0 = y0 - mu_d Pt y0 ----
#pragma unroll
  for (int u = 0; u < CL_PT; ++u) { const int idx = tid + u * nthr; if (idx < np * np) PT[(idx / np) * ldp + idx % np] = ptv[u]; }
  __syncthreads();
#pragma unroll
  for (int sidx = 0; sidx < CL_TILES; ++sidx) {
    const int t = wv + sidx * nw;
    if (t < ntile) {
      const int ri = t / nb, cj = t % nb;
      d4_t acc = d4_t{0, 0, 0, 0};
      mma_tile<false>(acc, PT + (ri * 16) * ldp, ldp, 1, ACL + cj * 16, lda, 1, np, lane);
      const int col = cj * 16 + (lane & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = ri * 16 + (lane >> 4) + 4 * q;
        const double v = ACL[row * lda + col] - mud * acc[q];
        if (ri == 0) Z0[row * np + col] = v;
        else if (row < n && col < n) g[L.oPhi + row * n + col] = v;
      }
    }
  }
  for (int i = wv; i < n; i += nw) {
    double s = 0;
    for (int j = lane; j < n; j += 64) s += PT[i * ldp + j] * y0[j];
    s = wave_sum(s);
    if (lane == 0) z0[i] = y0[i] - mud * s;
  }
  __syncthreads();
  // ---- rows 0..15 of Phi (base rows through T6) and phi ----
  for (int idx = tid; idx < 16 * n; idx += nthr) {
    const int i = idx / n, j = idx % n;
    if (i >= n) continue;
    double s = Z0[i * np + j];
    if (i < 6) { s = 0; for (int l = 0; l < 6; ++l) s += t6[i * 6 + l] * Z0[l * np + j]; }
    g[L.oPhi + i * n + j] = s;
  }
  for (int i = tid; i < n; i += nthr) {
    double s = z0[i];
    if (i < 6) { s = 0; for (int l = 0; l < 6; ++l) s += t6[i * 6 + l] * z0[l]; }
    g[L.ophi + i] = s;
  }
}

// Sequential part: dx' = Phi dx + phi, du = K dx + k.  One workgroup (8 wavefronts) per instance; the rows of Phi and K of
// knot k + 1 are pulled into registers while knot k computes (FW_PR rows of Phi, FW_KR rows of K per wavefront, two
// columns per lane: n <= 128).  One barrier per knot.  (A second register set to look two knots ahead does not fit the
// 256 registers of an 8-wavefront workgroup without spilling, and 16 wavefronts spill as well: measured slower.)
// Parallel-in-time legs (legs.h): the grid is B x nlegs, workgroup (b, leg) sweeps the knots of its leg from the cut state the
// consensus kernel left in dxs (leg 0: dx_0 = 0); the state at the next cut is NOT written (it is the consensus value).  nlegs = 1 is
// the sweep over the whole horizon.
template <int FW_KR, int FW_PR>
__global__ void __launch_bounds__(512) k_forward_phi(SolverArgs a) {
  const Layout& L = a.L;
  const int b = blockIdx.x % L.B, leg = blockIdx.x / L.B, tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, N = L.N, m = L.m;
  const bool cut_end = leg + 1 < a.nlegs;
  const int k0 = leg_start(a, leg), k1 = cut_end ? leg_start(a, leg + 1) : N;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* dx0 = lds;       // dx of the current knot (n)
  double* dx1 = lds + n;   // dx of the next knot (n): ping-pong, so a single barrier per knot suffices
  double Kr[FW_KR][2], kf[FW_KR], Pr[FW_PR][2], pf[FW_PR];
  const int c0 = lane, c1 = lane + 64;
  const int cn0 = c0 < n ? c0 : n - 1, cn1 = c1 < n ? c1 : n - 1;
  auto load_rows = [&](int k) {
    const double* g = gain_ptr(a, b, k);
#pragma unroll
    for (int q = 0; q < FW_KR; ++q) {
      const int r = wv + nw * q, rr = r < m ? r : 0;
      Kr[q][0] = g[L.oK + rr * n + cn0]; Kr[q][1] = g[L.oK + rr * n + cn1]; kf[q] = g[L.ok + rr];
    }
#pragma unroll
    for (int q = 0; q < FW_PR; ++q) {
      const int r = wv + nw * q, rr = r < n ? r : 0;
      Pr[q][0] = g[L.oPhi + rr * n + cn0]; Pr[q][1] = g[L.oPhi + rr * n + cn1]; pf[q] = g[L.ophi + rr];
    }
  };
  load_rows(k0);
  for (int i = tid; i < n; i += blockDim.x) {
    if (leg == 0) { dx0[i] = 0.0; a.dxs[(size_t)b * (N + 1) * n + i] = 0.0; }
    else dx0[i] = a.dxs[((size_t)b * (N + 1) + k0) * n + i];
  }
  FW_BARRIER();
  for (int k = k0; k < k1; ++k) {
    const double* cur = ((k - k0) & 1) ? dx1 : dx0;
    double* nxt = ((k - k0) & 1) ? dx0 : dx1;
    const bool keep = !(cut_end && k + 1 == k1);
    const double d0 = c0 < n ? cur[c0] : 0.0, d1 = c1 < n ? cur[c1] : 0.0;
#pragma unroll
    for (int q = 0; q < FW_PR; ++q) {
      const int r = wv + nw * q;
      if (r < n) {
        double s = Pr[q][0] * d0 + Pr[q][1] * d1;
        s = wave_sum(s);
        if (lane == 0) { s += pf[q]; nxt[r] = s; if (keep) a.dxs[((size_t)b * (N + 1) + k + 1) * n + r] = s; }
      }
    }
#pragma unroll
    for (int q = 0; q < FW_KR; ++q) {
      const int r = wv + nw * q;
      if (r < m) {
        double s = Kr[q][0] * d0 + Kr[q][1] * d1;
        s = wave_sum(s);
        if (lane == 0) a.dus[((size_t)b * N + k) * m + r] = s + kf[q];
      }
    }
    if (k + 1 < k1) load_rows(k + 1);
    FW_BARRIER();
  }
}
