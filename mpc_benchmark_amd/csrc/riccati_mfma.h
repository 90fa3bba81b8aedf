// riccati_mfma.h — proximal Riccati backward sweep with every matrix operand
// resident in LDS and the dense n x n / n x (n+m) products on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64, one 16x16 output tile per wavefront instruction).
//
// One workgroup (256 threads = 4 wavefronts, one per SIMD) per MPC instance walks the horizon backwards:
//   Ph = T^T P' T                         base-frame change of the co-state (6 x 6 block)
//   L L^T = I + mu_d Ph                   Cholesky in LDS
//   Pt = (I + mu_d Ph)^-1 Ph              blocked triangular solves, diagonal blocks pre-inverted -> pure MFMA
//   G = Pt [A B] ; Hh = H + [A B]^T G     16-column panels: G panel in LDS, Hh panel to L2-resident scratch
//   stage KKT (controls, then ACTIVE constraint rows), gains K, k, Knu, knu
//   P = Qh + Sh K + Ca^T Knu              MFMA, result stays in LDS as next knot's P'
// dims are padded to multiples of 16 inside LDS (np, mp); u-columns start at column np.
#pragma once
#include "solver_kernels.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

struct RicLds {
  int np, mp, nzp, ldl, nb, nbm, cap;  // padded dims, leading dim of L, #16-blocks of n and m, active-row capacity in LDS
  int PT, R1, LP, LI, AB, GP, vec, iwork, total_bytes;
  // KKT / value-update workspace carved from R1 once AB is dead
  int Lr, W, ST, CT, VX, Y, SC, lw;
};

static inline RicLds make_ric_lds(int n, int m, int c) {
  RicLds s;
  s.np = (n + 15) & ~15; s.mp = (m + 15) & ~15; s.nzp = s.np + s.mp; s.ldl = s.np + 1; s.nb = s.np / 16; s.nbm = s.mp / 16;
  s.cap = 16;
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
  s.PT = take(s.np * s.np);
  s.R1 = o;
  // phase 1 view of R1
  s.LP = take(s.np * s.ldl); s.LI = take(s.nb * 16 * 17);
  const int end1 = o;
  // phase 2 view of R1 (overlaps phase 1)
  o = s.R1;
  s.AB = take(s.np * s.nzp); s.GP = take(s.np * 16);
  const int end2 = o;
  // phase 3 view of R1 (overlaps AB)
  o = s.R1;
  s.lw = s.np + 16;  // W = [K | pad | k | pad]: x-columns at 0..n-1, the feed-forward column at np
  s.Lr = take(m * m); s.W = take(s.mp * s.lw); s.ST = take(s.mp * s.np);
  s.CT = take(s.cap * s.np); s.VX = take(s.cap * s.np); s.Y = take(m * s.cap); s.SC = take(s.cap * s.cap);
  const int end3 = o;
  o = end1 > end2 ? end1 : end2;
  if (end3 > o) o = end3;
  s.vec = take(8 * (s.nzp + c) + 64);
  s.iwork = o;
  s.total_bytes = o * 8 + (c + 8) * 4;
  return s;
}

// acc += sum_{k<K} A(i,k) * B(k,j) for one 16x16 tile; A(i,k) at A[i*a_is + k*a_ks], B(k,j) at B[k*b_ks + j*b_js].
// Fragment layout of v_mfma_f64_16x16x4_f64: lane l supplies A(l&15, l>>4) and B(l>>4, l&15); result register r of
// lane l is C((l>>4) + 4r, l&15).
template <bool NEG>
DEV void mma_tile(d4_t& acc, const double* A, int a_is, int a_ks, const double* B, int b_ks, int b_js, int K, int lane) {
  const int i = lane & 15, kk = lane >> 4;
  const double* ap = A + i * a_is + kk * a_ks;
  const double* bp = B + kk * b_ks + i * b_js;
  for (int k0 = 0; k0 < K; k0 += 4) {
    const double av = NEG ? -ap[k0 * a_ks] : ap[k0 * a_ks];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[k0 * b_ks], acc, 0, 0, 0);
  }
}
DEV d4_t tile_load(const double* C, int ldc, int lane) {
  d4_t r;
  const int row = lane >> 4, col = lane & 15;
  for (int q = 0; q < 4; ++q) r[q] = C[(row + 4 * q) * ldc + col];
  return r;
}
DEV void tile_store(double* C, int ldc, const d4_t& v, int lane) {
  const int row = lane >> 4, col = lane & 15;
  for (int q = 0; q < 4; ++q) C[(row + 4 * q) * ldc + col] = v[q];
}

__global__ void __launch_bounds__(256) k_riccati_mfma(SolverArgs a, RicLds S) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, nz = L.nz, N = L.N, nr = n + 1;
  const int np = S.np, mp = S.mp, nzp = S.nzp, ldl = S.ldl, nb = S.nb, nzt = nzp / 16, lw = S.lw;
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const bool ff = L.space == MPC_SPACE_MULTIBODY;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *PT = sm + S.PT, *LP = sm + S.LP, *LI = sm + S.LI, *AB = sm + S.AB, *GP = sm + S.GP, *vec = sm + S.vec;
  double *Lr = sm + S.Lr, *W = sm + S.W, *ST = sm + S.ST, *CTl = sm + S.CT, *VXl = sm + S.VX, *Yl = sm + S.Y, *SCl = sm + S.SC;
  int* act_idx = (int*)(sm + S.iwork);
  int* iflag = act_idx + L.c;
  double *ph = vec, *ft = vec + nzp, *vv = vec + 2 * nzp, *w = vec + 3 * nzp, *gh = vec + 4 * nzp, *pvec = vec + 5 * nzp, *dtl = vec + 6 * nzp;
  double* wk = a.work + (size_t)b * L.work_stride;
  double* Hh = wk + L.wHh;  // nz x nz, leading dimension nz (L2-resident scratch)

  // ---- terminal node: P_N = H + Ca^T Ca / mu ; p_N = grad + Ca^T dt / mu ----
  {
    const double* kn = knot_ptr(a, b, N);
    double* g = gain_ptr(a, b, N);
    const int c = (int)kn[L.oMISC + MISC_NC];
    for (int i = tid; i < c; i += nthr) g[L.oknu + i] = kn[L.oDT + i] / mu;
    for (int idx = tid; idx < c * n; idx += nthr) {
      const int i = idx / n, z = idx % n;
      g[L.oKnu + idx] = (kn[L.oACT + i] != 0.0) ? kn[L.oCD + i * nz + z] / mu : 0.0;
    }
    for (int idx = tid; idx < np * np; idx += nthr) PT[idx] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int r = idx / n, s = idx % n;
      double t = kn[L.oH + r * nz + s];
      for (int i = 0; i < c; ++i) if (kn[L.oACT + i] != 0.0) t += kn[L.oCD + i * nz + r] * g[L.oKnu + i * n + s];
      g[L.oP + idx] = t;
      PT[r * np + s] = t;
    }
    for (int r = tid; r < n; r += nthr) {
      double t = kn[L.oG + r];
      for (int i = 0; i < c; ++i) t += kn[L.oCD + i * nz + r] * g[L.oknu + i];
      g[L.op + r] = t;
      pvec[r] = t;
    }
    __syncthreads();
  }

  for (int k = N - 1; k >= 0; --k) {
    const double* kn = knot_ptr(a, b, k);
    double* g = gain_ptr(a, b, k);
    const int m = (int)kn[L.oMISC + MISC_M], c = (int)kn[L.oMISC + MISC_NC], nzk = n + m;
    const double* le = a.lams_e + ((size_t)b * (N + 1) + k + 1) * n;
    // ---- 1. y = Ebar x' ----
    if (tid == 0) {
      if (ff) {
        double Eb[36];
        for (int i = 0; i < 36; ++i) Eb[i] = -kn[L.oE6 + i];
        inv6_serial(Eb, g + L.oT6);
      } else {
        for (int i = 0; i < 36; ++i) g[L.oT6 + i] = (i % 7 == 0) ? 1.0 : 0.0;
      }
      int ca = 0;
      for (int i = 0; i < c; ++i) if (kn[L.oACT + i] != 0.0) act_idx[ca++] = i;
      iflag[1] = ca;
    }
    __syncthreads();
    const int ca = iflag[1];
    const double* T6 = g + L.oT6;
    if (ff) {
      // columns < 6: tmp = P' T (stored in GP-free scratch: use w/gh region? need n x 6) -> use LP as scratch (dead here)
      double* tmp = LP;  // n x 6
      for (int idx = tid; idx < n * 6; idx += nthr) {
        const int i = idx / 6, j = idx % 6;
        double s = 0;
        for (int l = 0; l < 6; ++l) s += PT[i * np + l] * T6[l * 6 + j];
        tmp[idx] = s;
      }
      __syncthreads();
      for (int idx = tid; idx < n * 6; idx += nthr) PT[(idx / 6) * np + idx % 6] = tmp[idx];
      __syncthreads();
      // rows < 6: T^T (P' T)
      for (int idx = tid; idx < 6 * n; idx += nthr) {
        const int i = idx / n, j = idx % n;
        double s = 0;
        for (int l = 0; l < 6; ++l) s += T6[l * 6 + i] * PT[l * np + j];
        tmp[idx] = s;
      }
      if (tid < 6) { double s = 0; for (int l = 0; l < 6; ++l) s += T6[l * 6 + tid] * pvec[l]; ph[tid] = s; }
      __syncthreads();
      for (int idx = tid; idx < 6 * n; idx += nthr) PT[(idx / n) * np + idx % n] = tmp[idx];
      for (int i = 6 + tid; i < n; i += nthr) ph[i] = pvec[i];
    } else {
      for (int i = tid; i < n; i += nthr) ph[i] = pvec[i];
    }
    for (int i = tid; i < n; i += nthr) ft[i] = kn[L.oF + i] + mud * le[i];
    __syncthreads();
    // ---- 2. LP = I + mud sym(Ph) ; vv = Ph ft + ph ----
    for (int idx = tid; idx < np * np; idx += nthr) {
      const int i = idx / np, j = idx % np;
      LP[i * ldl + j] = mud * 0.5 * (PT[idx] + PT[j * np + i]) + (i == j ? 1.0 : 0.0);
    }
    for (int i = tid; i < n; i += nthr) { double s = ph[i]; for (int j = 0; j < n; ++j) s += PT[i * np + j] * ft[j]; vv[i] = s; }
    __syncthreads();
    if (!chol_block(LP, n, ldl, tid, nthr, iflag)) { if (tid == 0) a.inst[b].done = 2; return; }
    // inverses of the 16x16 diagonal blocks (pad rows/cols are identity)
    for (int t = tid; t < nb * 16; t += nthr) {
      const int bi = t / 16, cc = t % 16;
      const double* D = LP + (bi * 16) * ldl + bi * 16;
      double* X = LI + bi * 272;
      double xcol[16];
      for (int r = 0; r < 16; ++r) {
        double s = (r == cc) ? 1.0 : 0.0;
        for (int q = cc; q < r; ++q) s -= D[r * ldl + q] * xcol[q];
        xcol[r] = (r < cc) ? 0.0 : s / D[r * ldl + r];
      }
      for (int r = 0; r < 16; ++r) X[r * 17 + cc] = xcol[r];
    }
    __syncthreads();
    // ---- 3. PT <- (L L^T)^-1 PT, column block per wave ----
    for (int bi = 0; bi < nb; ++bi) {
      for (int cj0 = 0; cj0 < nb; cj0 += nw) {
        const int cj = cj0 + wv;
        d4_t acc;
        if (cj < nb) {
          acc = tile_load(PT + (bi * 16) * np + cj * 16, np, lane);
          mma_tile<true>(acc, LP + (bi * 16) * ldl, ldl, 1, PT + cj * 16, np, 1, bi * 16, lane);
        }
        __syncthreads();
        if (cj < nb) tile_store(PT + (bi * 16) * np + cj * 16, np, acc, lane);
        __syncthreads();
        if (cj < nb) {
          acc = d4_t{0, 0, 0, 0};
          mma_tile<false>(acc, LI + bi * 272, 17, 1, PT + (bi * 16) * np + cj * 16, np, 1, 16, lane);
        }
        __syncthreads();
        if (cj < nb) tile_store(PT + (bi * 16) * np + cj * 16, np, acc, lane);
        __syncthreads();
      }
    }
    for (int bi = nb - 1; bi >= 0; --bi) {
      for (int cj0 = 0; cj0 < nb; cj0 += nw) {
        const int cj = cj0 + wv;
        d4_t acc;
        if (cj < nb) {
          acc = tile_load(PT + (bi * 16) * np + cj * 16, np, lane);
          // - sum_{bj > bi} L[bj][bi]^T X[bj][cj]
          mma_tile<true>(acc, LP + ((bi + 1) * 16) * ldl + bi * 16, 1, ldl, PT + ((bi + 1) * 16) * np + cj * 16, np, 1, (nb - 1 - bi) * 16, lane);
        }
        __syncthreads();
        if (cj < nb) tile_store(PT + (bi * 16) * np + cj * 16, np, acc, lane);
        __syncthreads();
        if (cj < nb) {
          acc = d4_t{0, 0, 0, 0};
          mma_tile<false>(acc, LI + bi * 272, 1, 17, PT + (bi * 16) * np + cj * 16, np, 1, 16, lane);
        }
        __syncthreads();
        if (cj < nb) tile_store(PT + (bi * 16) * np + cj * 16, np, acc, lane);
        __syncthreads();
      }
    }
    // symmetrise Pt, w = vv - mud Pt vv
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int i = idx / n, j = idx % n;
      if (j > i) { const double s = 0.5 * (PT[i * np + j] + PT[j * np + i]); PT[i * np + j] = s; PT[j * np + i] = s; }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nthr) { double s = 0; for (int j = 0; j < n; ++j) s += PT[i * np + j] * vv[j]; w[i] = vv[i] - mud * s; }
    for (int idx = tid; idx < n * n; idx += nthr) g[L.oMx + idx] = PT[(idx / n) * np + idx % n];
    for (int i = tid; i < n; i += nthr) g[L.omx + i] = ft[i] - mud * ph[i];
    // ---- 4. AB into LDS (zero padded; u-columns start at np) ----
    for (int idx = tid; idx < np * nzp; idx += nthr) {
      const int i = idx / nzp, zp = idx % nzp;
      double v = 0.0;
      if (i < n) {
        if (zp < n) v = kn[L.oAB + i * nz + zp];
        else if (zp >= np && zp - np < m) v = kn[L.oAB + i * nz + n + (zp - np)];
      }
      AB[idx] = v;
    }
    __syncthreads();
    for (int zp = tid; zp < nzp; zp += nthr) {
      const int z = (zp < n) ? zp : ((zp >= np && zp - np < m) ? n + zp - np : -1);
      if (z >= 0) { double s = kn[L.oG + z]; for (int i = 0; i < n; ++i) s += AB[i * nzp + zp] * w[i]; gh[z] = s; }
    }
    // ---- 5. panels: G_j = Pt AB_j ; Hh[:, j] = H[:, j] + AB^T G_j ----
    for (int cj = 0; cj < nzt; ++cj) {
      for (int ri0 = 0; ri0 < nb; ri0 += nw) {
        const int ri = ri0 + wv;
        if (ri < nb) {
          d4_t acc = d4_t{0, 0, 0, 0};
          mma_tile<false>(acc, PT + ri * 16, 1, np, AB + cj * 16, nzp, 1, np, lane);  // Pt symmetric: A(i,k) = Pt[k][i]
          tile_store(GP + (ri * 16) * 16, 16, acc, lane);
        }
      }
      __syncthreads();
      for (int zi0 = 0; zi0 < nzt; zi0 += nw) {
        const int zi = zi0 + wv;
        if (zi < nzt) {
          d4_t acc = d4_t{0, 0, 0, 0};
          mma_tile<false>(acc, AB + zi * 16, 1, nzp, GP, 16, 1, np, lane);
          // add H and write the valid entries to the scratch (leading dimension nz, compact z indexing)
          const int col_p = cj * 16 + (lane & 15);
          const int zc = (col_p < n) ? col_p : ((col_p >= np && col_p - np < m) ? n + col_p - np : -1);
          for (int q = 0; q < 4; ++q) {
            const int row_p = zi * 16 + (lane >> 4) + 4 * q;
            const int zr = (row_p < n) ? row_p : ((row_p >= np && row_p - np < m) ? n + row_p - np : -1);
            if (zr >= 0 && zc >= 0) Hh[zr * nz + zc] = kn[L.oH + zr * nz + zc] + acc[q];
          }
        }
      }
      __syncthreads();
    }
    // ---- 6. stage KKT (AB is dead: R1 is reused) ----
    const bool small_ca = ca <= S.cap;
    double* Ct = small_ca ? CTl : (wk + L.wCt);   // ca x ldc
    double* V = small_ca ? VXl : (wk + L.wV);     // ca x ldv
    double* Y = small_ca ? Yl : (wk + L.wY);      // m x ca
    double* Sc = small_ca ? SCl : (wk + L.wSc);   // ca x ca
    const int ldc = small_ca ? np : nz, ldv = small_ca ? np : nr;
    if (small_ca) for (int idx = tid; idx < S.cap * np; idx += nthr) { CTl[idx] = 0.0; VXl[idx] = 0.0; }
    for (int idx = tid; idx < mp * np; idx += nthr) ST[idx] = 0.0;
    for (int idx = tid; idx < mp * lw; idx += nthr) W[idx] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += nthr) {
      const int i = idx / m, j = idx % m;
      Lr[idx] = 0.5 * (Hh[(n + i) * nz + n + j] + Hh[(n + j) * nz + n + i]);
    }
    for (int idx = tid; idx < m * nr; idx += nthr) {
      const int i = idx / nr, z = idx % nr;
      const double sv = (z < n) ? Hh[(n + i) * nz + z] : gh[n + i];
      W[i * lw + ((z < n) ? z : np)] = -sv;
      if (z < n) ST[i * np + z] = sv;
    }
    for (int idx = tid; idx < ca * n; idx += nthr) Ct[(idx / n) * ldc + idx % n] = kn[L.oCD + act_idx[idx / n] * nz + idx % n];
    for (int idx = tid; idx < m * ca; idx += nthr) Y[(idx / ca) * ca + idx % ca] = kn[L.oCD + act_idx[idx % ca] * nz + n + idx / ca];
    for (int i = tid; i < ca; i += nthr) dtl[i] = kn[L.oDT + act_idx[i]];
    __syncthreads();
    if (!chol_block(Lr, m, m, tid, nthr, iflag)) { if (tid == 0) a.inst[b].done = 3; return; }
    trsm_lower_block(Lr, m, m, W, lw, lw, tid, nthr);
    // V holds [Kv | kv]: columns 0..n-1 and a separate vector for the constant column
    double* kvc = dtl + L.c;  // ca
    if (ca > 0) {
      trsm_lower_block(Lr, m, m, Y, ca, ca, tid, nthr);
      for (int idx = tid; idx < ca * ca; idx += nthr) {
        const int i = idx / ca, j = idx % ca;
        double s = (i == j) ? mu : 0.0;
        for (int l = 0; l < m; ++l) s += Y[l * ca + i] * Y[l * ca + j];
        Sc[idx] = s;
      }
      for (int idx = tid; idx < ca * nr; idx += nthr) {
        const int i = idx / nr, z = idx % nr;
        double s = (z < n) ? Ct[i * ldc + z] : dtl[i];
        for (int l = 0; l < m; ++l) s += Y[l * ca + i] * W[l * lw + ((z < n) ? z : np)];
        if (z < n) V[i * ldv + z] = s; else kvc[i] = s;
      }
      __syncthreads();
      if (!chol_block(Sc, ca, ca, tid, nthr, iflag)) { if (tid == 0) a.inst[b].done = 4; return; }
      potrs_block(Sc, ca, ca, V, n, ldv, tid, nthr);
      potrs_block(Sc, ca, ca, kvc, 1, 1, tid, nthr);
      for (int idx = tid; idx < m * nr; idx += nthr) {
        const int l = idx / nr, z = idx % nr;
        double s = 0;
        for (int i = 0; i < ca; ++i) s += Y[l * ca + i] * ((z < n) ? V[i * ldv + z] : kvc[i]);
        W[l * lw + ((z < n) ? z : np)] -= s;
      }
      __syncthreads();
    }
    trsm_lower_t_block(Lr, m, m, W, lw, lw, tid, nthr);
    for (int idx = tid; idx < m * n; idx += nthr) g[L.oK + idx] = W[(idx / n) * lw + idx % n];
    for (int i = tid; i < m; i += nthr) g[L.ok + i] = W[i * lw + np];
    for (int idx = tid; idx < c * n; idx += nthr) g[L.oKnu + idx] = 0.0;
    for (int i = tid; i < c; i += nthr) g[L.oknu + i] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < ca * n; idx += nthr) g[L.oKnu + act_idx[idx / n] * n + idx % n] = V[(idx / n) * ldv + idx % n];
    for (int i = tid; i < ca; i += nthr) g[L.oknu + act_idx[i]] = kvc[i];
    // p = qh + Sh k + Ca^T kv
    for (int r = tid; r < n; r += nthr) {
      double t = gh[r];
      for (int i = 0; i < m; ++i) t += ST[i * np + r] * W[i * lw + np];
      for (int i = 0; i < ca; ++i) t += Ct[i * ldc + r] * kvc[i];
      g[L.op + r] = t;
      pvec[r] = t;
    }
    __syncthreads();
    // ---- 7. P = Qh + Sh K + Ca^T Kv  ->  PT (next knot's P') ----
    if (small_ca) {
      const int kc = (ca + 3) & ~3;
      for (int t = wv; t < nb * nb; t += nw) {
        const int ri = t / nb, cj = t % nb;
        d4_t acc = d4_t{0, 0, 0, 0};
        mma_tile<false>(acc, ST + ri * 16, 1, np, W + cj * 16, lw, 1, mp, lane);
        if (kc > 0) mma_tile<false>(acc, CTl + ri * 16, 1, np, VXl + cj * 16, np, 1, kc, lane);
        const int col = cj * 16 + (lane & 15);
        for (int q = 0; q < 4; ++q) {
          const int row = ri * 16 + (lane >> 4) + 4 * q;
          PT[row * np + col] = (row < n && col < n) ? Hh[row * nz + col] + acc[q] : 0.0;
        }
      }
    } else {
      for (int idx = tid; idx < np * np; idx += nthr) {
        const int r = idx / np, s = idx % np;
        double t = 0.0;
        if (r < n && s < n) {
          t = Hh[r * nz + s];
          for (int i = 0; i < m; ++i) t += ST[i * np + r] * W[i * lw + s];
          for (int i = 0; i < ca; ++i) t += Ct[i * ldc + r] * V[i * ldv + s];
        }
        PT[idx] = t;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int i = idx / n, j = idx % n;
      if (j >= i) { const double s = 0.5 * (PT[i * np + j] + PT[j * np + i]); g[L.oP + i * n + j] = s; g[L.oP + j * n + i] = s; }
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthr) PT[(idx / n) * np + idx % n] = g[L.oP + idx];
    __syncthreads();
  }
}
