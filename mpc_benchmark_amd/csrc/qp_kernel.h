// qp_kernel.h — batched dense QP on the GPU ("next" row N3; replaces proxsuite.proxqp.dense.QP of QP_utils.py:437-575,
// 584-762): one workgroup per QP, the Newton matrices in LDS.  Same algorithm as oracle/qp.hpp (include/mpc_qp_abi.h):
// proximal augmented Lagrangian with a bound-constrained-Lagrangian outer loop, semismooth Newton on the active rows with an
// exact line search, Cholesky of the primal block P = H + rho I + C_I^T C_I / mu_in and of the equality Schur complement
// S = mu_eq I + A P^-1 A^T.  Two forms of the linear algebra of a Newton step: MF (problems whose padded matrices fit the LDS: the
// reference's n = 62, neq = 40) — blocked Cholesky with pre-inverted 16 x 16 diagonal blocks, blocked triangular solves and the
// Schur complement as tile products on the fp64 matrix cores (mfma_blocks.h, the building blocks of the Riccati sweep) — and the
// first version's column-by-column VALU loops (two workgroup barriers per column) for everything larger.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mpc_qp_abi.h"
#include "solver_args.h"  // wave_sum, wave_max_nonneg
#include "mfma_blocks.h"    // blocked Cholesky / triangular solves on the matrix cores

#define QP_THREADS 256

struct QpLds {
  int P, Y, S, vec, H, A, C, mats, total_bytes;  // mats: 1 = H, A, C are staged in LDS too, 2 = A and C only
  // MF form: padded dimensions (np, ep: multiples of 16 ; ncb: 16-column blocks of [A^T | r1]), leading dimensions, inverted diagonal blocks
  int mf, np, ep, ncb, ldp, ldy, lds, LIp, LIs, ZD;
};
static inline QpLds make_qp_lds(int n, int neq, int nin, int m, bool want_mats = true, bool allow_mf = true) {
  QpLds s;
  auto layout = [&](bool mf, int mats) {
    int o = 0;
    auto take = [&](int c) { int r = o; o += (c + 1) & ~1; return r; };
    s.mf = mf ? 1 : 0;
    s.np = (n + 15) & ~15; s.ep = (neq + 15) & ~15; s.ncb = (neq + 1 + 15) / 16;
    if (mf) {
      // P and the Gram matrix G = [Y | w]^T [Y | w] (its leading block becomes S) as the tiles of their lower block triangles
      // (mfma_blocks.h ptile: 272 doubles each, the inverse of a diagonal factor block replaces the block)
      const int nbp = s.np / 16;
      s.ldp = s.lds = 17; s.ldy = 16 * s.ncb + 1; s.LIp = s.LIs = 0;
      s.P = take(nbp * (nbp + 1) / 2 * 272); s.Y = take(s.np * s.ldy);
      s.S = take(s.ncb * (s.ncb + 1) / 2 * 272); s.ZD = take((s.np > s.ep ? s.np : s.ep) * 17);
    } else {
      s.ldp = n + 1; s.ldy = neq + 1; s.lds = neq + 1; s.LIp = s.LIs = s.ZD = 0;
      s.P = take(n * (n + 1)); s.Y = take(n * (neq + 1)); s.S = take(neq * (neq + 1));
    }
    s.vec = take(11 * n + 6 * neq + 4 * m + nin + 64);
    s.H = s.A = s.C = 0;
    if (mats == 1) s.H = take(n * n);
    if (mats >= 1) { s.A = take(neq * n); s.C = take(nin * n); }
    s.mats = mats;
    s.total_bytes = o * 8;
    return o * 8 + 64 <= 160 * 1024;
  };
  // preference: matrix-core form with all / some / none of the problem matrices in LDS, then the column-by-column form
  const bool can_mf = allow_mf && neq > 0 && n >= 16;
  if (can_mf && want_mats && layout(true, 1)) return s;
  if (can_mf && want_mats && layout(true, 2)) return s;
  if (can_mf && layout(true, 0)) return s;
  if (want_mats && layout(false, 1)) return s;
  layout(false, 0);
  return s;
}

struct QpArgs {
  mpc_qp_dims d;
  mpc_qp_settings S;
  const double *H, *g, *A, *b, *C, *l, *u, *lb, *ub;
  double *x, *y, *z;  // [B][n], [B][neq], [B][m]  (m = nin + n when box): start point in, solution out
  mpc_qp_info* info;
  QpLds lds;
};

DEV double qp_zplus(double zk, double s, double lo, double hi, double mu) {
  const double tu = zk + (s - hi) / mu, tl = zk + (s - lo) / mu;
  return tu > 0.0 ? tu : (tl < 0.0 ? tl : 0.0);
}

// block-wide sum / max of one value per thread (all threads get the result); red: 2 * 8 doubles of LDS
DEV double qp_block_sum(double v, double* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = 0;
  for (int i = 0; i < (QP_THREADS >> 6); ++i) r += red[i];
  return r;
}
DEV double qp_block_max(double v, double* red, int tid) {
  v = wave_max_nonneg(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = 0;
  for (int i = 0; i < (QP_THREADS >> 6); ++i) r = fmax(r, red[i]);
  return r;
}

// v <- L^-1 v / v <- L^-T v by ONE wavefront (lanes share each row's dot product); L: lower triangle, leading dimension ld
DEV void qp_fwd_wave(const double* L, int n, int ld, double* v, int lane) {
  for (int i = 0; i < n; ++i) {
    double t = 0;
    for (int k = lane; k < i; k += 64) t += L[i * ld + k] * v[k];
    t = wave_sum(t);
    if (lane == 0) v[i] = (v[i] - t) / L[i * ld + i];
  }
}
DEV void qp_bwd_wave(const double* L, int n, int ld, double* v, int lane) {
  for (int i = n - 1; i >= 0; --i) {
    double t = 0;
    for (int k = i + 1 + lane; k < n; k += 64) t += L[k * ld + i] * v[k];
    t = wave_sum(t);
    if (lane == 0) v[i] = (v[i] - t) / L[i * ld + i];
  }
}

// in-place Cholesky of the lower triangle of M (n x n, leading dimension ld) in LDS; returns false if not positive definite
DEV bool qp_chol(double* M, int n, int ld, int* flag, int tid) {
  (void)flag;
  for (int j = 0; j < n; ++j) {
    const double d = M[j * ld + j];  // final after the previous column's update (barrier below)
    if (!(d > 0.0)) return false;    // uniform: every thread reads the same value
    const double dj = sqrt(d);
    __syncthreads();                 // everyone has read the pivot before it is overwritten
    if (tid == 0) M[j * ld + j] = dj;
    for (int i = j + 1 + tid; i < n; i += QP_THREADS) M[i * ld + j] /= dj;
    __syncthreads();
    // trailing update of the lower triangle: M[i][k] -= L[i][j] L[k][j], j < k <= i  (one row per thread slice)
    for (int i = j + 1 + (tid >> 4); i < n; i += QP_THREADS >> 4) {
      const double lij = M[i * ld + j];
      for (int k = j + 1 + (tid & 15); k <= i; k += 16) M[i * ld + k] -= lij * M[k * ld + j];
    }
    __syncthreads();
  }
  return true;
}

// MATS: 0 = H, A, C read from global memory, 1 = all three staged in LDS, 2 = A and C only ; MF: matrix-core linear algebra
template <int MATS, bool MF>
__global__ void __launch_bounds__(QP_THREADS) k_qp_solve(QpArgs a) {
  const int bi = blockIdx.x, tid = threadIdx.x, nthr = QP_THREADS;
  const int n = a.d.n, neq = a.d.neq, nin = a.d.nin, box = a.d.box, m = nin + (box ? n : 0);
  const mpc_qp_settings& S = a.S;
  const double* Hg = a.H + (size_t)bi * n * n; const double* g = a.g + (size_t)bi * n;
  const double* Ag = a.A + (size_t)bi * neq * n; const double* b = a.b + (size_t)bi * neq;
  const double* Cg = a.C + (size_t)bi * nin * n; const double* l = a.l + (size_t)bi * nin; const double* u = a.u + (size_t)bi * nin;
  const double* lb = box ? a.lb + (size_t)bi * n : nullptr; const double* ub = box ? a.ub + (size_t)bi * n : nullptr;
  double* xg = a.x + (size_t)bi * n; double* yg = a.y + (size_t)bi * neq; double* zg = a.z + (size_t)bi * m;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int ldp = a.lds.ldp, lds_ = a.lds.lds, ldy = a.lds.ldy;
  double *Pm = sm + a.lds.P, *Y = sm + a.lds.Y, *Sm = sm + a.lds.S, *v = sm + a.lds.vec;
  const int np = a.lds.np, ep = a.lds.ep, ncb = a.lds.ncb, lane = tid & 63, wv = tid >> 6, nw = QP_THREADS >> 6;
  double* ZD = sm + a.lds.ZD;
  // H, A, C: LDS copies when they fit (MATS; every mat-vec below then runs on LDS), else the global arrays (H through its
  // symmetric image so that neighbouring threads read neighbouring addresses)
  const double* H = MATS == 1 ? sm + a.lds.H : Hg;
  const double* A = MATS ? sm + a.lds.A : Ag;
  const double* C = MATS ? sm + a.lds.C : Cg;
  if (MATS == 1) for (int i = tid; i < n * n; i += nthr) sm[a.lds.H + i] = Hg[i];
  if (MATS) {
    for (int i = tid; i < neq * n; i += nthr) sm[a.lds.A + i] = Ag[i];
    for (int i = tid; i < nin * n; i += nthr) sm[a.lds.C + i] = Cg[i];
  }
  double *x = v, *xk = x + n, *grad = xk + n, *r1 = grad + n, *dx = r1 + n, *w = dx + n, *hx = w + n, *hd = hx + n, *tmpn = hd + n;
  double *y = tmpn + n, *ye = y + neq, *yplus = ye + neq, *Ax = yplus + neq, *Ad = Ax + neq, *tmpe = Ad + neq;
  double *z = tmpe + neq, *zp = z + m, *s = zp + m, *ds = s + m;
  double* red = ds + m;  // 16
  double* hx0 = red + 16;   // H x, kept up to date (H x += alpha H dx): H is read for ONE mat-vec per Newton step (global memory unless MATS == 1)
  double* hdx = hx0 + n;    // H dx of the current step
  int* actl = (int*)(hdx + n);  // active inequality rows of the current Newton step (nin ints)
  __shared__ int flag;
  auto lo = [&](int r) { return r < nin ? l[r] : lb[r - nin]; };
  auto hi = [&](int r) { return r < nin ? u[r] : ub[r - nin]; };
  auto row_dot = [&](int r, const double* xx) {
    if (r >= nin) return xx[r - nin];
    double t = 0;
#pragma unroll 8
      for (int j = 0; j < n; ++j) t += C[r * n + j] * xx[j]; return t;
  };
  auto h_times = [&](const double* vv, double* out) {  // out = H vv  (n <= 128)
    const int c0 = lane < n ? lane : 0, c1 = lane + 64 < n ? lane + 64 : 0;
    const double v0 = lane < n ? vv[c0] : 0.0, v1 = lane + 64 < n ? vv[c1] : 0.0;
    for (int r0 = 4 * wv; r0 < n; r0 += 4 * nw) {
      double hv[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int r = r0 + q < n ? r0 + q : 0; hv[q][0] = H[r * n + c0]; hv[q][1] = H[r * n + c1]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const double sres = wave_sum(hv[q][0] * v0 + hv[q][1] * v1); if (lane == 0 && r0 + q < n) out[r0 + q] = sres; }
    }
  };
  for (int j = tid; j < n; j += nthr) x[j] = xg[j];
  for (int i = tid; i < neq; i += nthr) y[i] = yg[i];
  for (int r = tid; r < m; r += nthr) z[r] = zg[r];
  __syncthreads();
  h_times(x, hx0);
  __syncthreads();
  double mu_eq = S.mu_eq, mu_in = S.mu_in;
  double prim_tol = pow(0.1, S.alpha_bcl), inner_tol = 1.0;
  int status = 1, iters = 0, iters_in = 0;
  double rp_out = 0, rd_out = 0;
  // s, zp, Ax, ye at x (multiplier estimates around the current y, z)
  auto eval = [&]() {
    for (int r = tid; r < m; r += nthr) { const double sr = row_dot(r, x); s[r] = sr; zp[r] = qp_zplus(z[r], sr, lo(r), hi(r), mu_in); }
    for (int i = tid; i < neq; i += nthr) { double t = -b[i];
#pragma unroll 8
      for (int j = 0; j < n; ++j) t += A[i * n + j] * x[j]; Ax[i] = t; ye[i] = y[i] + t / mu_eq; }
    __syncthreads();
  };
  for (int outer = 0; outer <= S.max_iter; ++outer) {
    // ---- residuals at (x, y, z) ----
    double rp = 0, rd = 0;
    for (int i = tid; i < neq; i += nthr) { double t = -b[i];
#pragma unroll 8
      for (int j = 0; j < n; ++j) t += A[i * n + j] * x[j]; rp = fmax(rp, fabs(t)); }
    for (int r = tid; r < m; r += nthr) { const double sr = row_dot(r, x); rp = fmax(rp, fmax(sr - hi(r), lo(r) - sr)); }
    for (int j = tid; j < n; j += nthr) {
      double t = g[j] + hx0[j];
      #pragma unroll 8
      for (int i = 0; i < neq; ++i) t += A[i * n + j] * y[i];
      #pragma unroll 8
      for (int r = 0; r < nin; ++r) t += C[r * n + j] * z[r];
      if (box) t += z[nin + j];
      rd = fmax(rd, fabs(t));
    }
    rp = qp_block_max(fmax(rp, 0.0), red, tid);
    rd = qp_block_max(rd, red, tid);
    rp_out = rp; rd_out = rd;
    if (fmax(rp, rd) <= S.eps_abs) { status = 0; break; }
    if (outer == S.max_iter) break;
    iters = outer + 1;
    for (int j = tid; j < n; j += nthr) xk[j] = x[j];
    __syncthreads();
    for (int it = 0; it < S.max_iter_in; ++it) {
      eval();
      double gn = 0;
      for (int j = tid; j < n; j += nthr) {
        double t = g[j] + S.rho * (x[j] - xk[j]) + hx0[j];
        hx[j] = t;  // H x + g + rho (x - xk): reused by the line search
        #pragma unroll 8
        for (int r = 0; r < nin; ++r) t += C[r * n + j] * zp[r];
        if (box) t += zp[nin + j];
        r1[j] = -t;
        #pragma unroll 8
        for (int i = 0; i < neq; ++i) t += A[i * n + j] * ye[i];
        grad[j] = t;
        gn = fmax(gn, fabs(t));
      }
      gn = qp_block_max(gn, red, tid);
      if (gn <= inner_tol) break;
      iters_in += 1;
      if constexpr (MF) {
        // ---- matrix-core form.  P (np x np, symmetric, identity padding) ; Y = [A^T | r1 | 0] (np x 16 ncb, zero padding) ----
        if (tid == 0) { int na_ = 0; for (int r = 0; r < nin; ++r) if (zp[r] != 0.0) actl[na_++] = r; flag = na_; }  // active inequality rows (few)
        __syncthreads();
        const int nact = flag;
        const int nbp = np >> 4, nbs = ep >> 4;
        for (int idx = tid; idx < nbp * (nbp + 1) / 2 * 256; idx += nthr) {  // lower block triangle, diagonal tiles in full
          const int tl = idx >> 8, e = idx & 255;
          int bi_ = 0;
          while ((bi_ + 1) * (bi_ + 2) / 2 <= tl) ++bi_;
          const int j = bi_ * 16 + (e >> 4), k = (tl - bi_ * (bi_ + 1) / 2) * 16 + (e & 15);
          double t = (j == k) ? 1.0 : 0.0;
          if (j < n && k < n) {
            t = H[j * n + k] + (j == k ? S.rho : 0.0);
            for (int q = 0; q < nact; ++q) { const int r = actl[q]; t += C[r * n + j] * C[r * n + k] / mu_in; }
            if (box && j == k && zp[nin + j] != 0.0) t += 1.0 / mu_in;
          }
          Pm[tl * 272 + (e >> 4) * 17 + (e & 15)] = t;
        }
        for (int idx = tid; idx < 16 * ncb * np; idx += nthr) {  // j fastest: A is read along its rows
          const int i = idx / np, j = idx % np;
          Y[j * ldy + i] = (j < n) ? ((i < neq) ? A[i * n + j] : (i == neq ? r1[j] : 0.0)) : 0.0;
        }
        __syncthreads();
        if (!chol_tiles(Pm, nbp, tid, &flag)) { status = 2; goto done; }
        trsm_fwd_tiles(Pm, nbp, Y, ldy, ncb, wv, nw, lane);   // [Y | w] <- L^-1 [A^T | r1]
        __syncthreads();
        for (int j = tid; j < n; j += nthr) w[j] = Y[j * ldy + neq];
        // G = [Y | w]^T [Y | w] (lower block triangle): S = mu_eq I + G[:neq, :neq], Y^T w = G[neq, :neq]
        {
          const int nst = ncb * (ncb + 1) / 2;
          for (int t = wv; t < nst; t += nw) {
            int ri = 0, rem = t;
            while (rem > ri) { rem -= ri + 1; ++ri; }
            d4_t acc = d4_t{0, 0, 0, 0};
            mma_tile<false>(acc, Y + ri * 16, 1, ldy, Y + rem * 16, ldy, 1, np, lane);
            tile_store(ptile(Sm, ri, rem), 17, acc, lane);
          }
        }
        __syncthreads();
        for (int i = tid; i < neq; i += nthr) yplus[i] = ctile(Sm, neq >> 4, i >> 4)[(neq & 15) * 17 + (i & 15)] + Ax[i] + mu_eq * y[i];  // row neq of G
        __syncthreads();
        for (int idx = tid; idx < nbs * (nbs + 1) / 2 * 256; idx += nthr) {  // S proper: mu_eq on the diagonal, identity padding
          const int tl = idx >> 8, e = idx & 255;
          int bi_ = 0;
          while ((bi_ + 1) * (bi_ + 2) / 2 <= tl) ++bi_;
          const int i = bi_ * 16 + (e >> 4), k = (tl - bi_ * (bi_ + 1) / 2) * 16 + (e & 15);
          double* el = Sm + tl * 272 + (e >> 4) * 17 + (e & 15);
          *el = (i < neq && k < neq) ? *el + (i == k ? mu_eq : 0.0) : (i == k ? 1.0 : 0.0);
        }
        for (int idx = tid; idx < ep * 16; idx += nthr) { const int i = idx >> 4, c = idx & 15; ZD[i * 17 + c] = (c == 0 && i < neq) ? yplus[i] : 0.0; }
        __syncthreads();
        if (!chol_tiles(Sm, nbs, tid, &flag)) { status = 2; goto done; }
        trsm_fwd_tiles(Sm, nbs, ZD, 17, 1, wv, nw, lane);
        trsm_bwd_tiles(Sm, nbs, ZD, 17, 1, wv, nw, lane);   // (one column block: wavefront 0, its LDS operations in order)
        __syncthreads();
        for (int i = tid; i < neq; i += nthr) yplus[i] = ZD[i * 17];
        __syncthreads();
        for (int idx = tid; idx < np * 16; idx += nthr) {
          const int j = idx >> 4, c = idx & 15;
          double t = 0.0;
          if (c == 0 && j < n) { t = w[j];
#pragma unroll 8
            for (int i = 0; i < neq; ++i) t -= Y[j * ldy + i] * yplus[i]; }
          ZD[j * 17 + c] = t;
        }
        __syncthreads();
        trsm_bwd_tiles(Pm, nbp, ZD, 17, 1, wv, nw, lane);    // dx = L^-T (w - Y yplus)
        __syncthreads();
        for (int j = tid; j < n; j += nthr) dx[j] = ZD[j * 17];
        __syncthreads();
      } else {
      // ---- primal block P = H + rho I + active rows / mu_in (lower triangle), Y = A^T ----
      for (int idx = tid; idx < n * n; idx += nthr) {
        const int j = idx / n, k = idx % n;
        if (k <= j) {
          double t = H[j * n + k] + (j == k ? S.rho : 0.0);
          for (int r = 0; r < nin; ++r) if (zp[r] != 0.0) t += C[r * n + j] * C[r * n + k] / mu_in;
          if (box && j == k && zp[nin + j] != 0.0) t += 1.0 / mu_in;
          Pm[j * ldp + k] = t;
        }
      }
      for (int idx = tid; idx < n * ldy; idx += nthr) { const int j = idx / ldy, i = idx % ldy; Y[j * ldy + i] = (i < neq) ? A[i * n + j] : r1[j]; }
      __syncthreads();
      if (!qp_chol(Pm, n, ldp, &flag, tid)) { status = 2; goto done; }
      // [Y | w] <- L^-1 [A^T | r1]: right-looking substitution, all neq + 1 columns together
      for (int i = 0; i < n; ++i) {
        const double dinv = 1.0 / Pm[i * ldp + i];
        if (tid < ldy) Y[i * ldy + tid] *= dinv;
        __syncthreads();
        for (int idx = tid; idx < (n - i - 1) * ldy; idx += nthr) {
          const int r = i + 1 + idx / ldy, c = idx % ldy;
          Y[r * ldy + c] -= Pm[r * ldp + i] * Y[i * ldy + c];
        }
        __syncthreads();
      }
      for (int j = tid; j < n; j += nthr) w[j] = Y[j * ldy + neq];
      __syncthreads();
      // S = mu_eq I + Y^T Y (lower), rhs = Y^T w + (A x - b) + mu_eq y
      for (int idx = tid; idx < neq * neq; idx += nthr) {
        const int i = idx / neq, k = idx % neq;
        if (k <= i) { double t = (i == k) ? mu_eq : 0.0;
#pragma unroll 8
          for (int j = 0; j < n; ++j) t += Y[j * ldy + i] * Y[j * ldy + k]; Sm[i * lds_ + k] = t; }
      }
      for (int i = tid; i < neq; i += nthr) { double t = Ax[i] + mu_eq * y[i];
#pragma unroll 8
        for (int j = 0; j < n; ++j) t += Y[j * ldy + i] * w[j]; yplus[i] = t; }
      __syncthreads();
      if (!qp_chol(Sm, neq, lds_, &flag, tid)) { status = 2; goto done; }
      if (tid < 64) { qp_fwd_wave(Sm, neq, lds_, yplus, tid); qp_bwd_wave(Sm, neq, lds_, yplus, tid); }
      __syncthreads();
      for (int j = tid; j < n; j += nthr) { double t = w[j];
#pragma unroll 8
        for (int i = 0; i < neq; ++i) t -= Y[j * ldy + i] * yplus[i]; dx[j] = t; }
      __syncthreads();
      if (tid < 64) qp_bwd_wave(Pm, n, ldp, dx, tid);
      __syncthreads();
      }
      // ---- exact line search along dx ----
      for (int r = tid; r < m; r += nthr) ds[r] = row_dot(r, dx);
      h_times(dx, hdx);
      __syncthreads();
      double pa0 = 0, pa1 = 0, pe0 = 0, pe1 = 0;
      for (int j = tid; j < n; j += nthr) { pa0 += dx[j] * hx[j]; pa1 += dx[j] * (S.rho * dx[j] + hdx[j]); }
      for (int i = tid; i < neq; i += nthr) { double t = 0;
#pragma unroll 8
        for (int j = 0; j < n; ++j) t += A[i * n + j] * dx[j]; pe0 += t * ye[i]; pe1 += t * t / mu_eq; }
      const double lin = qp_block_sum(pa0 + pe0, red, tid), quad = qp_block_sum(pa1 + pe1, red, tid);
      // root of the increasing piecewise-linear phi' by ONE wavefront (safeguarded Newton; no workgroup barrier per trial)
      if (tid < 64) {
        double alpha = 1.0, lo_a = 0.0, hi_a = -1.0;
        for (int ls = 0; ls < 40; ++ls) {
          double pf = 0, pc = 0;
          for (int r = tid; r < m; r += 64) {
            const double zr = qp_zplus(z[r], s[r] + alpha * ds[r], lo(r), hi(r), mu_in);
            if (zr != 0.0) { pf += ds[r] * zr; pc += ds[r] * ds[r] / mu_in; }
          }
          const double f = lin + alpha * quad + wave_sum(pf);
          const double curv = quad + wave_sum(pc);
          if (fabs(f) <= 1e-13 * (fabs(lin) + 1.0)) break;
          if (f < 0) lo_a = alpha; else hi_a = alpha;
          double an = alpha - f / curv;
          if (an <= lo_a || (hi_a > 0 && an >= hi_a)) an = (hi_a > 0) ? 0.5 * (lo_a + hi_a) : 2.0 * alpha;
          if (fabs(an - alpha) <= 1e-15 * alpha) { alpha = an; break; }
          alpha = an;
        }
        if (tid == 0) red[15] = alpha;
      }
      __syncthreads();
      const double alpha = red[15];
      double stepn = 0, xn = 1.0;
      for (int j = tid; j < n; j += nthr) { stepn = fmax(stepn, fabs(alpha * dx[j])); xn = fmax(xn, fabs(x[j])); x[j] += alpha * dx[j]; hx0[j] += alpha * hdx[j]; }
      stepn = qp_block_max(stepn, red, tid);
      xn = qp_block_max(xn, red, tid);
      __syncthreads();
      if (stepn <= 1e-14 * xn) break;  // round-off floor: no further progress
    }
    eval();
    double rp_eq = 0, rp_in = 0;
    for (int i = tid; i < neq; i += nthr) rp_eq = fmax(rp_eq, fabs(Ax[i]));
    for (int r = tid; r < m; r += nthr) rp_in = fmax(rp_in, fmax(s[r] - hi(r), lo(r) - s[r]));
    rp_eq = qp_block_max(rp_eq, red, tid);
    rp_in = qp_block_max(fmax(rp_in, 0.0), red, tid);
    if (fmax(rp_eq, rp_in) <= prim_tol) {
      for (int i = tid; i < neq; i += nthr) y[i] = ye[i];
      for (int r = tid; r < m; r += nthr) z[r] = zp[r];
      prim_tol = fmax(prim_tol * pow(mu_in, S.beta_bcl), S.eps_abs);
      inner_tol = fmax(inner_tol * mu_in, S.eps_abs);
    } else {
      if (rp_eq > prim_tol) mu_eq = fmax(mu_eq * S.mu_update_factor, S.mu_min_eq);
      if (rp_in > prim_tol) mu_in = fmax(mu_in * S.mu_update_factor, S.mu_min_in);
      prim_tol = fmax(pow(mu_in, S.alpha_bcl) * pow(0.1, S.alpha_bcl), S.eps_abs);
      inner_tol = fmax(mu_in, S.eps_abs);
    }
    __syncthreads();
  }
done:
  __syncthreads();
  for (int j = tid; j < n; j += nthr) xg[j] = x[j];
  for (int i = tid; i < neq; i += nthr) yg[i] = y[i];
  for (int r = tid; r < m; r += nthr) zg[r] = z[r];
  int na = 0;
  for (int r = tid; r < m; r += nthr) if (z[r] != 0.0) ++na;
  const double nat = qp_block_sum((double)na, red, tid);
  if (tid == 0) {
    mpc_qp_info& o = a.info[bi];
    o.prim_res = rp_out; o.dual_res = rd_out; o.mu_eq = mu_eq; o.mu_in = mu_in; o.iters = iters; o.iters_in = iters_in; o.status = status;
    o.n_active = (int)nat;
  }
}
