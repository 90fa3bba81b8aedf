// mfma_blocks.h — dense fp64 building blocks on the CDNA4 matrix cores, shared by the Riccati sweep and the
// whole-body stage kernel: 16x16 MFMA tile products on LDS operands (v_mfma_f64_16x16x4_f64), a 16x16 Cholesky +
// inverse held in the registers of one wavefront (readlane broadcasts), blocked Cholesky and blocked triangular
// solves with pre-inverted diagonal blocks (pure MFMA, one column block per wavefront, no workgroup barrier).
#pragma once
#include "device_common.h"

typedef double d4_t __attribute__((ext_vector_type(4)));

// ---- MFMA tile primitives ---------------------------------------------------------------------------------
// acc += sum_{k<K} A(i,k) * B(k,j) for one 16x16 tile; A(i,k) at A[i*a_is + k*a_ks], B(k,j) at B[k*b_ks + j*b_js].
// Fragment layout of v_mfma_f64_16x16x4_f64: lane l supplies A(l&15, l>>4) and B(l>>4, l&15); result register r of
// lane l is C((l>>4) + 4r, l&15).  K must be a multiple of 4; groups of 16 are software-pipelined.
// The k-loop is software-pipelined by hand: the eight operands of the next 16-deep group are requested from LDS
// before the four (dependent) MFMAs of the current group issue, and the scheduler is fenced so that it cannot
// sink those loads back below the MFMAs — with one or two wavefronts per SIMD nothing else hides the LDS latency.
template <bool NEG>
DEV void mma_tile(d4_t& acc, const double* A, int a_is, int a_ks, const double* B, int b_ks, int b_js, int K, int lane) {
  const int i = lane & 15, kk = lane >> 4;
  const double* ap = A + i * a_is + kk * a_ks;
  const double* bp = B + kk * b_ks + i * b_js;
  const int Kmain = K & ~15;
  int k0 = 0;
  if (Kmain > 0) {
    double a0 = ap[0], a1 = ap[4 * a_ks], a2 = ap[8 * a_ks], a3 = ap[12 * a_ks];
    double b0 = bp[0], b1 = bp[4 * b_ks], b2 = bp[8 * b_ks], b3 = bp[12 * b_ks];
    for (; k0 < Kmain; k0 += 16) {
      const int kn = (k0 + 16 < Kmain) ? k0 + 16 : k0;  // the last group re-requests its own operands (unused)
      const double na0 = ap[kn * a_ks], na1 = ap[(kn + 4) * a_ks], na2 = ap[(kn + 8) * a_ks], na3 = ap[(kn + 12) * a_ks];
      const double nb0 = bp[kn * b_ks], nb1 = bp[(kn + 4) * b_ks], nb2 = bp[(kn + 8) * b_ks], nb3 = bp[(kn + 12) * b_ks];
      __builtin_amdgcn_sched_barrier(0);
      if (NEG) { a0 = -a0; a1 = -a1; a2 = -a2; a3 = -a3; }
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a0 = na0; a1 = na1; a2 = na2; a3 = na3; b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
    }
  }
  for (; k0 < K; k0 += 4) {
    const double av = NEG ? -ap[k0 * a_ks] : ap[k0 * a_ks];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[k0 * b_ks], acc, 0, 0, 0);
  }
}
// two output tiles sharing the B fragment (A0 / A1 differ): halves the LDS traffic of the B operand; K % 8 == 0
DEV void mma_tile2(d4_t& acc0, d4_t& acc1, const double* A0, const double* A1, int a_is, int a_ks, const double* B, int b_ks, int b_js, int K, int lane) {
  const int i = lane & 15, kk = lane >> 4;
  const double* ap0 = A0 + i * a_is + kk * a_ks;
  const double* ap1 = A1 + i * a_is + kk * a_ks;
  const double* bp = B + kk * b_ks + i * b_js;
  for (int k0 = 0; k0 < K; k0 += 8) {
    const double a00 = ap0[k0 * a_ks], a01 = ap0[(k0 + 4) * a_ks], a10 = ap1[k0 * a_ks], a11 = ap1[(k0 + 4) * a_ks];
    const double b0 = bp[k0 * b_ks], b1 = bp[(k0 + 4) * b_ks];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a10, b0, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, b1, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, b1, acc1, 0, 0, 0);
  }
}
// two output tiles sharing the B fragment, with independent strides of the two A operands (acc0 += A0 B, acc1 += A1 B): one LDS
// read of B serves both, and the two accumulator chains interleave on the matrix core.  K % 4 == 0; groups of 16 software-pipelined
// like mma_tile.
DEV void mma_tile_2a(d4_t& acc0, d4_t& acc1, const double* A0, int a0_is, int a0_ks, const double* A1, int a1_is, int a1_ks,
                     const double* B, int b_ks, int b_js, int K, int lane) {
  const int i = lane & 15, kk = lane >> 4;
  const double* ap0 = A0 + i * a0_is + kk * a0_ks;
  const double* ap1 = A1 + i * a1_is + kk * a1_ks;
  const double* bp = B + kk * b_ks + i * b_js;
  const int Kmain = K & ~15;
  int k0 = 0;
  if (Kmain > 0) {
    double x0 = ap0[0], x1 = ap0[4 * a0_ks], x2 = ap0[8 * a0_ks], x3 = ap0[12 * a0_ks];
    double y0 = ap1[0], y1 = ap1[4 * a1_ks], y2 = ap1[8 * a1_ks], y3 = ap1[12 * a1_ks];
    double b0 = bp[0], b1 = bp[4 * b_ks], b2 = bp[8 * b_ks], b3 = bp[12 * b_ks];
    for (; k0 < Kmain; k0 += 16) {
      const int kn = (k0 + 16 < Kmain) ? k0 + 16 : k0;  // the last group re-requests its own operands (unused)
      const double nx0 = ap0[kn * a0_ks], nx1 = ap0[(kn + 4) * a0_ks], nx2 = ap0[(kn + 8) * a0_ks], nx3 = ap0[(kn + 12) * a0_ks];
      const double ny0 = ap1[kn * a1_ks], ny1 = ap1[(kn + 4) * a1_ks], ny2 = ap1[(kn + 8) * a1_ks], ny3 = ap1[(kn + 12) * a1_ks];
      const double nb0 = bp[kn * b_ks], nb1 = bp[(kn + 4) * b_ks], nb2 = bp[(kn + 8) * b_ks], nb3 = bp[(kn + 12) * b_ks];
      __builtin_amdgcn_sched_barrier(0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0, b0, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b1, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1, b1, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b2, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y2, b2, acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b3, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y3, b3, acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      x0 = nx0; x1 = nx1; x2 = nx2; x3 = nx3; y0 = ny0; y1 = ny1; y2 = ny2; y3 = ny3; b0 = nb0; b1 = nb1; b2 = nb2; b3 = nb3;
    }
  }
  for (; k0 < K; k0 += 4) {
    const double bv = bp[k0 * b_ks];
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap0[k0 * a0_ks], bv, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap1[k0 * a1_ks], bv, acc1, 0, 0, 0);
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

// value of `v` held by lane SRC (compile-time constant) broadcast to the whole wavefront
template <int SRC> DEV double readlane_d(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), SRC);
  const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), SRC);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// acc += (value of `src` held by lane C of the lane's own row of 16 lanes) * mul — ONE fp64 VALU instruction: the DPP form of
// v_fmac_f64 with row_newbcast (the only DPP control 64-bit operands take on CDNA3/4).  The 16x16 factorisation below keeps identical
// copies of the block in the four 16-lane rows of the wavefront, so "lane C of the row" is "lane C".  Replaces two v_readlane_b32
// (through SGPRs) + a v_fma_f64 per term: 5 us -> ~2 us per 16x16 block.  (s_nop 1: a VALU write of a VGPR that a DPP instruction
// reads needs two wait states, and the hazard recogniser does not look inside inline asm.)
template <int C> DEV void fmac_bcast(double& acc, double src, double mul) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(C));
}
template <int C> DEV double bcast_row(double src) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(C));
  return r;
}
template <int J, int C> struct CholCol {
  static DEV void run(double (&d)[16], double l, double nl) { fmac_bcast<C>(d[C], l, nl); CholCol<J, C + 1>::run(d, l, nl); }  // d[C] -= l * l_C
};
template <int J> struct CholCol<J, 16> { static DEV void run(double (&)[16], double, double) {} };
template <int J> struct CholStep {
  static DEV void run(double (&d)[16], double (&invd)[16], bool& ok) {
    const double djj = bcast_row<J>(d[J]);
    ok = ok && (djj > 0.0);
    double inv = rsqrt(djj);
    inv = inv * (1.5 - 0.5 * djj * inv * inv);  // one Newton step: 1/sqrt(djj) to full precision
    invd[J] = inv;
    const double l = d[J] * inv;  // lane J: sqrt(djj); lanes r > J: L[r][J]
    d[J] = l;
    CholCol<J, J + 1>::run(d, l, -l);
    CholStep<J + 1>::run(d, invd, ok);
  }
};
template <> struct CholStep<16> { static DEV void run(double (&)[16], double (&)[16], bool&) {} };
// (nd = -L: the sign rides on the broadcast operand, so no negated copy of x is kept — 32 VGPRs less, same bits)
template <int R, int K> struct InvDot {
  static DEV void run(const double (&nd)[16], const double (&x)[16], double& acc) { fmac_bcast<R>(acc, nd[K], x[K]); InvDot<R, K + 1>::run(nd, x, acc); }  // acc -= L[R][K] x[K]
};
template <int R> struct InvDot<R, R> { static DEV void run(const double (&)[16], const double (&)[16], double&) {} };
template <int R> struct InvRow {
  static DEV void run(const double (&nd)[16], const double (&invd)[16], double (&x)[16], int lane) {
    double acc = ((lane & 15) == R) ? 1.0 : 0.0;
    InvDot<R, 0>::run(nd, x, acc);
    x[R] = acc * invd[R];
    InvRow<R + 1>::run(nd, invd, x, lane);
  }
};
template <> struct InvRow<16> { static DEV void run(const double (&)[16], const double (&)[16], double (&)[16], int) {} };
// The same inverse RIGHT-looking: once x[K] = (row K of L^-1) is final, every later row R gets its term  x[R] -= L[R][K] x[K]  — fifteen
// independent instructions instead of a dot product per row whose R terms wait for one another (120 dependent fp64 operations per
// block).  Each x[R] receives its terms in the order K = 0, 1, ...: the same bits.  Opt-in (CHOL16_RIGHT_LOOKING before this header):
// the stage kernel's one-wavefront factorisation chain gains 0.6 us per knot, the Riccati sweep loses 5 % with it (more registers live
// across its tile phases) — profiles/r03_experiments.txt.
template <int K, int R> struct InvUpd {
  static DEV void run(const double (&nd)[16], double (&x)[16]) { fmac_bcast<R>(x[R], nd[K], x[K]); InvUpd<K, R + 1>::run(nd, x); }  // x[R] -= L[R][K] x[K]
  static DEV void run_neg(double nl, double (&x)[16]) { fmac_bcast<R>(x[R], nl, x[K]); InvUpd<K, R + 1>::run_neg(nl, x); }          // (nl = -column K of L, lane r: -L[r][K])
};
template <int K> struct InvUpd<K, 16> { static DEV void run(const double (&)[16], double (&)[16]) {} static DEV void run_neg(double, double (&)[16]) {} };
template <int K> struct InvCol {
  static DEV void run(const double (&nd)[16], const double (&invd)[16], double (&x)[16]) {
    x[K] *= invd[K];
    InvUpd<K, K + 1>::run(nd, x);
    InvCol<K + 1>::run(nd, invd, x);
  }
};
template <> struct InvCol<16> { static DEV void run(const double (&)[16], const double (&)[16], double (&)[16]) {} };

// Factorisation and inverse in ONE pass (CHOL16_FUSED): as soon as column K of L is known (step K of the factorisation), row K of L^-1 is
// final (x[K] *= 1 / L[K][K]) and every later row takes its term x[R] -= L[R][K] x[K] — the 2 (15 - K) updates of step K (trailing block
// and inverse) are mutually independent and depend only on the column just scaled, so the dependent chain of the whole block is the 16
// pivots (broadcast, rsqrt, one Newton step, one multiply), not 16 pivots + 120 accumulations.  Same additions in the same order as
// InvCol / CholStep: the same bits.
template <int K> struct FusedStep {
  static DEV void run(double (&d)[16], double (&x)[16], bool& ok) {
    const double dkk = bcast_row<K>(d[K]);
    ok = ok && (dkk > 0.0);
    double inv = rsqrt(dkk);
    inv = inv * (1.5 - 0.5 * dkk * inv * inv);
    const double l = d[K] * inv;   // lane K: sqrt(dkk) ; lanes r > K: L[r][K]
    d[K] = l;
    x[K] *= inv;                   // row K of L^-1 (lane c: column c)
    const double nl = -l;
    CholCol<K, K + 1>::run(d, l, nl);   // d[C] -= L[r][K] L[C][K], C > K
    InvUpd<K, K + 1>::run_neg(nl, x);   // x[R] -= L[R][K] x[K], R > K
    FusedStep<K + 1>::run(d, x, ok);
  }
};
template <> struct FusedStep<16> { static DEV void run(double (&)[16], double (&)[16], bool&) {} };

// ---- the same block on the matrix cores (round 5) -------------------------------------------------------------------------------------
// chol16_wave below keeps a row per lane and spends its time issuing ~256 DPP fmacs (v_fmac_f64_dpp row_newbcast: ~28 clocks each,
// tools/ubench/chol16: 7 200 clocks a block however the dependencies are arranged).  Here the block lives in ACCUMULATOR layout (lane l,
// register q: row (l >> 4) + 4 q, column l & 15 — all 64 lanes hold different entries) and is factorised right-looking in four panels of
// four columns:
//   1. the panel columns (16 x 4) and row block p of W (the running right-hand side of the inverse, W = I - L_done X_done) go through an LDS
//      scratch: every lane reads the 4 x 4 pivot block (uniform addresses), its own row of the panel and its own column of W's row block;
//   2. every lane factorises the pivot block T T^T and inverts T in registers (37 dependent fp64 operations, redundantly);
//   3. lane (x, k) = (l & 15, l >> 4) forms L[x][c0 + k] = sum_k' A[x][c0 + k'] Tinv[k][k'] and X[c0 + k][x] = sum_k' Tinv[k][k'] W[c0 + k'][x]:
//      one register each, already in the layout of BOTH operands of v_mfma_f64_16x16x4 (A(i, k) in lane i + 16 k, B(k, j) in lane j + 16 k);
//   4. two MFMAs: A -= Lp Lp^T (trailing block) and W -= Lp Xp.
// 4 x (LDS round trip + pivot block + 2 MFMAs) instead of 16 pivots + 256 broadcast-fmacs.  `scratch`: 128 doubles of LDS that nobody
// else touches during the call (the callers pass the output block LIb itself: the rows of the inverse are stored at the end).
DEV double rsqrt_refined(double d) { double i = rsqrt(d); return i * (1.5 - 0.5 * d * i * i); }
// ncols (wavefront-uniform): rows / columns from ncols on are identity padding — their panels factorise to themselves and are skipped (the Schur block of a knot
// with three active rows works on one panel instead of four)
DEV bool chol16_wave_mfma(double* D, int ld, double* LIb, int lane, int ncols = 16) {
  const int npan = (ncols + 3) >> 2;
  const int g = lane >> 4, c = lane & 15;
  d4_t t, w;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = g + 4 * q;
    t[q] = (row >= c) ? D[row * ld + c] : D[c * ld + row];   // the lower triangle, mirrored
    w[q] = (row == c) ? 1.0 : 0.0;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();   // every lane holds its entries: D / LIb may be overwritten from here on
  double* P = LIb;        // [16][4] panel columns
  double* Wp = LIb + 64;  // [4][16] row block p of W
  double xp[4];           // lane (j = c, k = g): X[4 p + k][j], p = 0 .. 3 (rows of L^-1, stored at the end)
  bool ok = true;
  const bool store_l = (D != LIb);
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c0 = 4 * p;
    if (p >= npan) { xp[p] = w[p]; continue; }   // identity pivot block: X rows = the rows of W (lane (j = c, k = g) holds W[4 p + g][c] in w[p])
    if ((c >> 2) == p) {
#pragma unroll
      for (int q = 0; q < 4; ++q) P[(g + 4 * q) * 4 + (c & 3)] = t[q];
    }
    Wp[g * 16 + c] = w[p];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // pivot block (uniform reads), own panel row, own column of W's row block
    const double a00 = P[(c0 + 0) * 4 + 0];
    const double a10 = P[(c0 + 1) * 4 + 0], a11 = P[(c0 + 1) * 4 + 1];
    const double a20 = P[(c0 + 2) * 4 + 0], a21 = P[(c0 + 2) * 4 + 1], a22 = P[(c0 + 2) * 4 + 2];
    const double a30 = P[(c0 + 3) * 4 + 0], a31 = P[(c0 + 3) * 4 + 1], a32 = P[(c0 + 3) * 4 + 2], a33 = P[(c0 + 3) * 4 + 3];
    const double r0 = P[c * 4 + 0], r1 = P[c * 4 + 1], r2 = P[c * 4 + 2], r3 = P[c * 4 + 3];
    const double w0 = Wp[0 * 16 + c], w1 = Wp[1 * 16 + c], w2 = Wp[2 * 16 + c], w3 = Wp[3 * 16 + c];
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();   // (the scratch is rewritten by the next panel)
    // T T^T = pivot block, M = T^-1
    ok = ok && (a00 > 0.0);
    const double i0 = rsqrt_refined(a00);
    const double t00 = a00 * i0, t10 = a10 * i0, t20 = a20 * i0, t30 = a30 * i0;
    const double d1 = a11 - t10 * t10;
    ok = ok && (d1 > 0.0);
    const double i1 = rsqrt_refined(d1);
    const double t11 = d1 * i1, t21 = (a21 - t20 * t10) * i1, t31 = (a31 - t30 * t10) * i1;
    const double d2 = a22 - t20 * t20 - t21 * t21;
    ok = ok && (d2 > 0.0);
    const double i2 = rsqrt_refined(d2);
    const double t22 = d2 * i2, t32 = (a32 - t30 * t20 - t31 * t21) * i2;
    const double d3 = a33 - t30 * t30 - t31 * t31 - t32 * t32;
    ok = ok && (d3 > 0.0);
    const double i3 = rsqrt_refined(d3);
    const double t33 = d3 * i3;
    const double m10 = -(t10 * i0) * i1;
    const double m21 = -(t21 * i1) * i2, m20 = -(t20 * i0 + t21 * m10) * i2;
    const double m32 = -(t32 * i2) * i3, m31 = -(t31 * i1 + t32 * m21) * i3, m30 = -(t30 * i0 + t31 * m10 + t32 * m20) * i3;
    // row g of M for this lane (k = g): coefficients of k' = 0 .. 3
    const double mk0 = g == 0 ? i0 : g == 1 ? m10 : g == 2 ? m20 : m30;
    const double mk1 = g == 0 ? 0.0 : g == 1 ? i1 : g == 2 ? m21 : m31;
    const double mk2 = g <= 1 ? 0.0 : g == 2 ? i2 : m32;
    const double mk3 = g <= 2 ? 0.0 : i3;
    double lp = ((r0 * mk0 + r1 * mk1) + r2 * mk2) + r3 * mk3;   // L[x][c0 + k], x = c, k = g
    // the rows of the pivot block itself are T (its strict upper part exactly zero); rows above it take no part any more
    const int xr = c - c0;
    if (xr >= 0 && xr < 4) {
      const double trow0 = xr == 0 ? t00 : xr == 1 ? t10 : xr == 2 ? t20 : t30;
      const double trow1 = xr == 1 ? t11 : xr == 2 ? t21 : t31;
      const double trow2 = xr == 2 ? t22 : t32;
      lp = g > xr ? 0.0 : (g == 0 ? trow0 : g == 1 ? trow1 : g == 2 ? trow2 : t33);
    }
    if (xr < 0) lp = 0.0;
    xp[p] = ((mk0 * w0 + mk1 * w1) + mk2 * w2) + mk3 * w3;       // X[c0 + k][j], j = c, k = g
    if (store_l && xr >= g) D[c * ld + c0 + g] = lp;
    const double nlp = -lp;
    if (p + 1 < npan) t = __builtin_amdgcn_mfma_f64_16x16x4f64(nlp, lp, t, 0, 0, 0);       // A -= Lp Lp^T
    if (p + 1 < npan) w = __builtin_amdgcn_mfma_f64_16x16x4f64(nlp, xp[p], w, 0, 0, 0);    // W -= Lp Xp
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int p = 0; p < 4; ++p) LIb[(4 * p + g) * 17 + c] = xp[p];
  return ok;
}

// One wavefront: Cholesky of the 16x16 block D (lower triangle, leading dimension ld) entirely in registers —
// lane r (< 16) holds row r — then its inverse.  Writes L back over D (lower part) and L^-1 to LIb (ld 17); LIb == D (ld 17): the
// inverse REPLACES the block (the blocked routines never read a diagonal block of L again, only its inverse).
DEV bool chol16_wave_nc(double* D, int ld, double* LIb, int lane, int ncols);
DEV bool chol16_wave(double* D, int ld, double* LIb, int lane, int ncols = 16) {
#ifdef CHOL16_MFMA
  return chol16_wave_mfma(D, ld, LIb, lane, ncols);
#endif
  if (ncols <= 12) return chol16_wave_nc(D, ld, LIb, lane, ncols);
  double d[16], x[16], invd[16];
  const int r = lane & 15;
#pragma unroll
  for (int cidx = 0; cidx < 16; ++cidx) d[cidx] = D[r * ld + cidx];
  bool ok = true;
#ifdef CHOL16_FUSED
#pragma unroll
  for (int cidx = 0; cidx < 16; ++cidx) x[cidx] = (r == cidx) ? 1.0 : 0.0;
  FusedStep<0>::run(d, x, ok);
  if (lane < 16 && D != LIb) {
#pragma unroll
    for (int cidx = 0; cidx < 16; ++cidx) if (cidx <= r) D[r * ld + cidx] = d[cidx];
  }
  if (lane < 16) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) LIb[rr * 17 + lane] = x[rr];
  }
  (void)invd;
  return ok;
#endif
  CholStep<0>::run(d, invd, ok);
  if (lane < 16 && D != LIb) {
#pragma unroll
    for (int cidx = 0; cidx < 16; ++cidx) if (cidx <= r) D[r * ld + cidx] = d[cidx];
  }
#pragma unroll
#ifdef CHOL16_RIGHT_LOOKING
  for (int cidx = 0; cidx < 16; ++cidx) { x[cidx] = (r == cidx) ? 1.0 : 0.0; d[cidx] = -d[cidx]; }
  InvCol<0>::run(d, invd, x);
#else
  for (int cidx = 0; cidx < 16; ++cidx) { x[cidx] = 0.0; d[cidx] = -d[cidx]; }
  InvRow<0>::run(d, invd, x, lane);  // lane c of every 16-lane row builds column c of L^-1
#endif
  if (lane < 16) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) LIb[rr * 17 + lane] = x[rr];
  }
  return ok;
}

// ---- the same block when its rows / columns from NC on are identity padding (round 6) ---------------------------------------------------------------
// chol16_wave spends its time issuing 16 pivots + 240 broadcast multiply-adds whatever the block holds.  The LAST diagonal block of the joint-space inertia has
// nv mod 16 real columns (6 of 16 for the complete Talos: 38 = 32 + 6) and the contact block S has 6 contacts' rows (6 or 12 of 16): their padding factorises to
// itself.  Only the leading NC x NC part is worked on — NC (NC - 1) multiply-adds instead of 240 (30 for NC = 6, 132 for NC = 12) — with the same operations in
// the same order on that part: the same bits as chol16_wave (CHOL16_RIGHT_LOOKING form).
template <int J, int C, int END> struct CholColN {
  static DEV void run(double (&d)[16], double l, double nl) { fmac_bcast<C>(d[C], l, nl); CholColN<J, C + 1, END>::run(d, l, nl); }
};
template <int J, int END> struct CholColN<J, END, END> { static DEV void run(double (&)[16], double, double) {} };
template <int J, int END> struct CholStepN {
  static DEV void run(double (&d)[16], double (&invd)[16], bool& ok) {
    const double djj = bcast_row<J>(d[J]);
    ok = ok && (djj > 0.0);
    double inv = rsqrt(djj);
    inv = inv * (1.5 - 0.5 * djj * inv * inv);
    invd[J] = inv;
    const double l = d[J] * inv;
    d[J] = l;
    CholColN<J, J + 1, END>::run(d, l, -l);
    CholStepN<J + 1, END>::run(d, invd, ok);
  }
};
template <int END> struct CholStepN<END, END> { static DEV void run(double (&)[16], double (&)[16], bool&) {} };
template <int K, int R, int END> struct InvUpdN {
  static DEV void run(const double (&nd)[16], double (&x)[16]) { fmac_bcast<R>(x[R], nd[K], x[K]); InvUpdN<K, R + 1, END>::run(nd, x); }
};
template <int K, int END> struct InvUpdN<K, END, END> { static DEV void run(const double (&)[16], double (&)[16]) {} };
template <int K, int END> struct InvColN {
  static DEV void run(const double (&nd)[16], const double (&invd)[16], double (&x)[16]) {
    x[K] *= invd[K];
    InvUpdN<K, K + 1, END>::run(nd, x);
    InvColN<K + 1, END>::run(nd, invd, x);
  }
};
template <int END> struct InvColN<END, END> { static DEV void run(const double (&)[16], const double (&)[16], double (&)[16]) {} };
template <int NC> DEV bool chol16_wave_n(double* D, int ld, double* LIb, int lane) {
  if constexpr (NC >= 16) return chol16_wave(D, ld, LIb, lane);
  double d[16], x[16], invd[16];
  const int r = lane & 15;
#pragma unroll
  for (int cidx = 0; cidx < 16; ++cidx) { d[cidx] = (cidx < NC) ? D[r * ld + cidx] : 0.0; invd[cidx] = 1.0; }
  bool ok = true;
  CholStepN<0, NC>::run(d, invd, ok);
  if (lane < 16 && D != LIb) {
#pragma unroll
    for (int cidx = 0; cidx < NC; ++cidx) if (cidx <= r) D[r * ld + cidx] = d[cidx];   // (the padding part of the lower triangle is identity already)
  }
#pragma unroll
  for (int cidx = 0; cidx < 16; ++cidx) { x[cidx] = (r == cidx) ? 1.0 : 0.0; d[cidx] = -d[cidx]; }
  InvColN<0, NC>::run(d, invd, x);
  if (lane < 16) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) LIb[rr * 17 + lane] = x[rr];
  }
  return ok;
}
// run-time column count (wavefront-uniform): the register form on the leading 4 / 8 / 12 columns when the rest is padding — the same bits as the full form on those
DEV bool chol16_wave_nc(double* D, int ld, double* LIb, int lane, int ncols) {
  if (ncols <= 4) return chol16_wave_n<4>(D, ld, LIb, lane);
  if (ncols <= 8) return chol16_wave_n<8>(D, ld, LIb, lane);
  if (ncols <= 12) return chol16_wave_n<12>(D, ld, LIb, lane);
  return chol16_wave_n<16>(D, ld, LIb, lane);
}

// Blocked Cholesky of the (16 nb) x (16 nb) matrix A in LDS (lower triangle; pad rows/cols must be identity).
// L overwrites the lower block triangle, LI[bi] (272 doubles each, ld 17) receives the inverse of diagonal block bi.
// the whole blocked factorisation by ONE wavefront (nb <= 3): its LDS operations execute in order, no workgroup barrier
// n_real: rows / columns from n_real on are identity padding (only the last diagonal block can hold some: chol16_wave skips its padding panels)
DEV bool chol_blocked_wave(double* A, int ld, int nb, double* LI, int lane, int n_real = 1 << 20) {
  bool ok = true;
  for (int kb = 0; kb < nb && ok; ++kb) {
    const int nc_ = n_real - 16 * kb;
    ok = chol16_wave(A + (kb * 16) * ld + kb * 16, ld, LI + kb * 272, lane, nc_ < 16 ? (nc_ > 0 ? nc_ : 0) : 16);
    for (int ri = kb + 1; ri < nb; ++ri) {
      d4_t acc = d4_t{0, 0, 0, 0};
      mma_tile<false>(acc, A + (ri * 16) * ld + kb * 16, ld, 1, LI + kb * 272, 1, 17, 16, lane);
      tile_store(A + (ri * 16) * ld + kb * 16, ld, acc, lane);
    }
    for (int ri = kb + 1; ri < nb; ++ri)
      for (int cj = kb + 1; cj <= ri; ++cj) {
        double* Ct = A + (ri * 16) * ld + cj * 16;
        d4_t acc = tile_load(Ct, ld, lane);
        mma_tile<true>(acc, A + (ri * 16) * ld + kb * 16, ld, 1, A + (cj * 16) * ld + kb * 16, 1, ld, 16, lane);
        tile_store(Ct, ld, acc, lane);
      }
  }
  return ok;
}

#ifndef CHOL_ONE_WAVE_BLOCKS
#define CHOL_ONE_WAVE_BLOCKS 3
#endif
DEV bool chol_blocked(double* A, int ld, int nb, double* LI, int tid, int* flag) {
  const int lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0) *flag = 1;
  __syncthreads();
  if (nb <= CHOL_ONE_WAVE_BLOCKS) {
    // small matrices: ONE wavefront runs the whole factorisation — its LDS operations execute in order, so the
    // 3 nb - 1 workgroup barriers (and the idle time around the serial 16x16 steps) of the cooperative form go away
    if (wv == 0) {
      const bool ok = chol_blocked_wave(A, ld, nb, LI, lane);
      if (!ok && lane == 0) *flag = 0;
    }
    __syncthreads();
    return *flag != 0;
  }
  for (int kb = 0; kb < nb; ++kb) {
    if (wv == 0) { if (!chol16_wave(A + (kb * 16) * ld + kb * 16, ld, LI + kb * 272, lane) && lane == 0) *flag = 0; }
    __syncthreads();
    if (*flag == 0) return false;
    // panel: L[ri][kb] = A[ri][kb] LI^T   (ri > kb)
    for (int ri = kb + 1 + wv; ri < nb; ri += nw) {
      d4_t acc = d4_t{0, 0, 0, 0};
      mma_tile<false>(acc, A + (ri * 16) * ld + kb * 16, ld, 1, LI + kb * 272, 1, 17, 16, lane);
      tile_store(A + (ri * 16) * ld + kb * 16, ld, acc, lane);
    }
    __syncthreads();
    // trailing update of the lower block triangle: A[ri][cj] -= L[ri][kb] L[cj][kb]^T
    const int rem = nb - kb - 1;
    for (int t = wv; t < rem * (rem + 1) / 2; t += nw) {
      int ri = 0, acc_t = 0;
      while (acc_t + ri + 1 <= t) { acc_t += ri + 1; ++ri; }
      const int cj = t - acc_t;
      double* Ct = A + ((kb + 1 + ri) * 16) * ld + (kb + 1 + cj) * 16;
      d4_t acc = tile_load(Ct, ld, lane);
      mma_tile<true>(acc, A + ((kb + 1 + ri) * 16) * ld + kb * 16, ld, 1, A + ((kb + 1 + cj) * 16) * ld + kb * 16, 1, ld, 16, lane);
      tile_store(Ct, ld, acc, lane);
    }
    __syncthreads();
  }
  return true;
}

// B <- L^-1 B (forward) for the column blocks owned by this wavefront; B is (16 nb) x (16 ncb), leading dim ldb.
// Column blocks are independent, LDS operations of one wavefront execute in order: no workgroup barrier needed.
DEV void trsm_fwd_blocked(const double* Lm, int ld, const double* LI, int nb, double* Bm, int ldb, int ncb, int wv, int nw, int lane) {
  for (int cj = wv; cj < ncb; cj += nw) {
    for (int bi = 0; bi < nb; ++bi) {
      double* Bt = Bm + (bi * 16) * ldb + cj * 16;
      d4_t acc = tile_load(Bt, ldb, lane);
      mma_tile<true>(acc, Lm + (bi * 16) * ld, ld, 1, Bm + cj * 16, ldb, 1, bi * 16, lane);
      tile_store(Bt, ldb, acc, lane);
      d4_t acc2 = d4_t{0, 0, 0, 0};
      mma_tile<false>(acc2, LI + bi * 272, 17, 1, Bt, ldb, 1, 16, lane);
      tile_store(Bt, ldb, acc2, lane);
    }
  }
}
// B <- L^-T B (backward)
DEV void trsm_bwd_blocked(const double* Lm, int ld, const double* LI, int nb, double* Bm, int ldb, int ncb, int wv, int nw, int lane) {
  for (int cj = wv; cj < ncb; cj += nw) {
    for (int bi = nb - 1; bi >= 0; --bi) {
      double* Bt = Bm + (bi * 16) * ldb + cj * 16;
      d4_t acc = tile_load(Bt, ldb, lane);
      mma_tile<true>(acc, Lm + ((bi + 1) * 16) * ld + bi * 16, 1, ld, Bm + ((bi + 1) * 16) * ldb + cj * 16, ldb, 1, (nb - 1 - bi) * 16, lane);
      tile_store(Bt, ldb, acc, lane);
      d4_t acc2 = d4_t{0, 0, 0, 0};
      mma_tile<false>(acc2, LI + bi * 272, 1, 17, Bt, ldb, 1, 16, lane);
      tile_store(Bt, ldb, acc2, lane);
    }
  }
}

// ---- tile-packed lower block triangle -------------------------------------------------------------------------
// The whole-body stage kernel keeps M = L L^T as the nb (nb + 1) / 2 tiles of its lower block triangle, tile (bi, bj), bj <= bi, at
// T + (bi (bi + 1) / 2 + bj) * 272 with leading dimension 17: no storage for the upper triangle, and the inverse of a diagonal block
// replaces the block (half the LDS of the square layout + separate inverses).
DEV double* ptile(double* T, int bi, int bj) { return T + (bi * (bi + 1) / 2 + bj) * 272; }
DEV const double* ctile(const double* T, int bi, int bj) { return T + (bi * (bi + 1) / 2 + bj) * 272; }

// acc (+/-)= A Breg: A(i, k) at A[i * a_is + k * a_ks] in LDS, the 16 x 16 operand B in REGISTERS in the fragment layout of a result tile
// (register r of lane l = B((l >> 4) + 4 r, l & 15)) — which is exactly what the k-step r of the MFMA wants from lane l, so a chain of
// dependent tile products (triangular solves) never goes back to LDS with its intermediate results.
template <bool NEG>
DEV void mma_tile_rb(d4_t& acc, const double* A, int a_is, int a_ks, const d4_t& B, int lane) {
  const double* ap = A + (lane & 15) * a_is + (lane >> 4) * a_ks;
  double a0 = ap[0], a1 = ap[4 * a_ks], a2 = ap[8 * a_ks], a3 = ap[12 * a_ks];
  if (NEG) { a0 = -a0; a1 = -a1; a2 = -a2; a3 = -a3; }
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, B[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, B[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, B[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, B[3], acc, 0, 0, 0);
}

// Cholesky of the tile-packed matrix by ONE wavefront (pad rows / columns must be identity): off-diagonal tiles receive L, diagonal
// tiles the INVERSE of their Cholesky factor.
DEV bool chol_tiles_wave(double* T, int nb, int lane) {
  bool ok = true;
  for (int kb = 0; kb < nb && ok; ++kb) {
    double* Dk = ptile(T, kb, kb);
    ok = chol16_wave(Dk, 17, Dk, lane);
    for (int ri = kb + 1; ri < nb; ++ri) {  // panel: L[ri][kb] = A[ri][kb] LI^T
      double* Pt = ptile(T, ri, kb);
      d4_t acc = d4_t{0, 0, 0, 0};
      mma_tile<false>(acc, Pt, 17, 1, Dk, 1, 17, 16, lane);
      tile_store(Pt, 17, acc, lane);
    }
    for (int ri = kb + 1; ri < nb; ++ri)
      for (int cj = kb + 1; cj <= ri; ++cj) {  // trailing update: A[ri][cj] -= L[ri][kb] L[cj][kb]^T
        double* Ct = ptile(T, ri, cj);
        d4_t acc = tile_load(Ct, 17, lane);
        mma_tile<true>(acc, ptile(T, ri, kb), 17, 1, ptile(T, cj, kb), 1, 17, 16, lane);
        tile_store(Ct, 17, acc, lane);
      }
  }
  return ok;
}
// the same by the whole workgroup (nb > 3): diagonal block by wavefront 0, panel and trailing tiles dealt to the wavefronts
DEV bool chol_tiles(double* T, int nb, int tid, int* flag) {
  const int lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  if (tid == 0) *flag = 1;
  __syncthreads();
  if (nb <= 3) {
    if (wv == 0) { if (!chol_tiles_wave(T, nb, lane) && lane == 0) *flag = 0; }
    __syncthreads();
    return *flag != 0;
  }
  for (int kb = 0; kb < nb; ++kb) {
    double* Dk = ptile(T, kb, kb);
    if (wv == 0) { if (!chol16_wave(Dk, 17, Dk, lane) && lane == 0) *flag = 0; }
    __syncthreads();
    if (*flag == 0) return false;
    for (int ri = kb + 1 + wv; ri < nb; ri += nw) {
      double* Pt = ptile(T, ri, kb);
      d4_t acc = d4_t{0, 0, 0, 0};
      mma_tile<false>(acc, Pt, 17, 1, Dk, 1, 17, 16, lane);
      tile_store(Pt, 17, acc, lane);
    }
    __syncthreads();
    const int rem = nb - kb - 1;
    for (int t = wv; t < rem * (rem + 1) / 2; t += nw) {
      int ri = 0, acc_t = 0;
      while (acc_t + ri + 1 <= t) { acc_t += ri + 1; ++ri; }
      const int cj = t - acc_t;
      double* Ct = ptile(T, kb + 1 + ri, kb + 1 + cj);
      d4_t acc = tile_load(Ct, 17, lane);
      mma_tile<true>(acc, ptile(T, kb + 1 + ri, kb), 17, 1, ptile(T, kb + 1 + cj, kb), 1, 17, 16, lane);
      tile_store(Ct, 17, acc, lane);
    }
    __syncthreads();
  }
  return true;
}
// B <- L^-1 B / B <- L^-T B on an LDS operand B ((16 nb) x (16 ncb), leading dimension ldb), column blocks dealt to the wavefronts
DEV void trsm_fwd_tiles(const double* T, int nb, double* Bm, int ldb, int ncb, int wv, int nw, int lane) {
  for (int cj = wv; cj < ncb; cj += nw)
    for (int bi = 0; bi < nb; ++bi) {
      double* Bt = Bm + (bi * 16) * ldb + cj * 16;
      d4_t acc = tile_load(Bt, ldb, lane);
      for (int bj = 0; bj < bi; ++bj) mma_tile<true>(acc, ctile(T, bi, bj), 17, 1, Bm + (bj * 16) * ldb + cj * 16, ldb, 1, 16, lane);
      d4_t acc2 = d4_t{0, 0, 0, 0};
      mma_tile_rb<false>(acc2, ctile(T, bi, bi), 17, 1, acc, lane);
      tile_store(Bt, ldb, acc2, lane);
    }
}
DEV void trsm_bwd_tiles(const double* T, int nb, double* Bm, int ldb, int ncb, int wv, int nw, int lane) {
  for (int cj = wv; cj < ncb; cj += nw)
    for (int bi = nb - 1; bi >= 0; --bi) {
      double* Bt = Bm + (bi * 16) * ldb + cj * 16;
      d4_t acc = tile_load(Bt, ldb, lane);
      for (int bj = bi + 1; bj < nb; ++bj) mma_tile<true>(acc, ctile(T, bj, bi), 1, 17, Bm + (bj * 16) * ldb + cj * 16, ldb, 1, 16, lane);
      d4_t acc2 = d4_t{0, 0, 0, 0};
      mma_tile_rb<false>(acc2, ctile(T, bi, bi), 1, 17, acc, lane);
      tile_store(Bt, ldb, acc2, lane);
    }
}
