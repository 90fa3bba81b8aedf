// eval_multibody.h — per-knot evaluation of whole-body stages on the GPU: the stage of fulldynamic_talos.py:100-232
// (MultibodyConstraintFwdDynamics = pin.constraintDynamics + its derivatives, IntegratorSemiImplEuler, state /
// control / centroidal-momentum / frame-placement / contact-force costs, torque & joint boxes, wrench cones).
//
// One workgroup (256 threads) per (knot, instance, linesearch candidate).  All rigid-body quantities are kept in
// LDS in a WORLD-FRAME formulation (spatial vectors [lin; ang] taken at the world origin): after one pass over
// the kinematic tree every entry of M, d tau/dq, d tau/dv, the contact Jacobians and their derivatives is a
// 6-dimensional dot product of per-dof vectors, so the nv x nv blocks are filled by all lanes in parallel with
// no further dependency on the tree (DESIGN.md §"Whole-body stage kernel" derives the formulas; they are
// cross-checked against the AD-based oracle: on the CPU through its C++ port (oracle/cpu_port/eval_closed_form.hpp, tests/test_cpu_port.py:
// per-phase dumps at 1e-9), on the GPU through the parity tests).
#pragma once
#include <type_traits>
#include "eval_common.h"
#include "mfma_blocks.h"
#include "eval_multibody_host.h"

// 6x6 inverse by Gauss-Jordan without pivoting, fully unrolled (registers only); used on M_bb (SPD)
DEV void inv6_unrolled_mb(const double* A, double* Ainv) {
  double M[6][12];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double inv = 1.0 / M[k][k];
#pragma unroll
    for (int j = 0; j < 12; ++j) M[k][j] *= inv;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if (i == k) continue;
      const double l = M[i][k];
#pragma unroll
      for (int j = 0; j < 12; ++j) M[i][j] -= l * M[k][j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) Ainv[i * 6 + j] = M[i][6 + j];
}

// ---- structure-of-arrays accessors (component e of element i at A[e * ld + i]) -------------------------------
DEV S6 ldc6(const double* A, int ld, int i) { S6 r; for (int e = 0; e < 6; ++e) r.v[e] = A[e * ld + i]; return r; }
DEV void stc6(double* A, int ld, int i, const S6& a) { for (int e = 0; e < 6; ++e) A[e * ld + i] = a.v[e]; }
DEV M3 ldcm3(const double* A, int ld, int i) { M3 r; for (int e = 0; e < 9; ++e) r.m[e] = A[e * ld + i]; return r; }
DEV V3 ldcv3(const double* A, int ld, int i) { return v3(A[i], A[ld + i], A[2 * ld + i]); }
// symmetric 6x6 blocks are stored packed: entry (r, c), r <= c, at row sym6(r, c) of the 21-row array
DEV constexpr int sym6(int r, int c) { return r <= c ? (r * (13 - r)) / 2 + (c - r) : (c * (13 - c)) / 2 + (r - c); }
struct Y21 { double y[21]; };
DEV Y21 ldy21(const double* A, int ld, int i) { Y21 r; for (int e = 0; e < 21; ++e) r.y[e] = A[e * ld + i]; return r; }
DEV S6 sym_mul(const Y21& Y, const S6& x) {
  S6 r;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    double s = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) s += Y.y[sym6(a, b)] * x.v[b];
    r.v[a] = s;
  }
  return r;
}
struct G36 { double g[36]; };
DEV G36 ldg36(const double* A, int ld, int i) { G36 r; for (int e = 0; e < 36; ++e) r.g[e] = A[e * ld + i]; return r; }

// lane ids pass through an empty asm at every phase boundary: index arithmetic stays phase-local instead of being
// kept live (and spilled) across the whole kernel — see RIC_LAUNDER in riccati_mfma.h
#define EV_LAUNDER() do { asm volatile("" : "+v"(tid)); lane = tid & 63; wv = __builtin_amdgcn_readfirstlane(tid >> 6); } while (0)
// sum of f(j) over the set bits j of `mm` in ascending order, four operands requested per round (the plain loop — find a bit, load, add,
// next bit — waits for one LDS round trip per term: 33 of them for the root's subtree sums).  Same additions in the same order.
template <class F> DEV double mask_sum(unsigned long long mm, F&& f, double s = 0.0) {
  while (mm) {
    const int j0 = __builtin_ctzll(mm); mm &= mm - 1;
    const bool h1 = mm != 0; const int j1 = h1 ? __builtin_ctzll(mm) : j0; mm &= mm - 1;
    const bool h2 = mm != 0; const int j2 = h2 ? __builtin_ctzll(mm) : j0; mm &= mm - 1;
    const bool h3 = mm != 0; const int j3 = h3 ? __builtin_ctzll(mm) : j0; mm &= mm - 1;
    const double a0 = f(j0), a1 = f(j1), a2 = f(j2), a3 = f(j3);
    s += a0;
    if (h1) s += a1;
    if (h2) s += a2;
    if (h3) s += a3;
  }
  return s;
}
// ---- sums over the kinematic tree on the matrix cores (round 6) -----------------------------------------------------------------------------
// Every "sum over the bodies of a subtree" / "sum over the dofs on the path to the root" is a product with a 0 / 1 matrix: OUT(r, c) = sum_k A(r, k) bit_k(mask[c]).
// mask_sum above does it a lane per entry: a data-dependent loop of bit scans and dependent LDS loads whose trip count is the root's (33) in every wavefront that
// holds one of its entries — a third of the kernel's vector instructions.  Here one wavefront forms a 16 x 16 tile of OUT with ceil(K / 4) MFMAs: lane l supplies
// A(l & 15, 4 s + (l >> 4)) (a callable: clamped LDS reads) and the bit 4 s + (l >> 4) of the mask of column l & 15 as a double.  Exact products (x 0 or x 1), fp64
// accumulation in ascending k like the loop — only the association of the additions differs.  Columns >= nc get a zero mask; rows past the operand's last are
// whatever the callable returns for them (finite: it clamps) and are not stored.
template <class FA> DEV d4_t mask_tile(FA&& aval, const unsigned long long* mask, int c0, int nc, int K, int lane, d4_t acc = d4_t{0, 0, 0, 0}) {
  const int col = c0 + (lane & 15), kk = lane >> 4, row = lane & 15;
  unsigned long long mm = mask[col < nc ? col : nc - 1];
  if (col >= nc) mm = 0ull;
  mm >>= kk;
  for (int k0 = 0; k0 < K; k0 += 4) {
    const double bv = (double)(unsigned)(mm & 1ull);
    mm >>= 4;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aval(row, k0 + kk), bv, acc, 0, 0, 0);
  }
  return acc;
}
// rows r0 + (l >> 4) + 4 q < nr, columns c0 + (l & 15) < nc of a result tile into a structure-of-arrays block (row r at out[r * ld + .])
DEV void tile_store_soa(double* out, int ld, int r0, int nr, int c0, int nc, const d4_t& v, int lane) {
  const int c = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; ++q) { const int r = r0 + (lane >> 4) + 4 * q; if (r < nr && c < nc) out[r * ld + c] = v[q]; }
}
// OUT (nr x nc, SoA ld ldo) = A (nr x K, SoA lda) x bits(mask): the tiles dealt to the wavefronts w0, w0 + 1, ... (nws of them), tile index offset t0 so that several
// products of one phase share the wavefronts evenly
DEV int tree_sum_mfma(double* out, int ldo, const double* A, int lda, int nr, const unsigned long long* mask, int nc, int K, int wv, int nws, int t0, int lane) {
  const int rt = (nr + 15) >> 4, ct = (nc + 15) >> 4;
  for (int t = 0; t < rt * ct; ++t) {
    if ((t0 + t) % nws != wv) continue;
    const int r0 = 16 * (t / ct), c0 = 16 * (t % ct);
    const d4_t acc = mask_tile([&](int row, int k) { const int r = r0 + row; return A[(r < nr ? r : nr - 1) * lda + (k < K ? k : K - 1)]; }, mask, c0, nc, K, lane);
    tile_store_soa(out, ldo, r0, nr, c0, nc, acc, lane);
  }
  return t0 + rt * ct;
}
#define MB_AB_U 8  // loads of d a in flight per thread while [A B] is written
#ifdef EV_SUBPROF
#define EV_SUB(slot) EV_PROF(slot)  // developer builds: sub-phases of the one-wavefront solve (tools/phase_timers.py)
#else
#define EV_SUB(slot) do {} while (0)
#endif
#define EV_PROF(slot) do { EV_LAUNDER(); if (TRIAL == 0 && tid == 0 && a.prof && k == 1) { const long long t1_ = clock64(); a.prof[(size_t)b * 64 + 32 + (slot)] += (double)(t1_ - t0_); t0_ = t1_; } } while (0)

#ifndef EVAL_MIN_WAVES
#define EVAL_MIN_WAVES (EVAL_THREADS / 128)  // waves per SIMD the register budget must allow: two workgroups per CU
#endif
static_assert(EVAL_THREADS >= 128 && EVAL_THREADS <= 512 && EVAL_THREADS % 64 == 0, "stage workgroup: 2 to 8 wavefronts");

// ---- implicit differentiation of the contact dynamics for ONE 16-column block, entirely in registers -----------------
//   W = L^-1 R1 ; T = Y^T W - R2 ; Z2 = S^-1 T ; Z1 = L^-T (W - Y Z2) ;  d a = -Z1 ,  d lambda = Z2
// w[bi] = the 16 x 16 tiles of R1 (fragment layout of an MFMA result), t = R2 (rows >= 12 zero).  Mt: tile-packed L with the inverses
// of the diagonal blocks in place, Y16 = L^-1 Jc^T (nvp x 16, zero padded, leading dimension MB_LDY = 17: the row-strided reads of
// W -= Y Z2 were 8-way bank conflicts at 16), LIs = inverse Cholesky factor of S = Y^T Y + mu I.
DEV void implicit_diff_block(d4_t (&w)[4], d4_t& t, const double* Mt, const double* Y16, const double* LIs, int nbm, int lane) {
#pragma unroll
  for (int bi = 0; bi < 4; ++bi) if (bi < nbm) {
    d4_t acc = w[bi];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) if (bj < bi) mma_tile_rb<true>(acc, ctile(Mt, bi, bj), 17, 1, w[bj], lane);
    w[bi] = d4_t{0, 0, 0, 0};
    mma_tile_rb<false>(w[bi], ctile(Mt, bi, bi), 17, 1, acc, lane);
  }
  t = -t;
#pragma unroll
  for (int bi = 0; bi < 4; ++bi) if (bi < nbm) mma_tile_rb<false>(t, Y16 + (bi * 16) * MB_LDY, 1, MB_LDY, w[bi], lane);
  d4_t u = d4_t{0, 0, 0, 0};
  mma_tile_rb<false>(u, LIs, 17, 1, t, lane);
  t = d4_t{0, 0, 0, 0};
  mma_tile_rb<false>(t, LIs, 1, 17, u, lane);
#pragma unroll
  for (int bi = 0; bi < 4; ++bi) if (bi < nbm) mma_tile_rb<true>(w[bi], Y16 + (bi * 16) * MB_LDY, MB_LDY, 1, t, lane);
#pragma unroll
  for (int bi = 3; bi >= 0; --bi) if (bi < nbm) {
    d4_t acc = w[bi];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) if (bj > bi && bj < nbm) mma_tile_rb<true>(acc, ctile(Mt, bj, bi), 1, 17, w[bj], lane);
    w[bi] = d4_t{0, 0, 0, 0};
    mma_tile_rb<false>(w[bi], ctile(Mt, bi, bi), 1, 17, acc, lane);
  }
}

// chol_tiles_wave (mfma_blocks.h) with NLAST real columns in the last diagonal block
template <int NLAST> DEV bool chol_tiles_wave_last(double* T, int nb, int lane) {
  bool ok = true;
  for (int kb = 0; kb < nb && ok; ++kb) {
    double* Dk = ptile(T, kb, kb);
    ok = (kb == nb - 1) ? chol16_wave_n<NLAST>(Dk, 17, Dk, lane) : chol16_wave(Dk, 17, Dk, lane);
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

// Barfoot's Q block with the series coefficients handed in (q_coeffs depends on |w| only: Q(v, w) and Q(-v, -w) share them)
DEV M3 Qmat_c(V3 v, V3 w, double a1, double a2, double a3) {
  const M3 P = skew_m(v), F = skew_m(w);
  const M3 FP = mul(F, P), PF = mul(P, F), FPF = mul(FP, F), FF = mul(F, F);
  M3 Q = scl3(0.5, P);
  Q = add3(Q, scl3(a1, add3(add3(FP, PF), FPF)));
  Q = add3(Q, scl3(a2, add3(add3(mul(FF, P), mul(P, FF)), scl3(-3.0, FPF))));
  Q = add3(Q, scl3(a3, add3(mul(FPF, F), mul(F, FPF))));
  return Q;
}
// 6 x 6 (row-major) [[Ji, -Ji Q Ji], [0, Ji]] scaled by sgn: Jlog6 from its pieces
DEV void jlog6_blocks(const M3& Ji, const M3& Q, double sgn, double* out) {
  const M3 Bm = mul(mul(Ji, Q), Ji);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    out[6 * i + j] = sgn * Ji.m[3 * i + j]; out[6 * (i + 3) + j + 3] = sgn * Ji.m[3 * i + j];
    out[6 * i + j + 3] = -sgn * Bm.m[3 * i + j]; out[6 * (i + 3) + j] = 0.0;
  }
}

// ============================================================================================================
// FJ, FV, FU > 0: the model's body / velocity / control counts as compile-time constants (a free-flyer model: nq = nv + 1, n = 2 nv; FCD: the
// carve-out with the blocks of the contact-constrained dynamics) — the loop bounds, strides and LDS offsets of the carve-out fold into the instructions, as in the
// fixed-dimension instantiations of the Riccati sweep (riccati_mfma.h).  The launcher picks such an instantiation only for that model.
template <int TRIAL, int FJ = 0, int FV = 0, int FU = 0, bool FCD = true>
__global__ void __launch_bounds__(EVAL_THREADS, EVAL_MIN_WAVES) k_eval_multibody(SolverArgs a, Layout KL, double* records, MbArgs mb, int cand0) {
  constexpr bool FX = FJ > 0;
  constexpr MbLds SC_ = FX ? make_mb_lds(FJ, FV, FV + 1, FU, 2 * FV + FU, FCD) : MbLds{};
  MbLds S_ = mb.lds;
  if constexpr (FX) S_ = SC_;
  const MbLds& S = S_;
  Layout L_ = a.L;  // (a local copy whose dimension members are the constants: every use of L.n, L.nz, ... below folds)
  if constexpr (FX) { L_.n = 2 * FV; L_.nx = 2 * FV + 1; L_.m = FU; L_.nz = 2 * FV + FU; L_.nj = FJ; }
  const Layout& L = L_;
  // TRIAL == 3 with one workgroup more per instance (blockIdx.x == N + 1): the SPECULATIVE evaluation of the knot the next tick appends —
  // the accepted terminal state as a running stage with the table of the current last stage and the last control (what the warm-start
  // shift makes of it).  Its record goes to a spare slot (a.spec_knot) ; if the table of the appended stage turns out to be that one
  // (mpc_cycle compares), the next tick takes it as knot N - 1 instead of evaluating it (k_reproject, knot_reused).
  const bool specw = TRIAL == 3 && (int)blockIdx.x == a.L.N + 1;
  const int k = specw ? a.L.N - 1 : ((TRIAL == 0 && a.only_knot >= 0) ? a.only_knot : (int)blockIdx.x);  // stage table, control, multipliers (only_knot: the launch is that knot alone)
  const int b = blockIdx.y;
  constexpr int nthr = EVAL_THREADS;  // (the launcher uses EVAL_THREADS threads)
  int cand = blockIdx.z + cand0;
  int tid = threadIdx.x;
  const InstState& st = a.inst[b];
  // TRIAL: 0 full evaluation, 1 value-only linesearch candidate, 3 the alpha = 1 candidate WITH derivatives, written into the knot
  // records themselves (tick reuse: if the full step is accepted these are the records of the next tick, one knot on),
  // 2 closed-loop simulation stand-in (N2): knot 0's contact
  // dynamics integrated mb.sim_substeps times under u = us[0] - K0 difference(x, xs[0]) (fulldynamic_talos.py:512-530)
  constexpr bool CAND = (TRIAL == 1 || TRIAL == 3);  // evaluated at the candidate point x (+) alpha dx
  if (TRIAL != 2 && (st.done || (CAND && st.skip_step))) return;
  if (TRIAL == 1 && cand > 0 && !st.ls_more) return;  // the full step was accepted: no backtracking candidates needed
  if (TRIAL == 0 && a.only_knot >= 0) { if (k != a.only_knot) return; }  // (refinement of the appended knot: this knot, afresh)
  else if (TRIAL == 0 && knot_reused(a, b, k)) return;     // tick reuse: the record is there already (k_reproject refreshes its projections)
  const int n = L.n, N = L.N, nx = L.nx, nv = S.nv, nq = S.nq, nj = S.nj, nu = L.m;
  const int slot = stage_slot(a, k);
  const int32_t* desc = a.stage_desc + (size_t)slot * L.max_stage_ints;
  const double* P = (a.inst_params ? a.inst_params + (size_t)b * (L.N + 1) * L.max_stage_doubles : a.stage_params) + (size_t)slot * L.max_stage_doubles;
  const int dyn = desc[0];
  const bool has_dyn = dyn == MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER;   // contact-constrained forward dynamics
  if (has_dyn && !S.contact_dyn) { if (threadIdx.x == 0) a.inst[b].done = 5; return; }  // (host bug guard: this carve-out has no room for the factor of M)
  const bool kino = dyn == MPC_DYN_KINODYNAMICS_SEMIEULER;               // kinodynamics: u = [wrenches ; joint accelerations]
  const int m = (has_dyn || kino) ? nu : 0, nz = n + m, nterms = desc[5], c = desc[6];
  const int nk = has_dyn ? desc[1] : 0, nl = 6 * nk;
  const unsigned mg_nz = m ? S.mg_nz : S.mg_n;  // nz = n + nu on a stage with dynamics, n on the terminal knot
  const bool derivs = (TRIAL == 0 || TRIAL == 3);
cand_loop:  // (TRIAL == 1 with mb.ncand_loop: next backtracking candidate of the same knot)
  const double alpha = CAND ? ldexp(1.0, -cand) : 0.0;
  const size_t wg = (TRIAL == 1) ? (((size_t)b * L.n_alpha + cand) * (N + 1) + k) : (specw ? (size_t)L.B * (N + 1) + b : ((size_t)b * (N + 1) + knot_slot(a, k)));
  double* kn = specw ? a.spec_knot + (size_t)b * KL.knot_stride : records + wg * KL.knot_stride;
  double* scr = mb.scratch + (derivs ? wg : 0) * mb.scratch_stride;  // value-only passes never touch it (B (N + 1) + B slots)
  double* dsol = scr;                         // [nv][nz]: d a w.r.t. (q, v, u) — written by the implicit differentiation, read by the integrator
  double* JtG = scr + (size_t)(nv + 12) * L.nz;  // [24][nz] HBM fallback for dense (non-diagonal) weights
  double* WJ = JtG + (size_t)24 * L.nz;

  extern __shared__ __attribute__((aligned(16))) double sm[];
  unsigned long long* anc = (unsigned long long*)((char*)sm + S.anc_bytes_off);
  unsigned long long* sub = anc + nj;    // bodies of the subtree rooted at i
  unsigned long long* dmask = sub + nj;  // dofs of the joints on the path root .. i
  unsigned long long* below = dmask + nj;  // dofs of the joints strictly inside the subtree of i
  int* dof_body = (int*)(below + nj);
  int* parent = dof_body + nv;
  int* jkind = parent + nj;
  int* jidxv = jkind + nj;
  double *oR = sm + S.oR, *op = sm + S.op, *ov = sm + S.ov, *oa = sm + S.oa, *oh = sm + S.oh, *of = sm + S.of, *Fc = sm + S.Fc, *Hc = sm + S.Hc;
  double *oY = sm + S.oY, *Yc = sm + S.Yc, *Bc = sm + S.Bc;
  double *J = sm + S.J, *U = sm + S.U, *Psd = sm + S.Psd, *Psdd = sm + S.Psdd, *Phi = sm + S.Phi, *Bt = sm + S.Bt, *Tq = sm + S.Tq, *Tv = sm + S.Tv;
  double *BcPsd = sm + S.BcPsd, *YcPsd = sm + S.YcPsd;
  double *Mt = sm + S.Mt, *Y16 = sm + S.Y16, *V16 = sm + S.V16, *Sp = sm + S.Sp, *LIs = sm + S.LIs, *DL = sm + S.DL;
  double *gam = sm + S.gam, *bias = sm + S.bias, *acc = sm + S.a, *lam = sm + S.lam;
  const int nvp = S.nvp, nbm = S.nbm, ncb = S.ncb, ldl = S.ldl;
  int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = nthr >> 6;
  double *x = sm + S.x, *u = sm + S.u, *xn = sm + S.xn, *cfr = sm + S.cfr, *small = sm + S.small, *red = sm + S.red;
  double* early = sm + S.early;  // [0,nz) gradient | [nz,2nz) diag(H) | [2nz,2nz+36) base block | [2nz+36, +24) per-term cost | [2nz+60] done flag
  __shared__ int iflag[2];
  __shared__ int ccid_s[2];
  __shared__ int cbody_s[2];  // contact models of the stage and their bodies (two dependent global loads where they were looked up in the loops)
  __shared__ double s_cost;
  // Jacobian staging in LDS, in the region that is dead once the dynamics derivatives are out (Yc, then the factor of M):
  // JS = stacked rows sqrt(W) J of the cost terms (<= 32 rows) ; JL = rows of the constraint term being emitted — the constraints are
  // emitted before the cost rows are stacked, so the two share the region
  double* JS = sm + S.JS;
  double* se3 = sm + S.se3;
  double* JL = JS;
  double* wrs = red + 16;  // sqrt(W) r of the stacked rows
  int* lterm = (int*)(small + 184);  // term records of the stage (24 x MPC_TERM_WORDS ints): read once, coalesced, instead of
                                     // a chain of dependent global loads in every term loop

  const int32_t* mi = a.model_i;
  const double* md = a.model_d;
  const int nframes = mi[3];
  const int32_t* mj = mi + MPC_MODEL_HEADER_WORDS;
  const int32_t* mframe = mj + MPC_MODEL_JOINT_WORDS * nj;
  const int32_t* mcontact = mframe + nframes;
  const double* jd = md + MPC_MODEL_HEADER_DOUBLES;
  const double* fd = jd + MPC_MODEL_JOINT_DOUBLES * nj;
  const double* cd = fd + MPC_MODEL_FRAME_DOUBLES * nframes;
  const double grav[3] = {md[0], md[1], md[2]};
  const double prox_mu = md[3];
  const S6 a0 = kino ? mk6(v3(0, 0, 0), v3(0, 0, 0)) : mk6(v3(-grav[0], -grav[1], -grav[2]), v3(0, 0, 0));

  long long t0_ = clock64();
  // ---- P0: evaluation point, tree tables ---------------------------------------------------------------
  {
    const int kx = specw ? N : k;  // state (the speculative knot: the terminal state, which the shift duplicates: its next state is the same one)
    const double* xs = a.xs + ((size_t)b * (N + 1) + kx) * nx;
    const double* dx = a.dxs + ((size_t)b * (N + 1) + kx) * n;
    if (TRIAL == 2) {
      const double* xsim = mb.sim_u ? a.x0 + (size_t)b * nx : xs;  // torque-driven form: from the measured state
      for (int i = tid; i < nx; i += nthr) { x[i] = xsim[i]; xn[i] = xs[i]; }  // simulated state ; xn = xs[0], the feedback reference
    } else if (CAND) {
      if (wv == 0) state_integrate_group(MPC_SPACE_MULTIBODY, nx, n, xs, dx, alpha, x, lane, 64);
      if (wv == 1 && k < N) state_integrate_group(MPC_SPACE_MULTIBODY, nx, n, specw ? xs : xs + nx, specw ? dx : dx + n, alpha, xn, lane, 64);
    } else {
      for (int i = tid; i < nx; i += nthr) { x[i] = xs[i]; if (k < N) xn[i] = xs[nx + i]; }
    }
    if (k < N && TRIAL != 2) {
      const double* us = a.us + ((size_t)b * N + k) * nu;
      const double* du = a.dus + ((size_t)b * N + k) * nu;
      for (int i = tid; i < nu; i += nthr) u[i] = us[i] + (CAND ? alpha * du[i] : 0.0);
    }
    for (int i = tid; i < nterms * MPC_TERM_WORDS; i += nthr) lterm[i] = desc[MPC_STAGE_HEADER_WORDS + i];
    // tree tables; the bit masks (model constants) were built on the host by mpc_set_model
    const unsigned long long* gmask = (const unsigned long long*)(a.model_i + L.model_mask_off);
    for (int i = tid; i < nj; i += nthr) {
      parent[i] = mj[4 * i]; jkind[i] = mj[4 * i + 1]; jidxv[i] = mj[4 * i + 3];
      anc[i] = gmask[i]; sub[i] = gmask[nj + i]; dmask[i] = gmask[2 * nj + i];
      const int ndof = (mj[4 * i + 1] == MPC_JOINT_FREEFLYER) ? 6 : 1;
      for (int d = 0; d < ndof; ++d) dof_body[mj[4 * i + 3] + d] = i;
      below[i] = gmask[3 * nj + i];
    }
    if (tid == 0) s_cost = 0.0;
    if (has_dyn && tid < nk) { const int cid = desc[2 + tid]; ccid_s[tid] = cid; cbody_s[tid] = mcontact[cid]; }
  }
  // model constants of body `tid`, requested here: their round trip to L2 overlaps the loads above and the barrier
  // (the kernels with derivatives only: the value-only candidates are short of registers right here)
  constexpr bool JD_PREFETCH = (TRIAL == 0 || TRIAL == 3);
  const int wv_inertia = nw > 2 ? 2 : 0;  // wavefront that builds the bodies' spatial inertias in P4 (wavefront 0: bias accelerations, 1: contact frames)
  double jdl[12];
  if (JD_PREFETCH && tid < nj) {
#pragma unroll
    for (int e = 0; e < 12; ++e) jdl[e] = jd[25 * tid + e];
  }
  __syncthreads();
  const double* q = x;
  const double* v = x + nq;
  int sim_sub = 0;
sim_loop:
  if (TRIAL == 2) {
    // feedback law of the low-level loop: u = us[0] - K0 difference(x, xs[0]),  difference(a, b) = b (-) a
    if (!has_dyn) return;
    double* dd = Tq;
    if (mb.sim_u) {  // mpc_simulate_torque: the caller's joint torques, held over the sub-steps
      for (int i = tid; i < nu; i += nthr) u[i] = mb.sim_u[(size_t)b * nu + i];
      __syncthreads();
      goto sim_u_set;
    }
    if (tid == 0) {
      const M3 Rx = quat_to_rot(x + 3), R0 = quat_to_rot(xn + 3);
      V3 ev, ew;
      log6(tmul(Rx, R0), tmul(Rx, v3(xn[0] - x[0], xn[1] - x[1], xn[2] - x[2])), ev, ew);
      dd[0] = ev.x; dd[1] = ev.y; dd[2] = ev.z; dd[3] = ew.x; dd[4] = ew.y; dd[5] = ew.z;
    }
    for (int i = 6 + tid; i < nv; i += nthr) dd[i] = xn[i + 1] - x[i + 1];
    for (int i = tid; i < nv; i += nthr) dd[nv + i] = xn[nq + i] - x[nq + i];
    __syncthreads();
    const double* K0 = gain_ptr(a, b, 0) + L.oK;
    const double* us0 = a.us + (size_t)b * N * nu;
    for (int i = tid; i < nu; i += nthr) {
      double su = us0[i];
      for (int j = 0; j < n; ++j) su -= K0[i * n + j] * dd[j];
      u[i] = su;
    }
    __syncthreads();
  }
sim_u_set:
#define BELOW(kdof, body) ((anc[(body)] >> dof_body[(kdof)]) & 1ull)

  // ---- P1: local joint transforms (stored in the Bc region), then world placements ----------------------
  double* lR = Bc;
  double* lp = Bc + 9 * nj;
  if (tid < nj) {  // (nj <= 64: check_multibody_model)
    const int i = tid;
    M3 Rp;
#pragma unroll
    for (int e = 0; e < 9; ++e) Rp.m[e] = JD_PREFETCH ? jdl[e] : jd[25 * i + e];
    const V3 pp = JD_PREFETCH ? v3(jdl[9], jdl[10], jdl[11]) : ldv3(jd + 25 * i + 9);
    M3 Rj;
    V3 pj = v3(0, 0, 0);
    if (jkind[i] == MPC_JOINT_FREEFLYER) { Rj = quat_to_rot(q + 3); pj = v3(q[0], q[1], q[2]); }
    else {
      const double th = q[jidxv[i] + 1], cs = cos(th), sn = sin(th);
      const int ax = jkind[i] - MPC_JOINT_RX, b1 = (ax + 1) % 3, b2 = (ax + 2) % 3;
      for (int e = 0; e < 9; ++e) Rj.m[e] = (e % 4 == 0) ? 1.0 : 0.0;
      Rj.m[3 * b1 + b1] = cs; Rj.m[3 * b1 + b2] = -sn; Rj.m[3 * b2 + b1] = sn; Rj.m[3 * b2 + b2] = cs;
    }
    const M3 Rl = mul(Rp, Rj);
    const V3 pl = mul(Rp, pj) + pp;
    for (int e = 0; e < 9; ++e) lR[e * nj + i] = Rl.m[e];
    lp[i] = pl.x; lp[nj + i] = pl.y; lp[2 * nj + i] = pl.z;
  }
  double jdi[13];  // mass, centre of mass, inertia of body `tid` (P4): in flight across the placements
  if (JD_PREFETCH && wv == wv_inertia && lane < nj) {
#pragma unroll
    for (int e = 0; e < 13; ++e) jdi[e] = jd[25 * lane + 12 + e];
  }
  __syncthreads();
  for (int i = tid; i < nj; i += nthr) {
    M3 R = ldcm3(lR, nj, i);
    V3 p = ldcv3(lp, nj, i);
    for (int j = parent[i]; j >= 0; j = parent[j]) {
      const M3 Rj = ldcm3(lR, nj, j);
      p = mul(Rj, p) + ldcv3(lp, nj, j);
      R = mul(Rj, R);
    }
    for (int e = 0; e < 9; ++e) oR[e * nj + i] = R.m[e];
    op[i] = p.x; op[nj + i] = p.y; op[2 * nj + i] = p.z;
  }
  __syncthreads();
  // ---- P2: world-frame joint columns -------------------------------------------------------------------
  for (int kd = tid; kd < nv; kd += nthr) {
    const int i = dof_body[kd], loc = kd - jidxv[i];
    const M3 R = ldcm3(oR, nj, i);
    const V3 p = ldcv3(op, nj, i);
    S6 col;
    if (jkind[i] == MPC_JOINT_FREEFLYER && loc < 3) col = mk6(v3(R.m[loc], R.m[3 + loc], R.m[6 + loc]), v3(0, 0, 0));
    else {
      const int ax = (jkind[i] == MPC_JOINT_FREEFLYER) ? loc - 3 : jkind[i] - MPC_JOINT_RX;
      const V3 w = v3(R.m[ax], R.m[3 + ax], R.m[6 + ax]);
      col = mk6(cross(p, w), w);
    }
    stc6(J, nv, kd, col);
  }
  __syncthreads();
  EV_PROF(0);
  // ---- P3: body velocities ------------------------------------------------------------------------------
  for (int t = wv; t < ((nj + 15) >> 4); t += nw) {  // ov = (J diag(v)) x bits(dmask): 6 x nv times nv x nj
    const d4_t acc = mask_tile([&](int row, int k) { const int r = row < 6 ? row : 5, kc = k < nv ? k : nv - 1; return J[r * nv + kc] * v[kc]; }, dmask, 16 * t, nj, nv, lane);
    tile_store_soa(ov, nj, 0, 6, 16 * t, nj, acc, lane);
  }
  __syncthreads();
  // Contact frames (world placement, placement error of the Baumgarte term, Jlog6): single-lane SE(3) work that needs the placements
  // only — lane cc of wavefront 1 does it here, beside the bodies' inertias on wavefront 0 (after the factor's inputs it sat on the
  // critical path: every other thread waited at the barrier).  The error e6 is parked in gam until the M-tiles phase completes gamma.
  if (has_dyn && wv == 1 && lane < nk) {
    const int cc = lane, cid = ccid_s[cc], i = cbody_s[cc];
    const double* cm = cd + MPC_MODEL_CONTACT_DOUBLES * cid;
    const M3 Ri = ldcm3(oR, nj, i);
    const M3 Rc = mul(Ri, ldm3(cm));
    const V3 pc = mul(Ri, ldv3(cm + 9)) + ldcv3(op, nj, i);
    double* cf = cfr + 54 * cc;
    for (int e = 0; e < 9; ++e) cf[e] = Rc.m[e];
    cf[9] = pc.x; cf[10] = pc.y; cf[11] = pc.z;
    const M3 R2 = ldm3(cm + 12);
    const V3 p2 = ldv3(cm + 21);
    V3 ev, ew;
    log6(tmul(Rc, R2), tmul(Rc, p2 - pc), ev, ew);
    gam[6 * cc] = ev.x; gam[6 * cc + 1] = ev.y; gam[6 * cc + 2] = ev.z; gam[6 * cc + 3] = ew.x; gam[6 * cc + 4] = ew.y; gam[6 * cc + 5] = ew.z;
    if (derivs) Jlog6(tmul(R2, Rc), tmul(R2, pc - p2), cf + 12);  // Jlog6(c2Mc1)
  }
  // ---- P4: bias accelerations (gravity field), spatial inertias (packed symmetric), momenta ---------------
  if (tid < nj) {
    const int i = tid;
    S6 ai = a0;
    for (unsigned long long mm = dmask[i]; mm; mm &= mm - 1) {
      const int kd = __builtin_ctzll(mm);
      ai = add6(ai, scale6(v[kd], mcross(ldc6(ov, nj, dof_body[kd]), ldc6(J, nv, kd))));
    }
    stc6(oa, nj, i, ai);
  }
  if (wv == wv_inertia && lane < nj) {  // (its own wavefront: the chain above and the inertias side by side)
    const int i = lane;
    const M3 R = ldcm3(oR, nj, i);
    const double mass = JD_PREFETCH ? jdi[0] : jd[25 * i + 12];
    const V3 cw = mul(R, JD_PREFETCH ? v3(jdi[1], jdi[2], jdi[3]) : ldv3(jd + 25 * i + 13)) + ldcv3(op, nj, i);
    M3 Ib;
#pragma unroll
    for (int e = 0; e < 9; ++e) Ib.m[e] = JD_PREFETCH ? jdi[4 + e] : jd[25 * i + 16 + e];
    const M3 RI = mul(R, Ib);
    M3 Iww;  // R I R^T
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Iww.m[3 * r + cc] = RI.m[3 * r] * R.m[3 * cc] + RI.m[3 * r + 1] * R.m[3 * cc + 1] + RI.m[3 * r + 2] * R.m[3 * cc + 2];
    const M3 Sx = skew_m(cw), S2 = mul(Sx, Sx);
    double Y[36];
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
      Y[6 * r + cc] = (r == cc) ? mass : 0.0;
      Y[6 * r + cc + 3] = -mass * Sx.m[3 * r + cc];
      Y[6 * (r + 3) + cc] = mass * Sx.m[3 * r + cc];
      Y[6 * (r + 3) + cc + 3] = Iww.m[3 * r + cc] - mass * S2.m[3 * r + cc];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int cc = r; cc < 6; ++cc) oY[sym6(r, cc) * nj + i] = Y[6 * r + cc];
    stc6(oh, nj, i, mat6_mul(Y, ldc6(ov, nj, i)));
  }
  __syncthreads();
  // ---- P5: composite inertias / momenta, bias forces ----------------------------------------------------
  // composite inertias and momenta: Yc = oY x bits(sub), Hc = oh x bits(sub) — tiles over the wavefronts ; the bias forces of the bodies by the lanes of the
  // last wavefront first (its tiles come after)
  if (wv == nw - 1 && lane < nj) stc6(of, nj, lane, add6(sym_mul(ldy21(oY, nj, lane), ldc6(oa, nj, lane)), fcross(ldc6(ov, nj, lane), ldc6(oh, nj, lane))));
  { int t5 = tree_sum_mfma(Yc, nj, oY, nj, 21, sub, nj, nj, wv, nw, 0, lane);
    tree_sum_mfma(Hc, nj, oh, nj, 6, sub, nj, nj, wv, nw, t5, lane); }
  __syncthreads();
  tree_sum_mfma(Fc, nj, of, nj, 6, sub, nj, nj, wv, nw, 0, lane);
  if (wv == nw - 1 && lane < nv) stc6(U, nv, lane, sym_mul(ldy21(Yc, nj, dof_body[lane]), ldc6(J, nv, lane)));  // (nv <= 64: check_multibody_model)
  // total mass and centre of mass from the composite inertia of the root (Yc is recycled before the terms read them)
  const double mtot = Yc[0];
  const V3 com = v3(Yc[sym6(1, 5) * nj] / mtot, Yc[sym6(2, 3) * nj] / mtot, Yc[sym6(0, 4) * nj] / mtot);
  __syncthreads();
  EV_PROF(1);

  // ---- P6: derivative blocks that depend on (q, v) only — formed BEFORE the factorisation, so that the body-level 6 x 6 blocks are
  // dead when M is built (they share one LDS region): Psd, Phi, the body-level "Coriolis" matrices B_i (over Y_i) and their subtree
  // sums, Bt = Bc^T J, Tv, and the products Bc Psd / Yc Psd that the torque derivative and the momentum term need later
  if (derivs) {
    for (int kd = tid; kd < nv; kd += nthr) {
      const int bk = dof_body[kd], pb = parent[bk];
      const S6 vl = (pb >= 0) ? ldc6(ov, nj, pb) : zero6();
      const S6 Jk = ldc6(J, nv, kd);
      const S6 psd = mcross(vl, Jk);
      stc6(Psd, nv, kd, psd);
      stc6(Phi, nv, kd, mcross(add6(ldc6(ov, nj, bk), vl), Jk));
      stc6(YcPsd, nv, kd, sym_mul(ldy21(Yc, nj, bk), psd));
    }
    if (has_dyn || kino) {
      // column `col` of B_i for body i = lane: the columns are dealt to the wavefronts (wavefront-uniform, so that the unit vector stays a
      // compile-time constant and the products with it fold away) — one thread per body did the six columns one after the other, 3 us
      // with a single wavefront at work.  Every thread reads its body's packed inertia before any column is written over it.
      {
        const bool bact = lane < nj;  // (nj <= 64: check_multibody_model)
        const int i = bact ? lane : 0;
        const Y21 Yl = ldy21(oY, nj, i);
        const S6 vi = ldc6(ov, nj, i), hi = ldc6(oh, nj, i);
        __syncthreads();
        auto bcol = [&](auto colc) {
          constexpr int col = decltype(colc)::value;
          S6 e6 = zero6();
          e6.v[col] = 1.0;
          const S6 r = add6(add6(sym_mul(Yl, mcross(e6, vi)), fcross(e6, hi)), fcross(vi, sym_mul(Yl, e6)));
#pragma unroll
          for (int row = 0; row < 6; ++row) oY[(6 * row + col) * nj + i] = r.v[row];
        };
        for (int col = wv; col < 6; col += nw) {
          if (!bact) continue;
          switch (col) {
            case 0: bcol(std::integral_constant<int, 0>{}); break;
            case 1: bcol(std::integral_constant<int, 1>{}); break;
            case 2: bcol(std::integral_constant<int, 2>{}); break;
            case 3: bcol(std::integral_constant<int, 3>{}); break;
            case 4: bcol(std::integral_constant<int, 4>{}); break;
            default: bcol(std::integral_constant<int, 5>{}); break;
          }
        }
      }
      __syncthreads();
      tree_sum_mfma(Bc, nj, oY, nj, 36, sub, nj, nj, wv, nw, 0, lane);
      __syncthreads();
      for (int kd = tid; kd < nv; kd += nthr) {
        const int bk = dof_body[kd];
        const S6 Jk = ldc6(J, nv, kd);
        const G36 Bm = ldg36(Bc, nj, bk);
        stc6(Bt, nv, kd, mat6_tmul(Bm.g, Jk));
        stc6(Tv, nv, kd, add6(sym_mul(ldy21(Yc, nj, bk), ldc6(Phi, nv, kd)), mat6_mul(Bm.g, Jk)));
        stc6(BcPsd, nv, kd, mat6_mul(Bm.g, ldc6(Psd, nv, kd)));
      }
    }
    __syncthreads();
  }
  EV_PROF(7);

  // ---- SE(3)-valued terms: residual and Jacobian block (log map, Jlog6: single-lane work), one term per lane 0 of the wavefronts
  // w0 .. nw - 1.  Depends on the evaluation point only, so on stages with contact dynamics the wavefronts that idle during the
  // one-wavefront factorisation chain (P8) do it there.
  auto se3_prepass = [&](int w0) {
    const int nws = nw - w0;
    if (wv >= w0 && lane == 0) {
      int slot_ = 0;
      for (int t = 0; t < nterms; ++t) {
        const TermRec tr = lds_term(lterm, t);
        const bool se3_state = tr.type == MPC_TERM_STATE_ERROR && tr.i0 < 6;
        if (!se3_state && tr.type != MPC_TERM_FRAME_PLACEMENT) continue;
        const int my = slot_++;
        if (my >= MB_SE3_SLOTS || (my % nws) != wv - w0) continue;
        const double* tp = P + tr.poff;
        double* sl = se3 + 48 * my;
        V3 ev, ew;
        if (se3_state) {
          // r = x_ref (-) x on the base ; J = -Jlog6(Mref^-1 M)
          const M3 Rr = quat_to_rot(tp + 3), Rb = quat_to_rot(q + 3);
          const V3 pr = v3(tp[0], tp[1], tp[2]), pb = v3(q[0], q[1], q[2]);
          log6(tmul(Rb, Rr), tmul(Rb, pr - pb), ev, ew);
          if (derivs) { Jlog6(tmul(Rr, Rb), tmul(Rr, pb - pr), sl + 8); for (int e = 0; e < 36; ++e) sl[8 + e] = -sl[8 + e]; }
        } else {
          const int fi = tr.i0, i = mframe[fi];
          const M3 Ri = ldcm3(oR, nj, i);
          const M3 Rf = mul(Ri, ldm3(fd + 12 * fi));
          const V3 pf = mul(Ri, ldv3(fd + 12 * fi + 9)) + ldcv3(op, nj, i);
          const M3 Rr = ldm3(tp);
          const V3 pr = ldv3(tp + 9);
          log6(tmul(Rr, Rf), tmul(Rr, pf - pr), ev, ew);
          if (derivs) Jlog6(tmul(Rr, Rf), tmul(Rr, pf - pr), sl + 8);
        }
        sl[0] = ev.x; sl[1] = ev.y; sl[2] = ev.z; sl[3] = ew.x; sl[4] = ew.y; sl[5] = ew.z;
      }
    }
  };
  // Selector constraints (joint limits fulldynamic_talos.py:208-209, torque box :206-207): rows of +-1 straight into the record, values
  // from x / u alone.  Like the SE(3) pre-pass they do not depend on the solve, so on stages with contact dynamics the wavefronts
  // w0 .. nw - 1 write them while wavefront 0 runs the factorisation chain (64 rows x nz doubles of stores off the critical path).
  auto selector_rows = [&](int w0) {
    if (wv < w0) return;
    const int nws = nw - w0, t0_ = tid - 64 * w0, nt_ = 64 * nws;
    int row = 0;
    for (int t = 0; t < nterms; ++t) {
      const TermRec tr = lds_term(lterm, t);
      if (tr.role == MPC_ROLE_COST) continue;
      const int d = tr.dim;
      if ((tr.type == MPC_TERM_STATE_ERROR && tr.i0 >= 6) || tr.type == MPC_TERM_CONTROL_ERROR) {
        const double* tp = P + tr.poff;
        const bool st_ = tr.type == MPC_TERM_STATE_ERROR;
        for (int i = t0_; i < d; i += nt_) {
          const int ri = tr.i0 + i;
          kn[KL.oCV + row + i] = st_ ? ((ri < nv) ? (tp[ri + 1] - q[ri + 1]) : (tp[nq + ri - nv] - v[ri - nv])) : (u[ri] - tp[ri]);
          kn[KL.oCT + row + i] = (double)tr.role;
          kn[KL.oLO + row + i] = (tr.role == MPC_ROLE_BOX) ? P[tr.woff + i] : 0.0;
          kn[KL.oHI + row + i] = (tr.role == MPC_ROLE_BOX) ? P[tr.woff + d + i] : 0.0;
        }
        const int zc0 = st_ ? tr.i0 : n + tr.i0;
        const double sgn = st_ ? -1.0 : 1.0;
        if (derivs) for (int i = wv - w0; i < d; i += nws) for (int z = lane; z < nz; z += 64) kn[KL.oCD + (size_t)(row + i) * KL.nz + z] = (z == zc0 + i) ? sgn : 0.0;
      }
      row += d;
    }
  };

  // classification of the term table (one lane; on stages with contact dynamics beside the factorisation chain)
  unsigned char* tkind = (unsigned char*)(early + 2 * nz + 62);  // 0 workgroup pass, 1 stacked cost (wave pass), 2 dense-weight cost (HBM pass)
  unsigned char* trow = tkind + 24;    // first stack row of a stacked term
  unsigned char* tse3 = tkind + 48;    // slot in the SE(3) table
  unsigned char* tchunk = tkind + 72;  // stack chunk (32 rows each)
  unsigned char* tmeta = tkind + 96;   // [0] number of chunks, [1] any dense-weight cost
  auto classify_terms = [&]() {
    int rows = 0, chunk = 0, se3n = 0, dense = 0, nst = 0;
    for (int t = 0; t < nterms; ++t) {
      const TermRec tr = lds_term(lterm, t);
      const bool se3t = tr.type == MPC_TERM_FRAME_PLACEMENT || (tr.type == MPC_TERM_STATE_ERROR && tr.i0 < 6);
      tse3[t] = (unsigned char)(se3t ? se3n++ : 0);
      const bool diag_sel = (tr.type == MPC_TERM_STATE_ERROR || tr.type == MPC_TERM_CONTROL_ERROR) && (tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT);
      int kind = 0;
      if (tr.role == MPC_ROLE_COST && !diag_sel) {
        bool wdiag = (tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT) != 0;
        if (!wdiag) { const double* W = P + tr.woff; wdiag = true; for (int e = 0; e < tr.dim * tr.dim; ++e) if ((e / tr.dim != e % tr.dim) && W[e] != 0.0) wdiag = false; }
        if (wdiag && tr.dim <= 24) {
          kind = 1;
          if (rows + tr.dim > 32) { ++chunk; rows = 0; }
          trow[t] = (unsigned char)rows; tchunk[t] = (unsigned char)chunk; rows += tr.dim; ++nst;
        } else { kind = 2; dense = 1; }
      }
      tkind[t] = (unsigned char)kind;
    }
    tmeta[0] = (unsigned char)(nst ? chunk + 1 : 0); tmeta[1] = (unsigned char)dense;
  };
  // Cost terms on the state / control error with diagonal weights (fulldynamic_talos.py:176-177: the first two terms of every stage):
  // value, gradient and Hessian-diagonal contributions need x, u and the SE(3) table only, so wavefront w0 accumulates them beside the
  // factorisation chain too — in term order into their own accumulators, from which pass B starts (same sums in the same order as
  // accumulating them there).  Only if the SE(3) slots they read were filled by this very wavefront (its LDS operations execute in order).
  auto early_costs = [&](int w0) {
    if (wv != w0) return;
    const int nws = nw - w0;
    double *eg = early, *eh = early + nz, *eb = early + 2 * nz, *ec = early + 2 * nz + 36;
    bool ok = true;
    { int slot_ = 0;
      for (int t = 0; t < nterms; ++t) {
        const TermRec tr = lds_term(lterm, t);
        const bool se3_state = tr.type == MPC_TERM_STATE_ERROR && tr.i0 < 6;
        if (!se3_state && tr.type != MPC_TERM_FRAME_PLACEMENT) continue;
        const int my = slot_++;
        if (se3_state && tr.role == MPC_ROLE_COST && (tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT) && (my >= MB_SE3_SLOTS || (my % nws) != 0)) ok = false;
      } }
    if (!ok) { if (lane == 0) early[2 * nz + 60] = 0.0; return; }
    for (int z = lane; z < nz; z += 64) { eg[z] = 0.0; eh[z] = a.opt.reg_init; }
    if (lane < 36) eb[lane] = 0.0;
    if (lane < 24) ec[lane] = 0.0;
    int slot_ = 0;
    for (int t = 0; t < nterms; ++t) {
      const TermRec tr = lds_term(lterm, t);
      const bool se3t = tr.type == MPC_TERM_FRAME_PLACEMENT || (tr.type == MPC_TERM_STATE_ERROR && tr.i0 < 6);
      const int my = se3t ? slot_++ : 0;
      if (tr.role != MPC_ROLE_COST || !(tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT) || t >= 24) continue;
      const double* tp = P + tr.poff;
      const double* W = P + tr.woff;
      const double* sl = se3 + 48 * my;
      const int d = tr.dim;
      if (tr.type == MPC_TERM_STATE_ERROR) {
        const double* Jb = sl + 8;
        double cst = 0;
        for (int i = lane; i < d; i += 64) {
          const int ri = tr.i0 + i;
          const double e = (ri < 6) ? sl[ri] : ((ri < nv) ? (tp[ri + 1] - q[ri + 1]) : (tp[nq + ri - nv] - v[ri - nv]));
          cst += W[i] * e * e;
        }
        cst = wave_sum(cst);
        if (lane == 0) ec[t] = 0.5 * cst;
        if (derivs) {
          for (int z = lane; z < n; z += 64) {
            double g = 0;
            if (z < 6) { for (int i = 0; i < d && tr.i0 + i < 6; ++i) { const int ri = tr.i0 + i; g += Jb[6 * ri + z] * W[i] * sl[ri]; } }
            else if (z >= tr.i0 && z < tr.i0 + d) {
              const double e = (z < nv) ? (tp[z + 1] - q[z + 1]) : (tp[nq + z - nv] - v[z - nv]);
              g = -W[z - tr.i0] * e; eh[z] += W[z - tr.i0];
            }
            eg[z] += g;
          }
          if (lane < 36) {
            const int za = lane / 6, zb = lane % 6;
            double h = 0;
            for (int i = 0; i < d && tr.i0 + i < 6; ++i) { const int ri = tr.i0 + i; h += Jb[6 * ri + za] * W[i] * Jb[6 * ri + zb]; }
            eb[lane] += h;
          }
        }
      } else if (tr.type == MPC_TERM_CONTROL_ERROR) {
        double cst = 0;
        for (int i = lane; i < d; i += 64) { const double e = u[tr.i0 + i] - tp[tr.i0 + i]; cst += W[i] * e * e; }
        cst = wave_sum(cst);
        if (lane == 0) ec[t] = 0.5 * cst;
        if (derivs) for (int i = lane; i < d; i += 64) {
          const int z = n + tr.i0 + i;
          eg[z] += W[i] * (u[tr.i0 + i] - tp[tr.i0 + i]);
          eh[z] += W[i];
        }
      }
    }
    if (lane == 0) early[2 * nz + 60] = 1.0;
  };

  if (has_dyn) {
    // ---- P7: joint-space inertia (lower block triangle, tile-packed), bias torques, contact frames ---------------
    const int ntile = nbm * (nbm + 1) / 2;
    // a tile per wavefront: U^T J (entry (r, c) with c on the path to r) and J^T U (r on the path to c) as two depth-6 products on the matrix cores, the tree
    // masks pick one — 8 operand reads and 4 MFMAs for 256 entries where a lane per entry read 12 doubles for each of its six
    for (int t = wv; t < ntile; t += nw) {
      int bi = 0;
      while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
      const int bj = t - bi * (bi + 1) / 2, kk4 = lane >> 4;
      const int ra = 16 * bi + (lane & 15), rc = ra < nv ? ra : nv - 1, ca = 16 * bj + (lane & 15), cl = ca < nv ? ca : nv - 1;
      const double a0 = U[kk4 * nv + rc], a1 = (kk4 < 2) ? U[(4 + kk4) * nv + rc] : 0.0, b0 = J[kk4 * nv + cl], b1 = (kk4 < 2) ? J[(4 + kk4) * nv + cl] : 0.0;
      const double e0 = J[kk4 * nv + rc], e1 = (kk4 < 2) ? J[(4 + kk4) * nv + rc] : 0.0, f0 = U[kk4 * nv + cl], f1 = (kk4 < 2) ? U[(4 + kk4) * nv + cl] : 0.0;
      d4_t p1 = d4_t{0, 0, 0, 0}, p2 = d4_t{0, 0, 0, 0};
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e0, f0, p2, 0, 0, 0);
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e1, f1, p2, 0, 0, 0);
      const int j = lane & 15, cc = 16 * bj + j;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = (lane >> 4) + 4 * q, r = 16 * bi + i;
        double s = (r == cc) ? 1.0 : 0.0;  // identity padding
        if (r < nv && cc < nv) {
          s = 0;
          if (BELOW(cc, dof_body[r])) s = p1[q];
          else if (BELOW(r, dof_body[cc])) s = p2[q];
        }
        Mt[t * 272 + i * 17 + j] = s;
      }
    }
    for (int kd = tid; kd < nv; kd += nthr) bias[kd] = dot6(ldc6(J, nv, kd), ldc6(Fc, nj, dof_body[kd]));
    if (tid < nk) {
      // (the frame, its placement error and Jlog6 were left in cfr / gam by contact_frames(), beside P4)
      const int cid = ccid_s[tid], i = cbody_s[tid];
      const double* cm = cd + MPC_MODEL_CONTACT_DOUBLES * cid;
      const M3 Rc = ldm3(cfr + 54 * tid);
      const V3 pc = ldv3(cfr + 54 * tid + 9);
      const S6 acb = adinv(Rc, pc, sub6(ldc6(oa, nj, i), a0));
      const S6 vcb = adinv(Rc, pc, ldc6(ov, nj, i));
      for (int r = 0; r < 6; ++r) gam[6 * tid + r] = acb.v[r] + cm[30 + r] * vcb.v[r] - cm[24 + r] * gam[6 * tid + r];
    }
    __syncthreads();
    // Y16 = [Jc^T | r1 | 0]  (nvp x 16): the contact columns (LOCAL frame: Ad(M_c)^-1 J) and the dynamics right-hand side r1 = B u - bias
    for (int idx = tid; idx < nvp * 16; idx += nthr) {
      const int l = idx >> 4, j = idx & 15;
      double s = 0.0;
      if (l < nv) {
        if (j < nl) {
          const int cc = j / 6;
          if (BELOW(l, cbody_s[cc])) s = adinv(ldm3(cfr + 54 * cc), ldv3(cfr + 54 * cc + 9), ldc6(J, nv, l)).v[j - 6 * cc];
        } else if (j == 12) {
          s = -bias[l] + (l >= nv - nu ? u[l - (nv - nu)] : 0.0);
          // disturbance of the simulation stand-in: a world-frame force f at the base origin acts on the linear base dofs only
          // (J_l . [f ; p x f] = (R e_l) . f ; the angular dofs cancel)
          if (TRIAL == 2 && mb.f_ext && l < 3) s += J[0 * nv + l] * mb.f_ext[3 * b] + J[1 * nv + l] * mb.f_ext[3 * b + 1] + J[2 * nv + l] * mb.f_ext[3 * b + 2];
        }
      }
      Y16[l * MB_LDY + j] = s;
    }
    if (tid == 0) { iflag[0] = 1; iflag[1] = 1; }
    __syncthreads();
    EV_PROF(2);
    // ---- P8: M = L L^T ; Y = L^-1 [Jc^T | r1] ; S = Y^T Y + mu I = Ls Ls^T ; multipliers — ONE wavefront, no barrier inside (its LDS
    // operations execute in order) ; then the accelerations
    if (wv == 0) {
      constexpr int NLAST_M = FX ? ((FV % 16) ? (FV % 16) : 16) : 16;  // real columns of the last diagonal block of M (fixed dimensions: known ; else: all 16)
      if (!chol_tiles_wave_last<NLAST_M>(Mt, nbm, lane)) { if (lane == 0) iflag[0] = 0; }
      else {
        EV_SUB(3);
        trsm_fwd_tiles(Mt, nbm, Y16, MB_LDY, 1, 0, 1, lane);
        EV_SUB(4);
        d4_t g = d4_t{0, 0, 0, 0};
        mma_tile<false>(g, Y16, 1, MB_LDY, Y16, MB_LDY, 1, nvp, lane);  // [Y w]^T [Y w]
        const int col = lane & 15;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int row = (lane >> 4) + 4 * qq;
          double sv = (row < nl && col < nl) ? g[qq] : 0.0;
          if (row == col) sv += (row < nl) ? prox_mu : 1.0;
          Sp[row * 17 + col] = sv;
          if (col == 12 && row < nl) small[row] = g[qq] + gam[row];  // t = Y^T w - r2,  r2 = -gamma
        }
        const bool s_ok = (nl == 12) ? chol16_wave_n<12>(Sp, 17, LIs, lane) : ((nl == 6) ? chol16_wave_n<6>(Sp, 17, LIs, lane) : chol16_wave(Sp, 17, LIs, lane));
        if (!s_ok && lane == 0) iflag[1] = 0;
        EV_SUB(16);
        // z2 = Ls^-T Ls^-1 t ; lambda = -z2
        // (sixteen terms for every lane, all operands requested at once: the entries of Ls^-1 above the diagonal are exact zeros, the
        // sums keep their order and their bits ; with the trip count depending on the lane every term waited for its own LDS round trip)
        if (lane < 16) {
          double y = 0;
#pragma unroll
          for (int j = 0; j < 16; ++j) y += LIs[lane * 17 + j] * ((j < nl) ? small[j] : 0.0);
          small[16 + lane] = y;
        }
        if (lane < 16) {
          double z = 0;
#pragma unroll
          for (int j = 0; j < 16; ++j) z += LIs[j * 17 + lane] * small[16 + j];
          small[32 + lane] = z;
          if (lane < nl) lam[lane] = -z;
        }
        EV_SUB(17);
        // V16 column 0 = w - Y z2, then accelerations = L^-T (.)
        // (a lane per ROW: with a lane per entry only the four lanes of column 0 worked, twelve rounds of a dependent nl-term sum)
        for (int idx = lane; idx < nvp * 16; idx += 64) if (idx & 15) V16[idx] = 0.0;
        for (int l = lane; l < nvp; l += 64) {
          double s = Y16[l * MB_LDY + 12];
          for (int i = 0; i < nl; ++i) s -= Y16[l * MB_LDY + i] * small[32 + i];
          V16[l * 16] = s;
        }
        EV_SUB(18);
        trsm_bwd_tiles(Mt, nbm, V16, 16, 1, 0, 1, lane);
        for (int i = lane; i < nvp; i += 64) { acc[i] = V16[i * 16]; Y16[i * MB_LDY + 12] = 0.0; }  // from here on Y16 = Y (zero padded)
        EV_SUB(19);
      }
    } else { se3_prepass(1); selector_rows(1); early_costs(1); if (wv == nw - 1 && lane == 0) classify_terms(); }
    __syncthreads();
    if (iflag[0] == 0 || iflag[1] == 0) { if (tid == 0) a.inst[b].done = iflag[0] == 0 ? 5 : 6; return; }
    EV_PROF(5);
    if (derivs) {
      for (int i = tid; i < n; i += nthr) kn[KL.oXD + i] = (i < nv) ? v[i] : acc[i - nv];
      for (int i = tid; i < 12; i += nthr) {
        double wr = 0.0;
        for (int cc = 0; cc < nk; ++cc) if (ccid_s[cc] == i / 6) wr = lam[6 * cc + i % 6];
        kn[KL.oWR + i] = wr;
      }
      // world-frame wrench of every contact (at the origin), for the forces at the solution
      if (tid < nk) {
        const M3 Rc = ldm3(cfr + 54 * tid);
        const V3 pc = ldv3(cfr + 54 * tid + 9);
        const V3 fl = mul(Rc, v3(lam[6 * tid], lam[6 * tid + 1], lam[6 * tid + 2]));
        const V3 fa = mul(Rc, v3(lam[6 * tid + 3], lam[6 * tid + 4], lam[6 * tid + 5])) + cross(pc, fl);
        st6(cfr + 54 * tid + 48, mk6(fl, fa));
      }
    }
  }

  EV_PROF(6);
  if (TRIAL == 2) {
    // semi-implicit Euler step of length sim_dt (same scheme as the stage dynamics), then the next sub-step
    const double dts = mb.sim_dt;
    double* xnew = Tv;
    if (tid == 0) {
      const V3 dl = v3(dts * (v[0] + dts * acc[0]), dts * (v[1] + dts * acc[1]), dts * (v[2] + dts * acc[2]));
      const V3 da_ = v3(dts * (v[3] + dts * acc[3]), dts * (v[4] + dts * acc[4]), dts * (v[5] + dts * acc[5]));
      M3 dR; V3 dp;
      exp6(dl, da_, dR, dp);
      const M3 Rb = quat_to_rot(q + 3);
      const M3 Rn = mul(Rb, dR);
      const V3 pn = mul(Rb, dp) + v3(q[0], q[1], q[2]);
      xnew[0] = pn.x; xnew[1] = pn.y; xnew[2] = pn.z;
      rot_to_quat(Rn, xnew + 3);
    }
    for (int i = tid; i < nv; i += nthr) {
      const double vp = v[i] + dts * acc[i];
      xnew[nq + i] = vp;
      if (i >= 6) xnew[i + 1] = q[i + 1] + dts * vp;
    }
    __syncthreads();
    for (int i = tid; i < nx; i += nthr) x[i] = xnew[i];
    __syncthreads();
    if (++sim_sub < mb.sim_substeps) goto sim_loop;
    for (int i = tid; i < nx; i += nthr) a.x0[(size_t)b * nx + i] = x[i];  // the measured state of the next tick
    if (mb.sim_wrench)
      for (int i = tid; i < 12; i += nthr) {
        double wr = 0.0;
        for (int cc = 0; cc < nk; ++cc) if (ccid_s[cc] == i / 6) wr = lam[6 * cc + i % 6];
        mb.sim_wrench[(size_t)b * 12 + i] = wr;
      }
    return;
  }
  // ---- kinodynamics (kinodynamic_talos.py:107-112): a_joint = u[12:], base acceleration from the momentum balance
  // about the world origin  sum_k U_k a_k + hdot(a = 0) = [sum f + m g ; sum p_i x f_i + tau_i + c x m g],
  // projected on the base columns: (J_b^T U_b) a_b = J_b^T (...)  with J_b^T U_b = M_bb symmetric positive definite.
  if (kino) {
    const double* dp = P + desc[4];
    const int nkk = desc[1], nf = 6 * nkk;
    const V3 mg = v3(mtot * dp[1], mtot * dp[2], mtot * dp[3]);
    double* Minv6 = small + 144;  // 36
    if (tid < nkk) {
      const int fi = (int)dp[4 + tid], i = mframe[fi];
      const V3 pf = mul(ldcm3(oR, nj, i), ldv3(fd + 12 * fi + 9)) + ldcv3(op, nj, i);
      cfr[54 * tid + 9] = pf.x; cfr[54 * tid + 10] = pf.y; cfr[54 * tid + 11] = pf.z;
      cfr[54 * tid] = (double)i;
    }
    for (int i = tid; i < nv; i += nthr) acc[i] = (i >= 6) ? u[nf + i - 6] : 0.0;
    __syncthreads();
    if (tid == 0) {
      S6 r0 = mk6(mg, cross(com, mg));
      for (int cc = 0; cc < nkk; ++cc) {
        if (!desc[2 + cc]) continue;
        const V3 pf = ldv3(cfr + 54 * cc + 9);
        const V3 f = v3(u[6 * cc], u[6 * cc + 1], u[6 * cc + 2]), tq = v3(u[6 * cc + 3], u[6 * cc + 4], u[6 * cc + 5]);
        r0 = add6(r0, mk6(f, cross(pf, f) + tq));
      }
      r0 = sub6(r0, ldc6(Fc, nj, 0));  // hdot at a = 0 (true accelerations: a0 = 0 in this mode)
      for (int j = 6; j < nv; ++j) r0 = sub6(r0, scale6(acc[j], ldc6(U, nv, j)));
      double Mbb[36], rb[6];
      for (int r = 0; r < 6; ++r) {
        rb[r] = dot6(ldc6(J, nv, r), r0);
        for (int cc = 0; cc < 6; ++cc) Mbb[6 * r + cc] = dot6(ldc6(U, nv, r), ldc6(J, nv, cc));
      }
      inv6_unrolled_mb(Mbb, Minv6);
      for (int r = 0; r < 6; ++r) { double sacc = 0; for (int cc = 0; cc < 6; ++cc) sacc += Minv6[6 * r + cc] * rb[cc]; acc[r] = sacc; }
    }
    __syncthreads();
    if (derivs) {
      for (int i = tid; i < n; i += nthr) kn[KL.oXD + i] = (i < nv) ? v[i] : acc[i - nv];
      for (int i = tid; i < 12; i += nthr) kn[KL.oWR + i] = 0.0;
    }
  }

  // single-lane SE(3) work of the integrator (P12): step, gap and its Jacobian blocks Jlog6(G), Jexp6(delta), Ad(exp6(delta))^-1, E6 = -Jlog6(G^-1).  Depends on the
  // accelerations only: on stages with contact dynamics it runs on the wavefront that has one column block less in P11 (7 blocks on 4 wavefronts), beside
  // the implicit differentiation instead of after it.
  const double dt_se3 = (has_dyn || kino) ? P[desc[4]] : 0.0;
  const bool direct_ab = has_dyn && 6 * nz <= 21 * nj;  // the base rows of d a fit in the region of Yc (every Talos model ; else: the HBM scratch)
  // ONE lane, ONE logarithm (round 6).  Rounds 3 - 5 split this over two lanes whose parts each began with exp6 and ended in a Jlog6 that recomputed the
  // logarithm just taken — 33 calls of sin / cos / atan2 between them, 13 us on the two diverged lanes of a wavefront the other three waited for at the end of
  // P11.  Here: exp6 once ; (gv, gw) = log6(G) once ; log6(G^-1) = -(gv, gw) exactly, so Jlog6(G^-1) (E6) takes the same series coefficients with the signs of
  // its odd terms flipped ; Jexp6 shares so3_coeffs with exp6.  9 calls.
  // mode 0: everything ; 1: everything but Jexp6(delta) / Ad^-1 ; 2: only those two (they need the step alone: EV_SE3_X1 runs them on a wavefront that idles in P10)
  auto step_se3_fused = [&](int mode = 0) {
    const double dt = dt_se3;
    double* Jl6 = small; double* Je6 = small + 36; double* Jq6 = small + 72;
    const V3 dl = v3(dt * (v[0] + dt * acc[0]), dt * (v[1] + dt * acc[1]), dt * (v[2] + dt * acc[2]));
    const V3 da_ = v3(dt * (v[3] + dt * acc[3]), dt * (v[4] + dt * acc[4]), dt * (v[5] + dt * acc[5]));
    const double td2 = dot(da_, da_);
    double A, B, C;
    so3_coeffs(td2, A, B, C);
    const M3 Kd = skew_m(da_), Kd2 = mul(Kd, Kd);
    M3 dR;
    for (int i = 0; i < 9; ++i) dR.m[i] = ((i % 4 == 0) ? 1.0 : 0.0) + A * Kd.m[i] + B * Kd2.m[i];
    const V3 wv_ = cross(da_, dl);
    const V3 dp = dl + B * wv_ + C * cross(da_, wv_);
    if (mode != 1 && derivs) {
      // Jexp6(delta)
      double b1, b2, b3;
      q_coeffs(td2, b1, b2, b3);
      const M3 Qe = Qmat_c(v3(-dl.x, -dl.y, -dl.z), v3(-da_.x, -da_.y, -da_.z), b1, b2, b3);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        const double jr = ((i == j) ? 1.0 : 0.0) - B * Kd.m[3 * i + j] + C * Kd2.m[3 * i + j];
        Je6[6 * i + j] = jr; Je6[6 * (i + 3) + j + 3] = jr; Je6[6 * i + j + 3] = Qe.m[3 * i + j]; Je6[6 * (i + 3) + j] = 0.0;
      }
      // Ad(exp6(delta))^-1 = [[dR^T, -dR^T [dp]x],[0, dR^T]]
      const M3 Sx = skew_m(dp);
      const M3 RtS = tmul(dR, Sx);
      for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
        Jq6[6 * r + cc] = dR.m[3 * cc + r]; Jq6[6 * (r + 3) + cc + 3] = dR.m[3 * cc + r];
        Jq6[6 * r + cc + 3] = -RtS.m[3 * r + cc]; Jq6[6 * (r + 3) + cc] = 0.0;
      }
    }
    if (mode == 2) return;
    const M3 Rb = quat_to_rot(q + 3);
    const M3 Rn = mul(Rb, dR);
    const V3 pn = mul(Rb, dp) + v3(q[0], q[1], q[2]);
    const M3 Rt = quat_to_rot(xn + 3);
    const M3 GR = tmul(Rt, Rn);
    const V3 Gp = tmul(Rt, pn - v3(xn[0], xn[1], xn[2]));
    if (derivs) { kn[KL.oXN] = pn.x; kn[KL.oXN + 1] = pn.y; kn[KL.oXN + 2] = pn.z; rot_to_quat(Rn, kn + KL.oXN + 3); }
    const V3 gw = log3(GR);
    const double t2 = dot(gw, gw);
    double c;
    if (t2 < kSmall2) c = 1.0 / 12 + t2 * (1.0 / 720 + t2 * (1.0 / 30240 + t2 * (1.0 / 1209600)));
    else { const double t = sqrt(t2), sh = sin(0.5 * t), ch = cos(0.5 * t); c = (1.0 - t * ch / (2.0 * sh)) / t2; }
    const V3 wp = cross(gw, Gp);
    const V3 gv = Gp - 0.5 * wp + c * cross(gw, wp);
    kn[KL.oF] = gv.x; kn[KL.oF + 1] = gv.y; kn[KL.oF + 2] = gv.z; kn[KL.oF + 3] = gw.x; kn[KL.oF + 4] = gw.y; kn[KL.oF + 5] = gw.z;
    if (!derivs) return;
    double a1, a2, a3;
    q_coeffs(t2, a1, a2, a3);
    const M3 K = skew_m(gw), K2 = mul(K, K);
    M3 Ji, Jn;
    for (int i = 0; i < 9; ++i) { const double e = ((i % 4 == 0) ? 1.0 : 0.0) + c * K2.m[i]; Ji.m[i] = e + 0.5 * K.m[i]; Jn.m[i] = e - 0.5 * K.m[i]; }
    jlog6_blocks(Ji, Qmat_c(v3(-gv.x, -gv.y, -gv.z), v3(-gw.x, -gw.y, -gw.z), a1, a2, a3), 1.0, Jl6);   // Jlog6(G)
    double E[36];
    jlog6_blocks(Jn, Qmat_c(gv, gw, a1, a2, a3), -1.0, E);                                              // E6 = -Jlog6(G^-1)
    for (int e = 0; e < 36; ++e) kn[KL.oE6 + e] = E[e];
    {
      // T6 = (-E6)^-1, the base-frame change the Riccati sweep applies to the co-state: -E6 = Jlog6(G^-1) and Jlog6(M)^-1 = Jexp6(log6(M)), so T6 = Jexp6(-gv, -gw) in closed
      // form — its Q block is the one E6 just used.  (The sweep inverted -E6 numerically on ONE lane per knot, 511 threads waiting at a barrier.)
      double A6, B6, C6;
      so3_coeffs(t2, A6, B6, C6);
      const M3 Q6 = Qmat_c(gv, gw, a1, a2, a3);
      double T6o[36];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        const double jr = ((i == j) ? 1.0 : 0.0) + B6 * K.m[3 * i + j] + C6 * K2.m[3 * i + j];
        T6o[6 * i + j] = jr; T6o[6 * (i + 3) + j + 3] = jr; T6o[6 * i + j + 3] = Q6.m[3 * i + j]; T6o[6 * (i + 3) + j] = 0.0;
      }
      for (int e = 0; e < 36; ++e) kn[KL.oT6k + e] = T6o[e];
    }
  };
  // ---- P9: body accelerations and subtree forces AT THE SOLUTION (only the derivative blocks read them):
  //   da_i = sum_{k on the path to i} J_k acc_k ;  Fc_i += Yc_i da_i + sum_{k strictly below i} U_k acc_k - (wrenches of the contacts below i)
  // (sum over the subtree of Y_j da_j, regrouped by dof: the composite inertias and U = Yc J are at hand, the body inertias are not)
  if (derivs && (has_dyn || kino)) {
    double* da = Tq;  // scratch [6][nj] (Tq is formed afterwards)
    // da = (J diag(acc)) x bits(dmask) and the "dofs strictly below" part of the forces, (U diag(acc)) x bits(below), as tiles: wavefront t does column tile t of
    // both (the second into its registers), then the rest of the update on its own columns after the barrier
    d4_t fbel = d4_t{0, 0, 0, 0};
    const int ctn = (nj + 15) >> 4;
    if (wv < ctn) {
      const d4_t acc_ = mask_tile([&](int row, int k) { const int r = row < 6 ? row : 5, kc = k < nv ? k : nv - 1; return J[r * nv + kc] * acc[kc]; }, dmask, 16 * wv, nj, nv, lane);
      tile_store_soa(da, nj, 0, 6, 16 * wv, nj, acc_, lane);
      fbel = mask_tile([&](int row, int k) { const int r = row < 6 ? row : 5, kc = k < nv ? k : nv - 1; return U[r * nv + kc] * acc[kc]; }, below, 16 * wv, nj, nv, lane);
    }
    for (int t = nw + wv; t < ctn; t += nw) {  // (models with more than 16 nw bodies: not the Talos — the plain form for the remaining columns)
      for (int idx = lane; idx < 6 * 16; idx += 64) { const int e = idx >> 4, i = 16 * t + (idx & 15); if (i < nj) da[e * nj + i] = mask_sum(dmask[i], [&](int kd) { return J[e * nv + kd] * acc[kd]; }); }
    }
    __syncthreads();
    for (int t = wv; t < ctn; t += nw) {
      const int i = 16 * t + (lane & 15);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = (lane >> 4) + 4 * q;
        if (e < 6 && i < nj) {
          const int idx = e * nj + i;
          double s = Fc[idx];
#pragma unroll
          for (int bb = 0; bb < 6; ++bb) s += Yc[sym6(e, bb) * nj + i] * da[bb * nj + i];
          s += (t < nw) ? fbel[q] : mask_sum(below[i], [&](int kd) { return U[e * nv + kd] * acc[kd]; });
          for (int cc = 0; cc < nk; ++cc) if ((anc[cbody_s[cc]] >> i) & 1ull) s -= cfr[54 * cc + 48 + e];
          Fc[idx] = s;
          oa[idx] += da[idx];
        }
      }
    }
    __syncthreads();
    // Psdd (needs the parents' accelerations at the solution) and Tq = Yc Psdd + Bc Psd + J x* Fc
    for (int kd = tid; kd < nv; kd += nthr) {
      const int bk = dof_body[kd], pb = parent[bk];
      const S6 vl = (pb >= 0) ? ldc6(ov, nj, pb) : zero6();
      const S6 al = (pb >= 0) ? ldc6(oa, nj, pb) : a0;
      const S6 Jk = ldc6(J, nv, kd);
      const S6 psdd = add6(mcross(al, Jk), mcross(vl, ldc6(Psd, nv, kd)));
      stc6(Psdd, nv, kd, psdd);
      stc6(Tq, nv, kd, add6(add6(sym_mul(ldy21(Yc, nj, bk), psdd), ldc6(BcPsd, nv, kd)), fcross(Jk, ldc6(Fc, nj, bk))));
    }
    __syncthreads();
    if (kino) {
      // d r0 / d(q, v, u) (6 x nz), then  d a_b = -Mbb^-1 J_b^T d r0 ; joint accelerations are controls
      const double* dp = P + desc[4];
      const int nkk = desc[1], nf = 6 * nkk;
      const V3 mg = v3(mtot * dp[1], mtot * dp[2], mtot * dp[3]);
      const double* Minv6 = small + 144;
      for (int z = tid; z < nz; z += nthr) {
        S6 col = zero6();
        if (z < nv) {
          col = ldc6(Tq, nv, z);
          V3 dang = cross((1.0 / mtot) * lin(ldc6(U, nv, z)), mg);
          const S6 Jz = ldc6(J, nv, z);
          for (int cc = 0; cc < nkk; ++cc) {
            if (!desc[2 + cc] || !BELOW(z, (int)cfr[54 * cc])) continue;
            const V3 pf = ldv3(cfr + 54 * cc + 9);
            dang = dang + cross(lin(Jz) + cross(ang(Jz), pf), v3(u[6 * cc], u[6 * cc + 1], u[6 * cc + 2]));
          }
          col = sub6(col, mk6(v3(0, 0, 0), dang));
        } else if (z < n) {
          col = ldc6(Tv, nv, z - nv);
        } else if (z < n + nf) {
          const int cc = (z - n) / 6, e = (z - n) % 6;
          if (desc[2 + cc]) {
            V3 ev = v3(e % 3 == 0 ? 1.0 : 0.0, e % 3 == 1 ? 1.0 : 0.0, e % 3 == 2 ? 1.0 : 0.0);
            if (e < 3) col = mk6(v3(-ev.x, -ev.y, -ev.z), cross(ev, ldv3(cfr + 54 * cc + 9)));  // -[I ; p x]
            else col = mk6(v3(0, 0, 0), v3(-ev.x, -ev.y, -ev.z));
          }
        } else {
          col = ldc6(U, nv, 6 + z - n - nf);
        }
        double jb[6];
        for (int r = 0; r < 6; ++r) jb[r] = dot6(ldc6(J, nv, r), col);
        for (int r = 0; r < 6; ++r) {
          double sacc = 0;
          for (int cc = 0; cc < 6; ++cc) sacc += Minv6[6 * r + cc] * jb[cc];
          dsol[(size_t)r * L.nz + z] = -sacc;
        }
        for (int r = 6; r < nv; ++r) dsol[(size_t)r * L.nz + z] = (z == n + nf + r - 6) ? 1.0 : 0.0;
      }
      __syncthreads();
    }
    EV_PROF(8);
    if (has_dyn) {
      // ---- P10: contact rows R2 = d r2 / d(q, v) into DL (zero padded ; d r2 / du = 0) -------------------------------------
      // (nk nv <= 2 x 64 entries: one per thread ; its contact's gains are requested before the zero fill and the barrier)
      int c10_body = 0;
      double cm10[12];
      if (tid < nk * nv) {
        const int cid = ccid_s[qdiv(tid, S.mg_nv)];
        c10_body = cbody_s[qdiv(tid, S.mg_nv)];
#pragma unroll
        for (int r = 0; r < 12; ++r) cm10[r] = cd[MPC_MODEL_CONTACT_DOUBLES * cid + 24 + r];
      }
      for (int idx = tid; idx < 12 * ldl; idx += nthr) DL[idx] = 0.0;
      __syncthreads();
      if (tid < nk * nv) {
        const int idx = tid;
        const int cc = qdiv(idx, S.mg_nv), j = (idx - qdiv(idx, S.mg_nv) * nv);
        const int i = c10_body;
        if (BELOW(j, i)) {
        const M3 Rc = ldm3(cfr + 54 * cc);
        const V3 pc = ldv3(cfr + 54 * cc + 9);
        const int pb = parent[dof_body[j]];
        const S6 vl = (pb >= 0) ? ldc6(ov, nj, pb) : zero6();
        const S6 al = (pb >= 0) ? ldc6(oa, nj, pb) : a0;
        const S6 Jj = ldc6(J, nv, j), psd = ldc6(Psd, nv, j);
        const S6 w = sub6(ldc6(ov, nj, i), vl);
        const S6 dacq = adinv(Rc, pc, add6(mcross(sub6(al, a0), Jj), mcross(psd, w)));
        const S6 dacv = adinv(Rc, pc, add6(mcross(ldc6(ov, nj, dof_body[j]), Jj), mcross(Jj, w)));
        const S6 apsd = adinv(Rc, pc, psd);
        const S6 Jcj = adinv(Rc, pc, Jj);
        const S6 jl = mat6_mul(cfr + 54 * cc + 12, Jcj);
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          DL[(6 * cc + r) * ldl + j] = dacq.v[r] + cm10[6 + r] * apsd.v[r] + cm10[r] * jl.v[r];
          DL[(6 * cc + r) * ldl + nv + j] = dacv.v[r] + cm10[6 + r] * Jcj.v[r];
        }
        }
      }
      // (the contact rows above occupy nk nv <= 128 threads = two wavefronts: the third does the part of the integrator's single-lane work that needs the step only)
      if (nw > 2 && wv == nw - 2 && lane == 0) step_se3_fused(2);
      __syncthreads();
      double* dbase = Yc;  // [6][nz]: base rows of d a (direct_ab)
      // ---- P11: R1 = d r1 / d(q, v, u) built in registers, one 16-column block per wavefront at a time, and the whole chain of
      // blocked solves on it without touching LDS for the intermediate results (implicit_diff_block)
      const int n2 = 2 * nv;
#ifdef EV_P11PROF  // developer build: where the time of P11 goes, per wavefront (slots 16 + wv: its column blocks ; 3: the SE(3) work of the last one ; 4: R1 of wave 0's first block)
      const long long tp11_ = clock64();
#endif
      // the tree tables of this lane's rows (16 bi + (l >> 4) + 4 qq), once for all its column blocks and unconditionally: inside the entry loop each was a
      // dependent pair of LDS reads behind two branches
      int rbr_[16];
      unsigned long long ranc_[16];
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int r_ = 16 * bi + (lane >> 4) + 4 * qq;
          rbr_[4 * bi + qq] = (bi < nbm) ? dof_body[r_ < nv ? r_ : nv - 1] : 0;
        }
#pragma unroll
      for (int e = 0; e < 16; ++e) ranc_[e] = ((e >> 2) < nbm) ? anc[rbr_[e]] : 0ull;
      for (int cj = wv; cj < ncb; cj += nw) {
        const int z = 16 * cj + (lane & 15), rq = lane >> 4;
        const int kind = z < nv ? 0 : (z < n2 ? 1 : (z < n2 + nu ? 2 : 3));
        const int j = kind == 0 ? z : (kind == 1 ? z - nv : 0);
        const int bj = dof_body[j];
        d4_t w[4], t;
        // Both candidates of every entry as tile products on the matrix cores — [U ; Bt]^T [c1 ; c2] (depth 12: row r below column j) and J^T c3 (depth 6,
        // padded to 8: row r above column j) — then the tree masks pick one.  10 operand reads and 5 MFMAs per 16 x 16 tile where the lane-per-entry form
        // read 4 x (12 .. 24) doubles and issued as many multiply-adds.  Operands: A(row, e) of lane (row = l & 15, e = 4 s + (l >> 4)), B(e, column l & 15).
        {
          const int kk4 = lane >> 4, jc = j;  // (j: the dof of this lane's column, 0 for the control columns — finite operands, masked below)
          const double* c1p = kind ? Phi : Psdd; const double* c2p = kind ? J : Psd; const double* c3p = kind ? Tv : Tq;
          // B operands of the five k-steps (rows e = kk4, 4 + kk4, 8 + kk4 of [c1 ; c2] and e = kk4, 4 + kk4 of [c3 ; 0])
          const double b0 = c1p[kk4 * nv + jc];
          const double b1 = (kk4 < 2) ? c1p[(4 + kk4) * nv + jc] : c2p[(kk4 - 2) * nv + jc];
          const double b2 = c2p[(2 + kk4) * nv + jc];
          const double d0 = c3p[kk4 * nv + jc];
          const double d1 = (kk4 < 2) ? c3p[(4 + kk4) * nv + jc] : 0.0;
          const unsigned long long ancj = anc[bj];
#pragma unroll
          for (int bi = 0; bi < 4; ++bi) {
            w[bi] = d4_t{0, 0, 0, 0};
            if (bi < nbm) {
              const int ra = 16 * bi + (lane & 15), rc = ra < nv ? ra : nv - 1;
              const double a0 = U[kk4 * nv + rc];
              const double a1 = (kk4 < 2) ? U[(4 + kk4) * nv + rc] : Bt[(kk4 - 2) * nv + rc];
              const double a2 = Bt[(2 + kk4) * nv + rc];
              const double e0 = J[kk4 * nv + rc];
              const double e1 = (kk4 < 2) ? J[(4 + kk4) * nv + rc] : 0.0;
              d4_t p1 = d4_t{0, 0, 0, 0}, p2 = d4_t{0, 0, 0, 0};
              p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, p1, 0, 0, 0);
              p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e0, d0, p2, 0, 0, 0);
              p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, p1, 0, 0, 0);
              p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e1, d1, p2, 0, 0, 0);
              p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, p1, 0, 0, 0);
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) {
                const int r = 16 * bi + rq + 4 * qq;
                const bool m1 = (ranc_[4 * bi + qq] >> bj) & 1ull, m2 = (ancj >> rbr_[4 * bi + qq]) & 1ull;
                const double vt = m1 ? p1[qq] : (m2 ? p2[qq] : 0.0);
                const double vu = (kind == 2 && r == nv - nu + (z - n2)) ? -1.0 : 0.0;  // d r1 / du = -B
                w[bi][qq] = (r < nv) ? (kind < 2 ? vt : vu) : 0.0;
              }
            }
          }
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) t[qq] = (qq < 3) ? DL[(rq + 4 * qq) * ldl + 16 * cj + (lane & 15)] : 0.0;
#ifdef EV_P11PROF
        if (TRIAL == 0 && a.prof && k == 1 && tid == 0 && cj == wv) a.prof[(size_t)b * 64 + 32 + 4] += (double)(clock64() - tp11_);
#endif
        implicit_diff_block(w, t, Mt, Y16, LIs, nbm, lane);
        // d a = -Z1: the velocity rows of [A B] (dvp = dt da + [0 I 0]) and the joint-position rows (dt dvp + [I 0 0]) leave for the record
        // straight from the result registers (a fragment is 4 rows x 16 consecutive columns per store: whole 128-byte segments) ; only the six
        // base rows of d a, which the integrator combines with the SE(3) blocks of P12, are kept — in LDS (dbase: the region of Yc, dead since P9).
        // (Rounds 3 - 5 sent all of d a through an HBM scratch and read it back in P12: 33 KB out and in per knot, 16 rounds of loads.)
        // d lambda = Z2 stays in LDS for the force terms.
        if (z < nz) {
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int r = 16 * bi + rq + 4 * qq;
              if (bi < nbm && r < nv) {
                if (direct_ab) {
                  const double dav = -w[bi][qq];
                  const double dvp = dt_se3 * dav + ((z == nv + r) ? 1.0 : 0.0);
                  kn[KL.oAB + (size_t)(nv + r) * KL.nz + z] = dvp;
                  if (r >= 6) kn[KL.oAB + (size_t)r * KL.nz + z] = dt_se3 * dvp + ((z == r) ? 1.0 : 0.0);
                  else dbase[r * nz + z] = dav;
                } else dsol[(size_t)r * L.nz + z] = -w[bi][qq];
              }
            }
        }
#pragma unroll
        for (int qq = 0; qq < 3; ++qq) DL[(rq + 4 * qq) * ldl + 16 * cj + (lane & 15)] = t[qq];
      }
#ifdef EV_P11PROF
      const long long tp11b_ = clock64();
      if (TRIAL == 0 && a.prof && k == 1 && lane == 0) a.prof[(size_t)b * 64 + 32 + 16 + wv] += (double)(tp11b_ - tp11_);
#endif
      if (wv == nw - 1 && lane == 0) step_se3_fused(nw > 2 ? 1 : 0);  // (P12's single-lane work, on the wavefront with the fewest column blocks ; Jexp6 / Ad^-1: done in P10)
#ifdef EV_P11PROF
      if (TRIAL == 0 && a.prof && k == 1 && lane == 0 && wv == nw - 1) a.prof[(size_t)b * 64 + 32 + 3] += (double)(clock64() - tp11b_);
#endif
      __syncthreads();
    }
  }

  EV_PROF(9);
  if (!has_dyn) se3_prepass(nw > 2 ? 2 : nw - 1);  // (stages with contact dynamics: done beside the factorisation)
  // ---- P12: semi-implicit Euler, gap and its Jacobians ----------------------------------------------------
  if (has_dyn || kino) {
    const double dt = P[desc[4]];
    double* Jl6 = small;        // Jlog6(G)
    double* Je6 = small + 36;   // Jexp6(delta)
    double* Jq6 = small + 72;   // Ad(exp6(delta))^-1
    double* D12 = small + 108;  // D1_b = Jlog6(G) Jq6 (36) | Dd_b = dt Jlog6(G) Jexp6 (36)   (layout.h, oD12)
    // The SE(3) pieces are single-lane work (log / exp maps and their Jacobians): spread them over the wavefronts
    // — wave 0: step, gap, Jlog6(G) ; wave 1: Jexp6, Ad^-1, E6 ; waves 2..: the SE(3)-valued cost / constraint terms
    if (!(has_dyn && derivs) && tid == 0) step_se3_fused();
    for (int i = 6 + tid; i < n; i += nthr) {
      if (i < nv) { const double vp = v[i] + dt * acc[i]; kn[KL.oF + i] = q[i + 1] + dt * vp - xn[i + 1]; if (derivs) kn[KL.oXN + i + 1] = q[i + 1] + dt * vp; }
      else if (i >= nv) { const int j = i - nv; const double vp = v[j] + dt * acc[j]; kn[KL.oF + i] = vp - xn[nq + j]; if (derivs) kn[KL.oXN + nq + j] = vp; }
    }
    __syncthreads();
    if (derivs) {
      // base rows in factored form (layout.h, oD12): D1_b = Jlog6(G) Jq6, Dd_b = dt Jlog6(G) Jexp6
      if (tid < 72) {
        const int e = tid % 36, r = e / 6, cc = e % 6;
        const double* Rm_ = (tid < 36) ? Jq6 : Je6;
        double s = 0;
        for (int l = 0; l < 6; ++l) s += Jl6[6 * r + l] * Rm_[6 * l + cc];
        s = (tid < 36) ? s : dt * s;
        D12[tid] = s;
        kn[KL.oD12 + tid] = s;
      } else if (tid == 72) { kn[KL.oD12 + 72] = dt; kn[KL.oD12 + 73] = 1.0; }
      // dvp = dt * da + [0 I 0];  rows nv..n of AB = dvp; rows 6..nv = dt dvp + [I 0 0]
      // (d a comes back from the HBM scratch: MB_AB_U loads are requested before the first store — a load per iteration, ordered behind
      // the stores of the one before, costs a round trip to L2 every time: 16 of them for nv = 38)
      if (!direct_ab) for (int base = tid; base < nv * nz; base += nthr * MB_AB_U) {
        double dav[MB_AB_U];
#pragma unroll
        for (int uu = 0; uu < MB_AB_U; ++uu) {
          const int idx = base + uu * nthr, r = qdiv(idx, mg_nz), z = idx - r * nz;
          dav[uu] = (idx < nv * nz) ? dsol[(size_t)r * L.nz + z] : 0.0;
        }
#pragma unroll
        for (int uu = 0; uu < MB_AB_U; ++uu) {
          const int idx = base + uu * nthr, r = qdiv(idx, mg_nz), z = idx - r * nz;
          if (idx < nv * nz) {
            const double dvp = dt * dav[uu] + ((z == nv + r) ? 1.0 : 0.0);
            kn[KL.oAB + (size_t)(nv + r) * KL.nz + z] = dvp;
            if (r >= 6) kn[KL.oAB + (size_t)r * KL.nz + z] = dt * dvp + ((z == r) ? 1.0 : 0.0);
          }
        }
      }
      __syncthreads();  // (D12)
      // base rows: D1_b [I 0 0] + Dd_b dvp[0:6]
      for (int idx = tid; idx < 6 * nz; idx += nthr) {
        const int r = qdiv(idx, mg_nz), z = (idx - qdiv(idx, mg_nz) * nz);
        double dl6[6];
#pragma unroll
        for (int l = 0; l < 6; ++l) dl6[l] = direct_ab ? Yc[l * nz + z] : dsol[(size_t)l * L.nz + z];
        double s = (z < 6) ? D12[6 * r + z] : 0.0;
#pragma unroll
        for (int l = 0; l < 6; ++l) s += D12[36 + 6 * r + l] * (dt * dl6[l] + ((z == nv + l) ? 1.0 : 0.0));
        kn[KL.oAB + (size_t)r * KL.nz + z] = s;
      }
      __syncthreads();
    }
  }

  else {
    if (derivs && tid == 0) kn[KL.oD12 + 73] = 0.0;  // no dynamics rows: nothing to factor
    __syncthreads();  // the SE(3) table must be complete before the terms read it
  }

  EV_PROF(10);
  // ---- P13: cost stack and constraints -----------------------------------------------------------------------
  // Cost terms with diagonal weights are evaluated CONCURRENTLY, one term per wavefront, straight into their rows of
  // the LDS stack JS = sqrt(W) J (<= 32 rows per chunk); the Gauss-Newton Hessian JS^T JS (+ the diagonal / base-block
  // contributions of the state and control error terms, accumulated in LDS) is written ONCE per knot by MFMA tiles —
  // no clear, no read-modify-write of the nz x nz block in HBM.  Constraints and the diagonal state / control costs
  // run through the whole workgroup term by term.
  const S6 h0 = ldc6(Hc, nj, 0);
  // leading dimension of the Jacobian rows staged in LDS: >= nz and = 16 (mod 32) doubles, so that the four k-rows a wavefront reads
  // per MFMA step of J^T J fall on disjoint banks (ld = nz = 108 gave 2-way conflicts on every operand read of the Hessian flush)
  const int ldj = mb_ldj(nz);
  double* gacc = Bt;     // nz: gradient accumulator        (the derivative vectors Bt, Tv, Phi are dead by now)
  double* hdg = Tv;      // nz: additions to diag(H)
  double* hbb = Phi;     // 36: additions to the base 6x6 block of H
  double* tcost = small; // per-term cost, summed in term order at the end (deterministic)
  const bool early_done = has_dyn && early[2 * nz + 60] != 0.0;  // the diagonal state / control costs are in the early accumulators already
  for (int z = tid; z < nz; z += nthr) { gacc[z] = early_done ? early[z] : 0.0; hdg[z] = early_done ? early[nz + z] : a.opt.reg_init; }
  for (int i = tid; i < 36; i += nthr) hbb[i] = early_done ? early[2 * nz + i] : 0.0;
  for (int i = tid; i < 24; i += nthr) tcost[i] = early_done ? early[2 * nz + 36 + i] : 0.0;
  if (!has_dyn && tid == 0) classify_terms();  // (stages with contact dynamics: done beside the factorisation)
  __syncthreads();
  EV_PROF(29);

  // residual entry ri of x_ref (-) x : base rows from the SE(3) table, joint rows in closed form
  auto state_res = [&](const double* tp, const double* sl, int ri) -> double {
    return (ri < 6) ? sl[ri] : ((ri < nv) ? (tp[ri + 1] - q[ri + 1]) : (tp[nq + ri - nv] - v[ri - nv]));
  };
  // r (dim) and, with derivatives, the Jacobian rows Jt (dim x nz, ld nz) of one term, by the threads t0 (of nt): the
  // whole workgroup (wg, barriers) or one wavefront (LDS operations of a wavefront execute in order: no barrier)
  auto term_rows = [&](const TermRec& tr, const double* tp, const double* sl, double* r, double* Jt, int t0, int nt, bool wg) {
    const int d = tr.dim;
    if (derivs) for (int idx = t0; idx < d * ldj; idx += nt) Jt[idx] = 0.0;
    if (wg) __syncthreads();
    if (tr.type == MPC_TERM_STATE_ERROR) {
      const double* Jb = sl + 8;  // -Jlog6 block of the base rows
      for (int i = t0; i < d; i += nt) r[i] = state_res(tp, sl, tr.i0 + i);
      if (derivs) for (int i = t0; i < d; i += nt) {
        const int ri = tr.i0 + i;
        if (ri < 6) { for (int z = 0; z < 6; ++z) Jt[i * ldj + z] = Jb[6 * ri + z]; }
        else Jt[i * ldj + ri] = -1.0;
      }
    } else if (tr.type == MPC_TERM_CONTROL_ERROR) {
      for (int i = t0; i < d; i += nt) { r[i] = u[tr.i0 + i] - tp[tr.i0 + i]; if (derivs) Jt[i * ldj + n + tr.i0 + i] = 1.0; }
    } else if (tr.type == MPC_TERM_FRAME_PLACEMENT || tr.type == MPC_TERM_FRAME_TRANSLATION || tr.type == MPC_TERM_FRAME_VELOCITY) {
      const int fi = tr.i0, i = mframe[fi];
      const M3 Ri = ldcm3(oR, nj, i);
      const M3 Rf = mul(Ri, ldm3(fd + 12 * fi));
      const V3 pf = mul(Ri, ldv3(fd + 12 * fi + 9)) + ldcv3(op, nj, i);
      if (tr.type == MPC_TERM_FRAME_PLACEMENT) {
        const double* Jl = sl + 8;
        if (t0 < 6) r[t0] = sl[t0];
        if (derivs) for (int j = t0; j < nv; j += nt) if (BELOW(j, i)) {
          const S6 col = mat6_mul(Jl, adinv(Rf, pf, ldc6(J, nv, j)));
          for (int rr = 0; rr < 6; ++rr) Jt[rr * ldj + j] = col.v[rr];
        }
      } else if (tr.type == MPC_TERM_FRAME_TRANSLATION) {
        if (t0 < d) { const double pfa[3] = {pf.x, pf.y, pf.z}; r[t0] = pfa[tr.i1 + t0] - tp[tr.i1 + t0]; }
        if (derivs) for (int j = t0; j < nv; j += nt) if (BELOW(j, i)) {
          const S6 Jj = ldc6(J, nv, j);
          const V3 lv = lin(Jj) + cross(ang(Jj), pf);
          const double la[3] = {lv.x, lv.y, lv.z};
          for (int rr = 0; rr < d; ++rr) Jt[rr * ldj + j] = la[tr.i1 + rr];
        }
      } else {
        if (t0 == 0) { const S6 vf = adinv(Rf, pf, ldc6(ov, nj, i)); for (int rr = 0; rr < 6; ++rr) r[rr] = vf.v[rr] - tp[rr]; }
        if (derivs) for (int j = t0; j < nv; j += nt) if (BELOW(j, i)) {
          const S6 cq = adinv(Rf, pf, ldc6(Psd, nv, j)), cv = adinv(Rf, pf, ldc6(J, nv, j));
          for (int rr = 0; rr < 6; ++rr) { Jt[rr * ldj + j] = cq.v[rr]; Jt[rr * ldj + nv + j] = cv.v[rr]; }
        }
      }
    } else if (tr.type == MPC_TERM_COM_TRANSLATION) {
      if (t0 < d) { const double ca[3] = {com.x, com.y, com.z}; r[t0] = ca[tr.i1 + t0] - tp[tr.i1 + t0]; }
      if (derivs) for (int j = t0; j < nv; j += nt) for (int rr = 0; rr < d; ++rr) Jt[rr * ldj + j] = U[(tr.i1 + rr) * nv + j] / mtot;
    } else if (tr.type == MPC_TERM_CENTROIDAL_MOMENTUM) {
      if (t0 == 0) {
        const V3 hl = lin(h0), ha = ang(h0) - cross(com, lin(h0));
        r[0] = hl.x - tp[0]; r[1] = hl.y - tp[1]; r[2] = hl.z - tp[2]; r[3] = ha.x - tp[3]; r[4] = ha.y - tp[4]; r[5] = ha.z - tp[5];
      }
      if (derivs) for (int j = t0; j < nv; j += nt) {
        const int bj = dof_body[j];
        const S6 Uj = ldc6(U, nv, j);
        const S6 D = add6(fcross(ldc6(J, nv, j), ldc6(Hc, nj, bj)), ldc6(YcPsd, nv, j));
        const V3 dc = (1.0 / mtot) * lin(Uj);
        const V3 dql = lin(D), dqa = ang(D) - cross(dc, lin(h0)) - cross(com, lin(D));
        const V3 dvl = lin(Uj), dva = ang(Uj) - cross(com, lin(Uj));
        const double cq[6] = {dql.x, dql.y, dql.z, dqa.x, dqa.y, dqa.z}, cv[6] = {dvl.x, dvl.y, dvl.z, dva.x, dva.y, dva.z};
        for (int rr = 0; rr < 6; ++rr) { Jt[rr * ldj + j] = cq[rr]; Jt[rr * ldj + nv + j] = cv[rr]; }
      }
    } else if (tr.type == MPC_TERM_CONTACT_FORCE) {
      if (t0 < 6) r[t0] = lam[6 * tr.i0 + t0] - tp[t0];
      if (derivs) for (int idx = t0; idx < 6 * nz; idx += nt) { const int i = qdiv(idx, mg_nz), z = idx - i * nz; Jt[i * ldj + z] = DL[(6 * tr.i0 + i) * ldl + z]; }
    } else if (tr.type == MPC_TERM_CENTROIDAL_WRENCH_CONE) {
      for (int i = t0; i < d; i += nt) { double sacc = 0; for (int j = 0; j < 6; ++j) sacc += tp[i * 6 + j] * u[6 * tr.i0 + j]; r[i] = sacc; }
      if (derivs) for (int idx = t0; idx < d * 6; idx += nt) Jt[(idx / 6) * ldj + n + 6 * tr.i0 + idx % 6] = tp[idx];
    } else if (tr.type == MPC_TERM_CENTROIDAL_MOMENTUM_DER) {
      // r = [sum f + m g ; sum (p_i - c) x f_i + tau_i]   (kinodynamic_talos.py:125-127); params: g[3], states, frames
      const int nkk = tr.i0;
      if (t0 == 0) {
        V3 rl = v3(mtot * tp[0], mtot * tp[1], mtot * tp[2]), ra = v3(0, 0, 0);
        for (int cc = 0; cc < nkk; ++cc) {
          if (tp[3 + cc] == 0.0) continue;
          const int fi = (int)tp[3 + nkk + cc], i = mframe[fi];
          const V3 pf = mul(ldcm3(oR, nj, i), ldv3(fd + 12 * fi + 9)) + ldcv3(op, nj, i);
          const V3 f = v3(u[6 * cc], u[6 * cc + 1], u[6 * cc + 2]);
          rl = rl + f;
          ra = ra + cross(pf - com, f) + v3(u[6 * cc + 3], u[6 * cc + 4], u[6 * cc + 5]);
        }
        r[0] = rl.x; r[1] = rl.y; r[2] = rl.z; r[3] = ra.x; r[4] = ra.y; r[5] = ra.z;
      }
      if (derivs) {
        for (int j = t0; j < nv; j += nt) {
          V3 dang = v3(0, 0, 0);
          const S6 Jj = ldc6(J, nv, j);
          const V3 dc = (1.0 / mtot) * lin(ldc6(U, nv, j));
          for (int cc = 0; cc < nkk; ++cc) {
            if (tp[3 + cc] == 0.0) continue;
            const int fi = (int)tp[3 + nkk + cc], i = mframe[fi];
            const V3 pf = mul(ldcm3(oR, nj, i), ldv3(fd + 12 * fi + 9)) + ldcv3(op, nj, i);
            V3 dp_ = v3(0, 0, 0);
            if (BELOW(j, i)) dp_ = lin(Jj) + cross(ang(Jj), pf);
            dang = dang + cross(dp_ - dc, v3(u[6 * cc], u[6 * cc + 1], u[6 * cc + 2]));
          }
          Jt[3 * ldj + j] = dang.x; Jt[4 * ldj + j] = dang.y; Jt[5 * ldj + j] = dang.z;
        }
        if (t0 < nkk && tp[3 + t0] != 0.0) {
          const int cc = t0, fi = (int)tp[3 + nkk + cc], i = mframe[fi];
          const V3 rr = mul(ldcm3(oR, nj, i), ldv3(fd + 12 * fi + 9)) + ldcv3(op, nj, i) - com;
          const M3 Rx = skew_m(rr);
          for (int e = 0; e < 3; ++e) {
            Jt[e * ldj + n + 6 * cc + e] = 1.0;
            Jt[(3 + e) * ldj + n + 6 * cc + 3 + e] = 1.0;
            for (int e2 = 0; e2 < 3; ++e2) Jt[(3 + e) * ldj + n + 6 * cc + e2] = Rx.m[3 * e + e2];
          }
        }
      }
    } else if (tr.type == MPC_TERM_MB_WRENCH_CONE) {
      for (int i = t0; i < d; i += nt) { double s = 0; for (int j = 0; j < 6; ++j) s += tp[i * 6 + j] * lam[6 * tr.i0 + j]; r[i] = s; }
      if (derivs) for (int idx = t0; idx < d * nz; idx += nt) {
        const int i = qdiv(idx, mg_nz), z = (idx - qdiv(idx, mg_nz) * nz);
        double s = 0;
        for (int j = 0; j < 6; ++j) s += tp[i * 6 + j] * DL[(6 * tr.i0 + j) * ldl + z];
        Jt[i * ldj + z] = s;
      }
    }
    if (wg) __syncthreads();
  };

  // ---- pass B: constraints and the diagonal state / control costs, term by term through the whole workgroup ----
  {
    int row = 0;
    double* r = red + 48;
    for (int t = 0; t < nterms; ++t) {
      const TermRec tr = lds_term(lterm, t);
      if (tkind[t] != 0) continue;
      const double* tp = P + tr.poff;
      const double* sl = se3 + 48 * tse3[t];
      const int d = tr.dim;
      const bool is_cost = tr.role == MPC_ROLE_COST;
      if (early_done && is_cost && (tr.type == MPC_TERM_STATE_ERROR || tr.type == MPC_TERM_CONTROL_ERROR)) {
        // (done beside the factorisation: early_costs)
      } else if (tr.type == MPC_TERM_STATE_ERROR && is_cost) {
        const double* W = P + tr.woff;
        const double* Jb = sl + 8;
        if (wv == 0) {  // cost value: one wavefront, DPP reduction (a single thread walking d residuals costs ~15 us)
          double cst = 0;
          for (int i = lane; i < d; i += 64) { const double e = state_res(tp, sl, tr.i0 + i); cst += W[i] * e * e; }
          cst = wave_sum(cst);
          if (lane == 0) tcost[t] = 0.5 * cst;
        }
        if (derivs) {
          for (int z = tid; z < n; z += nthr) {
            double g = 0;
            if (z < 6) { for (int i = 0; i < d && tr.i0 + i < 6; ++i) { const int ri = tr.i0 + i; g += Jb[6 * ri + z] * W[i] * sl[ri]; } }  // base rows only
            else if (z >= tr.i0 && z < tr.i0 + d) { g = -W[z - tr.i0] * state_res(tp, sl, z); hdg[z] += W[z - tr.i0]; }
            gacc[z] += g;
          }
          for (int idx = tid; idx < 36; idx += nthr) {
            const int za = idx / 6, zb = idx % 6;
            double h = 0;
            for (int i = 0; i < d && tr.i0 + i < 6; ++i) { const int ri = tr.i0 + i; h += Jb[6 * ri + za] * W[i] * Jb[6 * ri + zb]; }
            hbb[idx] += h;
          }
        }
        __syncthreads();
      } else if (tr.type == MPC_TERM_CONTROL_ERROR && is_cost) {
        const double* W = P + tr.woff;
        if (wv == nw - 1) {
          double cst = 0;
          for (int i = lane; i < d; i += 64) { const double e = u[tr.i0 + i] - tp[tr.i0 + i]; cst += W[i] * e * e; }
          cst = wave_sum(cst);
          if (lane == 0) tcost[t] = 0.5 * cst;
        }
        if (derivs) for (int i = tid; i < d; i += nthr) {
          const int z = n + tr.i0 + i;
          gacc[z] += W[i] * (u[tr.i0 + i] - tp[tr.i0 + i]);
          hdg[z] += W[i];
        }
        __syncthreads();
      } else if (has_dyn && ((tr.type == MPC_TERM_STATE_ERROR && tr.i0 >= 6) || tr.type == MPC_TERM_CONTROL_ERROR)) {
        // (selector constraints of a stage with contact dynamics: written beside the factorisation, selector_rows)
      } else if ((tr.type == MPC_TERM_STATE_ERROR && tr.i0 >= 6) || tr.type == MPC_TERM_CONTROL_ERROR) {
        // selector constraints (joint limits fulldynamic_talos.py:208-209, torque box :206-207): straight into the record
        const bool st_ = tr.type == MPC_TERM_STATE_ERROR;
        for (int i = tid; i < d; i += nthr) {
          kn[KL.oCV + row + i] = st_ ? state_res(tp, sl, tr.i0 + i) : (u[tr.i0 + i] - tp[tr.i0 + i]);
          kn[KL.oCT + row + i] = (double)tr.role;
          kn[KL.oLO + row + i] = (tr.role == MPC_ROLE_BOX) ? P[tr.woff + i] : 0.0;
          kn[KL.oHI + row + i] = (tr.role == MPC_ROLE_BOX) ? P[tr.woff + d + i] : 0.0;
        }
        const int zc0 = st_ ? tr.i0 : n + tr.i0;
        const double sgn = st_ ? -1.0 : 1.0;
        if (derivs) for (int i = wv; i < d; i += nw) for (int z = lane; z < nz; z += 64) kn[KL.oCD + (size_t)(row + i) * KL.nz + z] = (z == zc0 + i) ? sgn : 0.0;
      } else if (tr.type == MPC_TERM_MB_WRENCH_CONE) {
        // rows of the cone matrix times lambda / d lambda, straight into the record (no staging, no barrier): a row per wavefront
        // (the cone rows of ALL the rows of this wavefront are requested first: a row's six parameters loaded when its turn comes
        // cost a round trip to L2 per row, 2.5 us per contact)
        constexpr int CONE_RPW = (MB_STAGE_CONSTRAINT_ROWS + EVAL_THREADS / 64 - 1) / (EVAL_THREADS / 64);
        double cwr[CONE_RPW][6];
#pragma unroll
        for (int qr = 0; qr < CONE_RPW; ++qr) {
          const int i = wv + qr * nw;
#pragma unroll
          for (int j = 0; j < 6; ++j) cwr[qr][j] = (derivs && i < d) ? tp[i * 6 + j] : 0.0;
        }
        for (int i = tid; i < d; i += nthr) {
          double sres = 0;
          for (int j = 0; j < 6; ++j) sres += tp[i * 6 + j] * lam[6 * tr.i0 + j];
          kn[KL.oCV + row + i] = sres;
          kn[KL.oCT + row + i] = (double)tr.role;
          kn[KL.oLO + row + i] = (tr.role == MPC_ROLE_BOX) ? P[tr.woff + i] : 0.0;
          kn[KL.oHI + row + i] = (tr.role == MPC_ROLE_BOX) ? P[tr.woff + d + i] : 0.0;
        }
        if (derivs) {
#pragma unroll
          for (int qr = 0; qr < CONE_RPW; ++qr) {
            const int i = wv + qr * nw;
            if (i < d) for (int z = lane; z < nz; z += 64) {
              double sres = 0;
#pragma unroll
              for (int j = 0; j < 6; ++j) sres += cwr[qr][j] * DL[(6 * tr.i0 + j) * ldl + z];
              kn[KL.oCD + (size_t)(row + i) * KL.nz + z] = sres;
            }
          }
          for (int i = wv + CONE_RPW * nw; i < d; i += nw) {  // (more rows than a cone has: not reached)
            double cw[6];
            for (int j = 0; j < 6; ++j) cw[j] = tp[i * 6 + j];
            for (int z = lane; z < nz; z += 64) {
              double sres = 0;
              for (int j = 0; j < 6; ++j) sres += cw[j] * DL[(6 * tr.i0 + j) * ldl + z];
              kn[KL.oCD + (size_t)(row + i) * KL.nz + z] = sres;
            }
          }
        }
      } else {
        term_rows(tr, tp, sl, r, JL, tid, nthr, true);
        emit_constraint(KL, kn, tr, P, row, r, JL, ldj, nz, derivs, tid, nthr);
      }
      if (!is_cost) row += d;
      EV_PROF(13 + tr.type);
    }
  }
  __syncthreads();

  // ---- pass A: stacked cost terms, one per wavefront; then H = JS^T JS (+ diagonal / base additions) on the MFMA ----
  const int nchunks = tmeta[0];
  for (int ch = 0; ch < (nchunks > 0 ? nchunks : 1); ++ch) {
    int rowc = 0, ord = 0;
    for (int t = 0; t < nterms; ++t) {
      if (tkind[t] != 1 || tchunk[t] != ch) continue;
      const TermRec tr = lds_term(lterm, t);
      const int d = tr.dim;
      if (trow[t] + d > rowc) rowc = trow[t] + d;
      if ((ord++ % nw) != wv) continue;
      double* r = red + 112 + 24 * wv;  // private residual scratch of this wavefront
      double* Jt = JS + trow[t] * ldj;
      term_rows(tr, P + tr.poff, se3 + 48 * tse3[t], r, Jt, lane, 64, false);
      const double* W = P + tr.woff;
      const int wstride = (tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT) ? 1 : d + 1;
      const double wl = (lane < d) ? W[lane * wstride] : 0.0;  // d <= 24: the weight of row `lane`
      {
        double cst = (lane < d) ? wl * r[lane] * r[lane] : 0.0;
        cst = wave_sum(cst);
        if (lane == 0) tcost[t] = 0.5 * cst;
      }
      if (derivs) {
        // sqrt(W) once per row, handed to the entries of the row through the lanes (a load of W and a square root per ENTRY was a
        // round trip to L2 in each of the d nz / 64 rounds)
        const double swl = sqrt(wl);
        if (lane < d) wrs[trow[t] + lane] = swl * r[lane];
        for (int idx = lane; idx < ((d * nz + 63) & ~63); idx += 64) {
          const int i = qdiv(idx, mg_nz), z = idx - i * nz;
          const double sw = __shfl(swl, i < d ? i : 0);
          if (idx < d * nz) Jt[i * ldj + z] *= sw;
        }
      }
    }
    EV_PROF(28);
    if (derivs) {
      const int kc = (rowc + 3) & ~3;  // MFMA K granularity: zero rows up to a multiple of 4
      __syncthreads();
      for (int idx = tid; idx < (kc - rowc) * ldj; idx += nthr) JS[rowc * ldj + idx] = 0.0;
      __syncthreads();
      for (int z = tid; z < nz; z += nthr) {
        double g = 0;
        for (int i = 0; i < rowc; ++i) g += JS[i * ldj + z] * wrs[i];
        gacc[z] += g;
      }
      // upper block triangle of 16x16 tiles, mirrored on the way out; the first chunk writes, later ones accumulate
      const int nzt = (nz + 15) >> 4;
      for (int t = wv; t < nzt * (nzt + 1) / 2; t += nw) {
        int ta = 0, rem = t;
        while (rem >= nzt - ta) { rem -= nzt - ta; ++ta; }
        const int tb = ta + rem;
        d4_t h = d4_t{0, 0, 0, 0};
        mma_tile<false>(h, JS + ta * 16, 1, ldj, JS + tb * 16, ldj, 1, kc, lane);
        const int zb = tb * 16 + (lane & 15);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int za = ta * 16 + (lane >> 4) + 4 * qq;
          if (za < nz && zb < nz) {
            double hv = h[qq];
            if (ch == 0) {
              if (za == zb) hv += hdg[za];
              if (za < 6 && zb < 6) hv += hbb[6 * za + zb];
              kn[KL.oH + (size_t)za * KL.nz + zb] = hv;
              if (ta != tb) kn[KL.oH + (size_t)zb * KL.nz + za] = hv;
            } else {
              kn[KL.oH + (size_t)za * KL.nz + zb] += hv;
              if (ta != tb) kn[KL.oH + (size_t)zb * KL.nz + za] += hv;
            }
          }
        }
      }
      __syncthreads();
    }
    EV_PROF(27);
  }
  if (derivs) for (int z = tid; z < nz; z += nthr) kn[KL.oG + z] = gacc[z];
  // ---- pass C: cost terms with dense weights (none in the three Talos problems): HBM read-modify-write path ----
  if (tmeta[1]) {
    __syncthreads();
    double* r = red + 48;
    for (int t = 0; t < nterms; ++t) {
      if (tkind[t] != 2) continue;
      const TermRec tr = lds_term(lterm, t);
      term_rows(tr, P + tr.poff, se3 + 48 * tse3[t], r, JS, tid, nthr, true);
      if (derivs) for (int idx = tid; idx < tr.dim * nz; idx += nthr) { const int i = qdiv(idx, mg_nz), z = idx - i * nz; JtG[idx] = JS[i * ldj + z]; }
      __syncthreads();
      double cst = 0.0;
      accumulate_cost(KL, kn, tr, P + tr.woff, r, JtG, nz, nz, red + 112, WJ, derivs, cst, tid, nthr);
      if (tid == 0) tcost[t] = cst;
    }
  }
  __syncthreads();
  if (tid == 0) { double sc = 0; for (int t = 0; t < nterms; ++t) sc += tcost[t]; s_cost = sc; }
  __syncthreads();

  EV_PROF(11);
  // ---- P14: projections, AL penalty, infeasibility ------------------------------------------------------------
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const size_t vo = ((size_t)b * (N + 1) + k) * L.c, lo = ((size_t)b * (N + 1) + k + 1) * n;
  double pen = 0, prim = 0;
  knot_merit(KL, kn, c, (k < N) ? kn + KL.oF : nullptr, a.vs + vo, CAND ? a.dvs + vo : nullptr, a.vs_e + vo,
             a.lams + lo, CAND ? a.dlams + lo : nullptr, a.lams_e + lo, alpha, mu, mud, derivs, red, pen, prim, tid, nthr);
  if (tid == 0) {
    if (CAND && !specw) a.trial_phi[((size_t)b * L.n_alpha + cand) * (N + 1) + k] = s_cost + pen;
    if (TRIAL == 0 || TRIAL == 3) {
      double* ms = kn + KL.oMISC;
      ms[MISC_COST] = s_cost; ms[MISC_PEN] = pen; ms[MISC_PRIM] = prim; ms[MISC_NC] = (double)c; ms[MISC_M] = (double)m;
    }
  }
  EV_PROF(12);
  // Backtracking candidates one after the other in the same workgroup: a launch with one workgroup per candidate costs
  // 7 (N + 1) B dispatches that each need the whole LDS of a CU just to find out that the full step was accepted.
  if (TRIAL == 1 && mb.ncand_loop > 0 && cand + 1 < cand0 + mb.ncand_loop) { ++cand; __syncthreads(); goto cand_loop; }
#undef BELOW
}
