// eval_multibody.h — per-knot evaluation of whole-body stages on the GPU: the stage of fulldynamic_talos.py:100-232
// (MultibodyConstraintFwdDynamics = pin.constraintDynamics + its derivatives, IntegratorSemiImplEuler, state /
// control / centroidal-momentum / frame-placement / contact-force costs, torque & joint boxes, wrench cones).
//
// One workgroup (256 threads) per (knot, instance, linesearch candidate).  All rigid-body quantities are kept in
// LDS in a WORLD-FRAME formulation (spatial vectors [lin; ang] taken at the world origin): after one pass over
// the kinematic tree every entry of M, d tau/dq, d tau/dv, the contact Jacobians and their derivatives is a
// 6-dimensional dot product of per-dof vectors, so the nv x nv blocks are filled by all lanes in parallel with
// no further dependency on the tree (DESIGN.md §"Whole-body stage kernel" derives the formulas; they are
// cross-checked against the AD-based oracle through tests/proto_multibody.py and the GPU parity tests).
#pragma once
#include <atomic>
#include <stdexcept>
#include "eval_common.h"
#include "mfma_blocks.h"

#ifndef EVAL_THREADS
#define EVAL_THREADS 512  // threads per stage workgroup (8 wavefronts: one 16-column block of the KKT solves each)
#endif
#define MB_SE3_SLOTS 8
#define MB_STAGE_CONSTRAINT_ROWS 20  // LDS staging rows for the Jacobian of one constraint term (wrench cone: 17)

// ---- 6-vectors --------------------------------------------------------------------------------------------
struct S6 { double v[6]; };
DEV S6 ld6(const double* p) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = p[i]; return r; }
DEV void st6(double* p, const S6& a) { for (int i = 0; i < 6; ++i) p[i] = a.v[i]; }
DEV S6 zero6() { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = 0.0; return r; }
DEV S6 add6(const S6& a, const S6& b) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
DEV S6 sub6(const S6& a, const S6& b) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
DEV S6 scale6(double s, const S6& a) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = s * a.v[i]; return r; }
DEV double dot6(const S6& a, const S6& b) { double s = 0; for (int i = 0; i < 6; ++i) s += a.v[i] * b.v[i]; return s; }
DEV V3 lin(const S6& a) { return v3(a.v[0], a.v[1], a.v[2]); }
DEV V3 ang(const S6& a) { return v3(a.v[3], a.v[4], a.v[5]); }
DEV S6 mk6(V3 l, V3 a) { S6 r; r.v[0] = l.x; r.v[1] = l.y; r.v[2] = l.z; r.v[3] = a.x; r.v[4] = a.y; r.v[5] = a.z; return r; }
// motion x motion and motion x* force
DEV S6 mcross(const S6& a, const S6& b) { return mk6(cross(ang(a), lin(b)) + cross(lin(a), ang(b)), cross(ang(a), ang(b))); }
DEV S6 fcross(const S6& a, const S6& f) { return mk6(cross(ang(a), lin(f)), cross(ang(a), ang(f)) + cross(lin(a), lin(f))); }
DEV S6 mat6_mul(const double* Y, const S6& x) { S6 r; for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += Y[6 * i + j] * x.v[j]; r.v[i] = s; } return r; }
DEV S6 mat6_tmul(const double* Y, const S6& x) { S6 r; for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += Y[6 * j + i] * x.v[j]; r.v[i] = s; } return r; }
DEV M3 ldm3(const double* p) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
DEV V3 ldv3(const double* p) { return v3(p[0], p[1], p[2]); }
// Ad(M)^-1 on a motion, M = (R, p)
DEV S6 adinv(const M3& R, V3 p, const S6& m) { return mk6(tmul(R, lin(m) - cross(p, ang(m))), tmul(R, ang(m))); }

// ---- SE(3) Jacobians (Barfoot's Q block; right Jacobians as used by Pinocchio's Jlog6 / Jexp6) ------------
DEV void q_coeffs(double t2, double& a1, double& a2, double& a3) {
  if (t2 < kSmall2) {
    a1 = 1.0 / 6 - t2 * (1.0 / 120 - t2 * (1.0 / 5040 - t2 * (1.0 / 362880)));
    a2 = 1.0 / 24 - t2 * (1.0 / 720 - t2 * (1.0 / 40320 - t2 * (1.0 / 3628800)));
    a3 = 1.0 / 120 - t2 * (1.0 / 2520 - t2 * (1.0 / 120960 - t2 * (1.0 / 9979200)));
  } else {
    const double t = sqrt(t2), s = sin(t), c = cos(t);
    a1 = (t - s) / (t2 * t); a2 = (t2 + 2 * c - 2) / (2 * t2 * t2); a3 = (2 * t - 3 * s + t * c) / (2 * t2 * t2 * t);
  }
}
DEV M3 add3(const M3& A, const M3& B) { M3 C; for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] + B.m[i]; return C; }
DEV M3 scl3(double s, const M3& A) { M3 C; for (int i = 0; i < 9; ++i) C.m[i] = s * A.m[i]; return C; }
DEV M3 Qmat(V3 v, V3 w) {
  double a1, a2, a3;
  q_coeffs(dot(w, w), a1, a2, a3);
  const M3 P = skew_m(v), F = skew_m(w);
  const M3 FP = mul(F, P), PF = mul(P, F), FPF = mul(FP, F), FF = mul(F, F);
  M3 Q = scl3(0.5, P);
  Q = add3(Q, scl3(a1, add3(add3(FP, PF), FPF)));
  Q = add3(Q, scl3(a2, add3(add3(mul(FF, P), mul(P, FF)), scl3(-3.0, FPF))));
  Q = add3(Q, scl3(a3, add3(mul(FPF, F), mul(F, FPF))));
  return Q;
}
// out (6x6 row-major) = Jlog6 at M = (R, p)
DEV void Jlog6(const M3& R, V3 p, double* out) {
  V3 v, w;
  log6(R, p, v, w);
  const double t2 = dot(w, w);
  double c;
  if (t2 < kSmall2) c = 1.0 / 12 + t2 * (1.0 / 720 + t2 * (1.0 / 30240 + t2 * (1.0 / 1209600)));
  else { const double t = sqrt(t2); c = (1.0 - t * cos(0.5 * t) / (2.0 * sin(0.5 * t))) / t2; }
  const M3 K = skew_m(w), K2 = mul(K, K);
  M3 Ji;
  for (int i = 0; i < 9; ++i) Ji.m[i] = ((i % 4 == 0) ? 1.0 : 0.0) + 0.5 * K.m[i] + c * K2.m[i];
  const M3 Q = Qmat(v3(-v.x, -v.y, -v.z), v3(-w.x, -w.y, -w.z));
  const M3 B = mul(mul(Ji, Q), Ji);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    out[6 * i + j] = Ji.m[3 * i + j]; out[6 * (i + 3) + j + 3] = Ji.m[3 * i + j];
    out[6 * i + j + 3] = -B.m[3 * i + j]; out[6 * (i + 3) + j] = 0.0;
  }
}
DEV void Jexp6(V3 v, V3 w, double* out) {
  double A, B, C;
  so3_coeffs(dot(w, w), A, B, C);
  const M3 K = skew_m(w), K2 = mul(K, K);
  const M3 Q = Qmat(v3(-v.x, -v.y, -v.z), v3(-w.x, -w.y, -w.z));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    const double jr = ((i == j) ? 1.0 : 0.0) - B * K.m[3 * i + j] + C * K2.m[3 * i + j];
    out[6 * i + j] = jr; out[6 * (i + 3) + j + 3] = jr; out[6 * i + j + 3] = Q.m[3 * i + j]; out[6 * (i + 3) + j] = 0.0;
  }
}

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

// ---- LDS carve-out ------------------------------------------------------------------------------------------
struct MbLds {
  int nj, nv, nq, nl_max;
  int nvp, ldm, nbm, ldR, ncb;  // padded nv, leading dim / block count of the mass matrix, leading dim / column blocks of R
  // body arrays
  int oR, op, ov, oa, oh, of, Fc, Hc, oY, Yc, Bc;
  // dof arrays
  int J, U, Psd, Psdd, Phi, Bt, Tq, Tv, vlam;
  // matrices / vectors
  int M, LIm, Y16, V16, Sp, LIs, R, Jc, gam, bias, a, lam, x, u, xn, cfr, small, se3, red, total;
  int stage_rows;  // rows of Jacobian staging (ld nz) that fit in the R region
  int anc_bytes_off, total_bytes;
  unsigned mg_nv, mg_nvp, mg_nz, mg_n;  // magic_div (device_common.h) of the run-time divisors nv, nvp, n + nu, n of the per-element loops
};

// The contact KKT system [[M, Jc^T], [Jc, -mu I]] is never inverted: M = L L^T (blocked Cholesky on the matrix
// cores), Y = L^-1 Jc^T, S = Y^T Y + mu I = Ls Ls^T, and every right-hand side — the dynamics residual and the
// nz columns of its derivatives, stacked in R = [R1 (nvp rows) ; R2 (16 rows)] — goes through
//   W = L^-1 R1 ; Z2 = S^-1 (Y^T W - R2) ; Z1 = L^-T (W - Y Z2)
// one 16-column block per wavefront.  R aliases the body-level inertia / Coriolis blocks (oY, Bc), dead by then,
// and later hosts the Jacobian rows of the cost / constraint terms.
static inline MbLds make_mb_lds(int nj, int nv, int nq, int nu, int nz) {
  MbLds s;
  s.nj = nj; s.nv = nv; s.nq = nq; s.nl_max = 12;
  s.nvp = (nv + 15) & ~15; s.ldm = s.nvp + 1; s.nbm = s.nvp / 16;
  s.ncb = (nz + 15) / 16; s.ldR = 16 * s.ncb + 1;
  int o = 0;
  auto take = [&](int c) { int r = o; o += (c + 1) & ~1; return r; };
  s.oR = take(9 * nj); s.op = take(3 * nj); s.ov = take(6 * nj); s.oa = take(6 * nj); s.oh = take(6 * nj); s.of = take(6 * nj);
  s.Fc = take(6 * nj); s.Hc = take(6 * nj); s.Yc = take(36 * nj);
  s.J = take(6 * nv); s.U = take(6 * nv); s.Psd = take(6 * nv); s.Psdd = take(6 * nv); s.Phi = take(6 * nv);
  s.Bt = take(6 * nv); s.Tq = take(6 * nv); s.Tv = take(6 * nv); s.vlam = take(12 * nv);
  s.M = take(s.nvp * s.ldm); s.LIm = take(s.nbm * 272); s.Y16 = take(s.nvp * 16); s.Sp = take(272); s.LIs = take(272); s.Jc = take(12 * nv);
  s.gam = take(16); s.bias = take(nv); s.a = take(s.nvp); s.lam = take(16);
  s.x = take(nq + nv); s.u = take(nu > 0 ? nu : 1); s.xn = take(nq + nv);
  s.cfr = take(2 * (12 + 36 + 6));  // per contact: R(9) p(3), Jlog6(c2Mc1) (36), spare(6)
  s.small = take(6 * 36 + 64);      // integrator 6x6 blocks and scratch
  s.se3 = take(MB_SE3_SLOTS * 48);  // per SE(3)-valued term: residual (6), Jacobian block (36 at +8)
  s.red = take(2 * 256 + 8);
  // R region: oY | Bc | rest
  s.oY = take(36 * nj); s.Bc = take(36 * nj);
  s.R = s.oY;
  // R = [R2 (16 rows: contact part, stays alive for the force terms) ; R1 (nvp rows)]; the Jacobian staging rows
  // (32 stacked cost rows + MB_STAGE_CONSTRAINT_ROWS rows of the constraint being emitted, ld nz) reuse the R1 part
  int r1size = s.nvp * s.ldR;
  if (r1size < (32 + MB_STAGE_CONSTRAINT_ROWS) * nz) r1size = (32 + MB_STAGE_CONSTRAINT_ROWS) * nz;
  const int rsize = 16 * s.ldR + r1size + 8;
  if (s.R + rsize > o) o = (s.R + rsize + 1) & ~1;
  if (o < s.Bc + 36 * nj + s.nvp * 16 + 4) o = (s.Bc + 36 * nj + s.nvp * 16 + 4 + 1) & ~1;  // V16 must not overlap oY / Bc
  s.V16 = o - s.nvp * 16 - 2;  // tail of the R region: alive only between the two solves, while oY / Bc are in use
  s.stage_rows = (o - s.R) / nz;
  s.total = o;
  s.anc_bytes_off = o * 8;
  s.total_bytes = o * 8 + 3 * nj * 8 + nv * 4 + nj * 4 * 3 + 64;
  s.mg_nv = magic_div(nv); s.mg_nvp = magic_div(s.nvp); s.mg_nz = magic_div(nz); s.mg_n = magic_div(2 * nv);
  return s;
}

static inline void check_multibody_model(const int32_t* itab, int n_i) {
  const int nj = itab[0];
  if (nj > 64 || itab[2] > 64) throw std::runtime_error("multibody kernel supports at most 64 bodies / 64 velocity dofs (bitmask tree tables)");
  const int32_t* ip = itab + MPC_MODEL_HEADER_WORDS;
  for (int i = 0; i < nj; ++i, ip += MPC_MODEL_JOINT_WORDS) {
    if (ip[0] >= i) throw std::runtime_error("model joints must be topologically ordered");
    if ((ip[1] == MPC_JOINT_FREEFLYER) != (i == 0)) throw std::runtime_error("the multibody kernel needs a free-flyer root followed by revolute joints");
    if (i > 0 && ip[2] != ip[3] + 1) throw std::runtime_error("unexpected idx_q / idx_v layout");
  }
  (void)n_i;
}

// doubles of per-workgroup HBM scratch: dsol [nK x nz] (da ; dlam) and the term Jacobian / weighted Jacobian
static inline size_t multibody_work_doubles(const Layout& L) {
  const int nv = L.n / 2;
  return (size_t)(nv + 12) * L.nz + 2 * (size_t)24 * L.nz + 64;
}

// lane ids pass through an empty asm at every phase boundary: index arithmetic stays phase-local instead of being
// kept live (and spilled) across the whole kernel — see RIC_LAUNDER in riccati_mfma.h
#define EV_LAUNDER() do { asm volatile("" : "+v"(tid)); lane = tid & 63; wv = __builtin_amdgcn_readfirstlane(tid >> 6); } while (0)
#define EV_PROF(slot) do { EV_LAUNDER(); if (TRIAL == 0 && tid == 0 && a.prof && k == 1) { const long long t1_ = clock64(); a.prof[(size_t)b * 64 + 32 + (slot)] += (double)(t1_ - t0_); t0_ = t1_; } } while (0)

struct MbArgs {
  MbLds lds;
  double* scratch;        // per-workgroup HBM scratch
  size_t scratch_stride;  // doubles
  int ncand_loop;         // TRIAL == 1: > 0 = the workgroup walks this many candidates itself (grid z = 1)
  int sim_substeps;       // TRIAL == 2 (closed-loop simulation stand-in): integration steps ...
  double sim_dt;          // ... of this length
};

// ============================================================================================================
template <int TRIAL>
__global__ void __launch_bounds__(EVAL_THREADS) k_eval_multibody(SolverArgs a, Layout KL, double* records, MbArgs mb, int cand0) {
  const Layout& L = a.L;
  const MbLds& S = mb.lds;
  // TRIAL == 3 with one workgroup more per instance (blockIdx.x == N + 1): the SPECULATIVE evaluation of the knot the next tick appends —
  // the accepted terminal state as a running stage with the table of the current last stage and the last control (what the warm-start
  // shift makes of it).  Its record goes to a spare slot (a.spec_knot) ; if the table of the appended stage turns out to be that one
  // (mpc_cycle compares), the next tick takes it as knot N - 1 instead of evaluating it (k_reproject, knot_reused).
  const bool specw = TRIAL == 3 && (int)blockIdx.x == a.L.N + 1;
  const int k = specw ? a.L.N - 1 : (int)blockIdx.x;  // stage table, control, multipliers
  const int b = blockIdx.y, nthr = blockDim.x;
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
  if (TRIAL == 0 && knot_reused(a, b, k)) return;     // tick reuse: the record is there already (k_reproject refreshes its projections)
  const int n = L.n, N = L.N, nx = L.nx, nv = S.nv, nq = S.nq, nj = S.nj, nu = L.m;
  const int slot = stage_slot(a, k);
  const int32_t* desc = a.stage_desc + (size_t)slot * L.max_stage_ints;
  const double* P = a.stage_params + (size_t)slot * L.max_stage_doubles;
  const int dyn = desc[0];
  const bool has_dyn = dyn == MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER;   // contact-constrained forward dynamics
  const bool kino = dyn == MPC_DYN_KINODYNAMICS_SEMIEULER;               // kinodynamics: u = [wrenches ; joint accelerations]
  const int m = (has_dyn || kino) ? nu : 0, nz = n + m, nterms = desc[5], c = desc[6];
  const int nk = has_dyn ? desc[1] : 0, nl = 6 * nk, nK = nv + nl;
  const unsigned mg_nz = m ? S.mg_nz : S.mg_n;  // nz = n + nu on a stage with dynamics, n on the terminal knot
  const bool derivs = (TRIAL == 0 || TRIAL == 3);
cand_loop:  // (TRIAL == 1 with mb.ncand_loop: next backtracking candidate of the same knot)
  const double alpha = CAND ? ldexp(1.0, -cand) : 0.0;
  const size_t wg = (TRIAL == 1) ? (((size_t)b * L.n_alpha + cand) * (N + 1) + k) : (specw ? (size_t)L.B * (N + 1) + b : ((size_t)b * (N + 1) + knot_slot(a, k)));
  double* kn = specw ? a.spec_knot + (size_t)b * KL.knot_stride : records + wg * KL.knot_stride;
  double* scr = mb.scratch + (derivs ? wg : 0) * mb.scratch_stride;  // value-only passes never touch it (B (N + 1) + B slots)
  double* dsol = scr;                         // [nK][nz]: rows < nv = da, rows >= nv = dlam
  double* JtG = scr + (size_t)(nv + 12) * L.nz;  // [24][nz] HBM fallback for dense (non-diagonal) weights
  double* WJ = JtG + (size_t)24 * L.nz;

  extern __shared__ __attribute__((aligned(16))) double sm[];
  unsigned long long* anc = (unsigned long long*)((char*)sm + S.anc_bytes_off);
  unsigned long long* sub = anc + nj;    // bodies of the subtree rooted at i
  unsigned long long* dmask = sub + nj;  // dofs of the joints on the path root .. i
  int* dof_body = (int*)(dmask + nj);
  int* parent = dof_body + nv;
  int* jkind = parent + nj;
  int* jidxv = jkind + nj;
  double *oR = sm + S.oR, *op = sm + S.op, *ov = sm + S.ov, *oa = sm + S.oa, *oh = sm + S.oh, *of = sm + S.of, *Fc = sm + S.Fc, *Hc = sm + S.Hc;
  double *oY = sm + S.oY, *Yc = sm + S.Yc, *Bc = sm + S.Bc;
  double *J = sm + S.J, *U = sm + S.U, *Psd = sm + S.Psd, *Psdd = sm + S.Psdd, *Phi = sm + S.Phi, *Bt = sm + S.Bt, *Tq = sm + S.Tq, *Tv = sm + S.Tv, *vlam = sm + S.vlam;
  double *M = sm + S.M, *LIm = sm + S.LIm, *Y16 = sm + S.Y16, *V16 = sm + S.V16, *Sp = sm + S.Sp, *LIs = sm + S.LIs, *Rm = sm + S.R, *Jc = sm + S.Jc;
  double *gam = sm + S.gam, *bias = sm + S.bias, *acc = sm + S.a, *lam = sm + S.lam;
  const int nvp = S.nvp, ldm = S.ldm, nbm = S.nbm, ldR = S.ldR, ncb = S.ncb;
  int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = nthr >> 6;
  double *x = sm + S.x, *u = sm + S.u, *xn = sm + S.xn, *cfr = sm + S.cfr, *small = sm + S.small, *red = sm + S.red;
  __shared__ int iflag[2];
  __shared__ double s_cost;
  // Jacobian staging in LDS, in the R region that is dead once the dynamics derivatives are in HBM:
  // JS = stacked rows sqrt(W) J of the cost terms (<= 32 rows), JL = rows of the constraint term being emitted
  double* R2 = Rm;                // contact rows of R / d lambda
  double* R1 = Rm + 16 * ldR;     // joint rows of R / -d a
  double* JS = R1;
  double* se3 = sm + S.se3;
  double* JL = JS + 32 * nz;
  double* wrs = red + 256;  // sqrt(W) r of the stacked rows
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
      for (int i = tid; i < nx; i += nthr) { x[i] = xs[i]; xn[i] = xs[i]; }  // simulated state ; xn = xs[0], the feedback reference
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
    }
    if (tid == 0) s_cost = 0.0;
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
#define BELOW(kdof, body) ((anc[(body)] >> dof_body[(kdof)]) & 1ull)
#define INSUB(j, i) ((anc[(j)] >> (i)) & 1ull)

  // ---- P1: local joint transforms (stored in the Bc region), then world placements ----------------------
  double* lR = Bc;
  double* lp = Bc + 9 * nj;
  for (int i = tid; i < nj; i += nthr) {
    const M3 Rp = ldm3(jd + 25 * i);
    const V3 pp = ldv3(jd + 25 * i + 9);
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
    for (int e = 0; e < 9; ++e) lR[9 * i + e] = Rl.m[e];
    lp[3 * i] = pl.x; lp[3 * i + 1] = pl.y; lp[3 * i + 2] = pl.z;
  }
  __syncthreads();
  for (int i = tid; i < nj; i += nthr) {
    M3 R = ldm3(lR + 9 * i);
    V3 p = ldv3(lp + 3 * i);
    for (int j = parent[i]; j >= 0; j = parent[j]) {
      const M3 Rj = ldm3(lR + 9 * j);
      p = mul(Rj, p) + ldv3(lp + 3 * j);
      R = mul(Rj, R);
    }
    for (int e = 0; e < 9; ++e) oR[9 * i + e] = R.m[e];
    op[3 * i] = p.x; op[3 * i + 1] = p.y; op[3 * i + 2] = p.z;
  }
  __syncthreads();
  // ---- P2: world-frame joint columns -------------------------------------------------------------------
  for (int kd = tid; kd < nv; kd += nthr) {
    const int i = dof_body[kd], loc = kd - jidxv[i];
    const M3 R = ldm3(oR + 9 * i);
    const V3 p = ldv3(op + 3 * i);
    S6 col;
    if (jkind[i] == MPC_JOINT_FREEFLYER && loc < 3) col = mk6(v3(R.m[loc], R.m[3 + loc], R.m[6 + loc]), v3(0, 0, 0));
    else {
      const int ax = (jkind[i] == MPC_JOINT_FREEFLYER) ? loc - 3 : jkind[i] - MPC_JOINT_RX;
      const V3 w = v3(R.m[ax], R.m[3 + ax], R.m[6 + ax]);
      col = mk6(cross(p, w), w);
    }
    st6(J + 6 * kd, col);
  }
  __syncthreads();
  EV_PROF(0);
  // ---- P3: body velocities ------------------------------------------------------------------------------
  for (int idx = tid; idx < 6 * nj; idx += nthr) {
    const int i = idx / 6, e = idx % 6;
    double s = 0;
    for (unsigned long long mm = dmask[i]; mm; mm &= mm - 1) { const int kd = __builtin_ctzll(mm); s += J[6 * kd + e] * v[kd]; }
    ov[idx] = s;
  }
  __syncthreads();
  // ---- P4: bias accelerations (gravity field), spatial inertias, momenta --------------------------------
  for (int i = tid; i < nj; i += nthr) {
    S6 ai = a0;
    for (unsigned long long mm = dmask[i]; mm; mm &= mm - 1) {
      const int kd = __builtin_ctzll(mm);
      ai = add6(ai, scale6(v[kd], mcross(ld6(ov + 6 * dof_body[kd]), ld6(J + 6 * kd))));
    }
    st6(oa + 6 * i, ai);
    const M3 R = ldm3(oR + 9 * i);
    const double mass = jd[25 * i + 12];
    const V3 cw = mul(R, ldv3(jd + 25 * i + 13)) + ldv3(op + 3 * i);
    const M3 RI = mul(R, ldm3(jd + 25 * i + 16));
    M3 Iww;  // R I R^T
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Iww.m[3 * r + cc] = RI.m[3 * r] * R.m[3 * cc] + RI.m[3 * r + 1] * R.m[3 * cc + 1] + RI.m[3 * r + 2] * R.m[3 * cc + 2];
    const M3 Sx = skew_m(cw), S2 = mul(Sx, Sx);
    double* Y = oY + 36 * i;
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
      Y[6 * r + cc] = (r == cc) ? mass : 0.0;
      Y[6 * r + cc + 3] = -mass * Sx.m[3 * r + cc];
      Y[6 * (r + 3) + cc] = mass * Sx.m[3 * r + cc];
      Y[6 * (r + 3) + cc + 3] = Iww.m[3 * r + cc] - mass * S2.m[3 * r + cc];
    }
    st6(oh + 6 * i, mat6_mul(Y, ld6(ov + 6 * i)));
  }
  __syncthreads();
  // ---- P5: composite inertias / momenta, bias forces ----------------------------------------------------
  for (int idx = tid; idx < 36 * nj; idx += nthr) {
    const int i = idx / 36, e = idx % 36;
    double s = 0;
    for (unsigned long long mm = sub[i]; mm; mm &= mm - 1) { const int j = __builtin_ctzll(mm); s += oY[36 * j + e]; }
    Yc[idx] = s;
  }
  for (int idx = tid; idx < 6 * nj; idx += nthr) {
    const int i = idx / 6, e = idx % 6;
    double s = 0;
    for (unsigned long long mm = sub[i]; mm; mm &= mm - 1) { const int j = __builtin_ctzll(mm); s += oh[6 * j + e]; }
    Hc[idx] = s;
  }
  for (int i = tid; i < nj; i += nthr)
    st6(of + 6 * i, add6(mat6_mul(oY + 36 * i, ld6(oa + 6 * i)), fcross(ld6(ov + 6 * i), ld6(oh + 6 * i))));
  __syncthreads();
  for (int idx = tid; idx < 6 * nj; idx += nthr) {
    const int i = idx / 6, e = idx % 6;
    double s = 0;
    for (unsigned long long mm = sub[i]; mm; mm &= mm - 1) { const int j = __builtin_ctzll(mm); s += of[6 * j + e]; }
    Fc[idx] = s;
  }
  for (int kd = tid; kd < nv; kd += nthr) st6(U + 6 * kd, mat6_mul(Yc + 36 * dof_body[kd], ld6(J + 6 * kd)));
  __syncthreads();

  EV_PROF(1);
  if (has_dyn) {
    const double dt = P[desc[4]];
    // ---- P6: joint-space inertia, bias torques, contact frames -------------------------------------------
    for (int idx = tid; idx < nvp * nvp; idx += nthr) {
      const int r = qdiv(idx, S.mg_nvp), cc = (idx - qdiv(idx, S.mg_nvp) * nvp);
      double s = (r == cc) ? 1.0 : 0.0;  // identity padding
      if (r < nv && cc < nv) {
        s = 0;
        if (BELOW(cc, dof_body[r])) s = dot6(ld6(U + 6 * r), ld6(J + 6 * cc));
        else if (BELOW(r, dof_body[cc])) s = dot6(ld6(U + 6 * cc), ld6(J + 6 * r));
      }
      M[r * ldm + cc] = s;
    }
    for (int kd = tid; kd < nv; kd += nthr) bias[kd] = dot6(ld6(J + 6 * kd), ld6(Fc + 6 * dof_body[kd]));
    if (tid < nk) {
      const int cid = desc[2 + tid], i = mcontact[cid];
      const double* cm = cd + MPC_MODEL_CONTACT_DOUBLES * cid;
      const M3 Ri = ldm3(oR + 9 * i);
      const M3 Rc = mul(Ri, ldm3(cm));
      const V3 pc = mul(Ri, ldv3(cm + 9)) + ldv3(op + 3 * i);
      double* cf = cfr + 54 * tid;
      for (int e = 0; e < 9; ++e) cf[e] = Rc.m[e];
      cf[9] = pc.x; cf[10] = pc.y; cf[11] = pc.z;
      const M3 R2 = ldm3(cm + 12);
      const V3 p2 = ldv3(cm + 21);
      V3 ev, ew;
      log6(tmul(Rc, R2), tmul(Rc, p2 - pc), ev, ew);
      const S6 e6 = mk6(ev, ew);
      const S6 acb = adinv(Rc, pc, sub6(ld6(oa + 6 * i), a0));
      const S6 vcb = adinv(Rc, pc, ld6(ov + 6 * i));
      for (int r = 0; r < 6; ++r) gam[6 * tid + r] = acb.v[r] + cm[30 + r] * vcb.v[r] - cm[24 + r] * e6.v[r];
      if (derivs) Jlog6(tmul(R2, Rc), tmul(R2, pc - p2), cf + 12);  // Jlog6(c2Mc1)
    }
    __syncthreads();
    for (int idx = tid; idx < nk * nv; idx += nthr) {
      const int cc = qdiv(idx, S.mg_nv), kd = (idx - qdiv(idx, S.mg_nv) * nv);
      const int i = mcontact[desc[2 + cc]];
      S6 col = zero6();
      if (BELOW(kd, i)) col = adinv(ldm3(cfr + 54 * cc), ldv3(cfr + 54 * cc + 9), ld6(J + 6 * kd));
      for (int r = 0; r < 6; ++r) Jc[(6 * cc + r) * nv + kd] = col.v[r];
    }
    __syncthreads();
    // Y16 = [Jc^T | r1 | 0]  (nvp x 16): the contact columns and the dynamics right-hand side r1 = B u - bias
    for (int idx = tid; idx < nvp * 16; idx += nthr) {
      const int l = idx >> 4, j = idx & 15;
      double s = 0.0;
      if (l < nv) {
        if (j < nl) s = Jc[j * nv + l];
        else if (j == 12) s = -bias[l] + (l >= nv - nu ? u[l - (nv - nu)] : 0.0);
      }
      Y16[idx] = s;
    }
    __syncthreads();
    EV_PROF(2);
    // ---- P7: M = L L^T ; Y = L^-1 [Jc^T | r1] ; S = Y^T Y + mu I = Ls Ls^T ; multipliers ; accelerations -------
    if (tid == 0) iflag[1] = 1;
    if (!chol_blocked(M, ldm, nbm, LIm, tid, iflag)) { if (tid == 0) a.inst[b].done = 5; return; }
    EV_PROF(3);
    if (wv == 0) {
      trsm_fwd_blocked(M, ldm, LIm, nbm, Y16, 16, 1, 0, 1, lane);
      d4_t g = d4_t{0, 0, 0, 0};
      mma_tile<false>(g, Y16, 1, 16, Y16, 16, 1, nvp, lane);  // [Y w]^T [Y w]
      const int col = lane & 15;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 4) + 4 * q;
        double sv = (row < nl && col < nl) ? g[q] : 0.0;
        if (row == col) sv += (row < nl) ? prox_mu : 1.0;
        Sp[row * 17 + col] = sv;
        if (col == 12 && row < nl) small[row] = g[q] + gam[row];  // t = Y^T w - r2,  r2 = -gamma
      }
      if (!chol16_wave(Sp, 17, LIs, lane) && lane == 0) iflag[1] = 0;
      // z2 = Ls^-T Ls^-1 t ; lambda = -z2
      if (lane < 16) {
        double y = 0;
        for (int j = 0; j <= lane; ++j) y += LIs[lane * 17 + j] * ((j < nl) ? small[j] : 0.0);
        small[16 + lane] = y;
      }
      if (lane < 16) {
        double z = 0;
        for (int j = lane; j < 16; ++j) z += LIs[j * 17 + lane] * small[16 + j];
        small[32 + lane] = z;
        if (lane < nl) lam[lane] = -z;
      }
    }
    __syncthreads();
    if (iflag[1] == 0) { if (tid == 0) a.inst[b].done = 6; return; }
    // V16 column 0 = w - Y z2, then accelerations = L^-T (.)
    for (int idx = tid; idx < nvp * 16; idx += nthr) {
      const int l = idx >> 4, j = idx & 15;
      double s = 0.0;
      if (j == 0) { s = Y16[l * 16 + 12]; for (int i = 0; i < nl; ++i) s -= Y16[l * 16 + i] * small[32 + i]; }
      V16[idx] = s;
    }
    __syncthreads();
    if (wv == 0) trsm_bwd_blocked(M, ldm, LIm, nbm, V16, 16, 1, 0, 1, lane);
    __syncthreads();
    for (int i = tid; i < nv; i += nthr) acc[i] = V16[i * 16];
    for (int l = tid; l < nvp; l += nthr) Y16[l * 16 + 12] = 0.0;  // from here on Y16 = Y (zero padded)
    __syncthreads();
    EV_PROF(5);
    if (derivs) {
      for (int i = tid; i < n; i += nthr) kn[KL.oXD + i] = (i < nv) ? v[i] : acc[i - nv];
      for (int i = tid; i < 12; i += nthr) kn[KL.oWR + i] = 0.0;
      __syncthreads();
      for (int i = tid; i < nl; i += nthr) kn[KL.oWR + 6 * desc[2 + i / 6] + i % 6] = lam[i];
    }
    if (derivs) {  // body accelerations at the solution: only the derivative blocks read them
      for (int idx = tid; idx < 6 * nj; idx += nthr) {
        const int i = idx / 6, e = idx % 6;
        double s = oa[idx];
        for (unsigned long long mm = dmask[i]; mm; mm &= mm - 1) { const int kd = __builtin_ctzll(mm); s += J[6 * kd + e] * acc[kd]; }
        oa[idx] = s;
      }
      __syncthreads();
    }
    if (derivs) {
      for (int i = tid; i < nj; i += nthr) {
        S6 f = add6(mat6_mul(oY + 36 * i, ld6(oa + 6 * i)), fcross(ld6(ov + 6 * i), ld6(oh + 6 * i)));
        for (int cc = 0; cc < nk; ++cc) {
          if (mcontact[desc[2 + cc]] != i) continue;
          const M3 Rc = ldm3(cfr + 54 * cc);
          const V3 pc = ldv3(cfr + 54 * cc + 9);
          const V3 fl = mul(Rc, v3(lam[6 * cc], lam[6 * cc + 1], lam[6 * cc + 2]));
          const V3 fa = mul(Rc, v3(lam[6 * cc + 3], lam[6 * cc + 4], lam[6 * cc + 5])) + cross(pc, fl);
          f = sub6(f, mk6(fl, fa));
        }
        st6(of + 6 * i, f);
      }
      __syncthreads();
      for (int idx = tid; idx < 6 * nj; idx += nthr) {
        const int i = idx / 6, e = idx % 6;
        double s = 0;
        for (unsigned long long mm = sub[i]; mm; mm &= mm - 1) { const int j = __builtin_ctzll(mm); s += of[6 * j + e]; }
        Fc[idx] = s;
      }
    }
    __syncthreads();
    (void)dt;
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
    return;
  }
  // ---- kinodynamics (kinodynamic_talos.py:107-112): a_joint = u[12:], base acceleration from the momentum balance
  // about the world origin  sum_k U_k a_k + hdot(a = 0) = [sum f + m g ; sum p_i x f_i + tau_i + c x m g],
  // projected on the base columns: (J_b^T U_b) a_b = J_b^T (...)  with J_b^T U_b = M_bb symmetric positive definite.
  double mtot = 0;
  for (int i = 0; i < nj; ++i) mtot += jd[25 * i + 12];
  const V3 com = v3(Yc[6 * 5 + 1] / mtot, Yc[6 * 3 + 2] / mtot, Yc[6 * 4 + 0] / mtot);
  if (kino) {
    const double* dp = P + desc[4];
    const int nkk = desc[1], nf = 6 * nkk;
    const V3 mg = v3(mtot * dp[1], mtot * dp[2], mtot * dp[3]);
    double* Minv6 = small + 144;  // 36
    if (tid < nkk) {
      const int fi = (int)dp[4 + tid], i = mframe[fi];
      const V3 pf = mul(ldm3(oR + 9 * i), ldv3(fd + 12 * fi + 9)) + ldv3(op + 3 * i);
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
      r0 = sub6(r0, ld6(Fc));  // hdot at a = 0 (true accelerations: a0 = 0 in this mode)
      for (int j = 6; j < nv; ++j) r0 = sub6(r0, scale6(acc[j], ld6(U + 6 * j)));
      double Mbb[36], rb[6];
      for (int r = 0; r < 6; ++r) {
        rb[r] = dot6(ld6(J + 6 * r), r0);
        for (int cc = 0; cc < 6; ++cc) Mbb[6 * r + cc] = dot6(ld6(U + 6 * r), ld6(J + 6 * cc));
      }
      inv6_unrolled_mb(Mbb, Minv6);
      for (int r = 0; r < 6; ++r) { double sacc = 0; for (int cc = 0; cc < 6; ++cc) sacc += Minv6[6 * r + cc] * rb[cc]; acc[r] = sacc; }
    }
    __syncthreads();
    if (derivs) {
      for (int i = tid; i < n; i += nthr) kn[KL.oXD + i] = (i < nv) ? v[i] : acc[i - nv];
      for (int i = tid; i < 12; i += nthr) kn[KL.oWR + i] = 0.0;
    }
    // accelerations and body forces at the solution (needed by the derivative vectors)
    for (int idx = tid; idx < 6 * nj; idx += nthr) {
      const int i = idx / 6, e = idx % 6;
      double sacc = oa[idx];
      for (unsigned long long mm = dmask[i]; mm; mm &= mm - 1) { const int kd = __builtin_ctzll(mm); sacc += J[6 * kd + e] * acc[kd]; }
      oa[idx] = sacc;
    }
    __syncthreads();
    if (derivs) {
      for (int i = tid; i < nj; i += nthr) st6(of + 6 * i, add6(mat6_mul(oY + 36 * i, ld6(oa + 6 * i)), fcross(ld6(ov + 6 * i), ld6(oh + 6 * i))));
      __syncthreads();
      for (int idx = tid; idx < 6 * nj; idx += nthr) {
        const int i = idx / 6, e = idx % 6;
        double sacc = 0;
        for (unsigned long long mm = sub[i]; mm; mm &= mm - 1) { const int j = __builtin_ctzll(mm); sacc += of[6 * j + e]; }
        Fc[idx] = sacc;
      }
    }
    __syncthreads();
  }

  // ---- P9: derivative building blocks (Jacobians only: the value-only pass skips them) ----------------------
  if (derivs) for (int kd = tid; kd < nv; kd += nthr) {
    const int pb = parent[dof_body[kd]];
    const S6 vl = (pb >= 0) ? ld6(ov + 6 * pb) : zero6();
    const S6 al = (pb >= 0) ? ld6(oa + 6 * pb) : a0;
    st6(vlam + 12 * kd, vl);
    st6(vlam + 12 * kd + 6, al);
    const S6 Jk = ld6(J + 6 * kd);
    const S6 psd = mcross(vl, Jk);
    st6(Psd + 6 * kd, psd);
    st6(Psdd + 6 * kd, add6(mcross(al, Jk), mcross(vl, psd)));
    st6(Phi + 6 * kd, mcross(add6(ld6(ov + 6 * dof_body[kd]), vl), Jk));
  }
  __syncthreads();
  if (derivs && (has_dyn || kino)) {
    // body-level "Coriolis" matrices B_i (overwrite oY), then their subtree sums
    for (int i = tid; i < nj; i += nthr) {
      double Yl[36], Bm[36];
      for (int e = 0; e < 36; ++e) Yl[e] = oY[36 * i + e];
      const S6 vi = ld6(ov + 6 * i), hi = ld6(oh + 6 * i);
      for (int col = 0; col < 6; ++col) {
        S6 e6 = zero6();
        e6.v[col] = 1.0;
        const S6 r = add6(add6(mat6_mul(Yl, mcross(e6, vi)), fcross(e6, hi)), fcross(vi, mat6_mul(Yl, e6)));
        for (int row = 0; row < 6; ++row) Bm[6 * row + col] = r.v[row];
      }
      for (int e = 0; e < 36; ++e) oY[36 * i + e] = Bm[e];
    }
    __syncthreads();
    for (int idx = tid; idx < 36 * nj; idx += nthr) {
      const int i = idx / 36, e = idx % 36;
      double s = 0;
      for (unsigned long long mm = sub[i]; mm; mm &= mm - 1) { const int j = __builtin_ctzll(mm); s += oY[36 * j + e]; }
      Bc[idx] = s;
    }
    __syncthreads();
    for (int kd = tid; kd < nv; kd += nthr) {
      const int bk = dof_body[kd];
      const S6 Jk = ld6(J + 6 * kd);
      st6(Bt + 6 * kd, mat6_tmul(Bc + 36 * bk, Jk));
      st6(Tq + 6 * kd, add6(add6(mat6_mul(Yc + 36 * bk, ld6(Psdd + 6 * kd)), mat6_mul(Bc + 36 * bk, ld6(Psd + 6 * kd))), fcross(Jk, ld6(Fc + 6 * bk))));
      st6(Tv + 6 * kd, add6(mat6_mul(Yc + 36 * bk, ld6(Phi + 6 * kd)), mat6_mul(Bc + 36 * bk, Jk)));
    }
    __syncthreads();
    if (kino) {
      // d r0 / d(q, v, u) (6 x nz, stored in dr), then  d a_b = -Mbb^-1 J_b^T d r0 ; joint accelerations are controls
      const double* dp = P + desc[4];
      const int nkk = desc[1], nf = 6 * nkk;
      const V3 mg = v3(mtot * dp[1], mtot * dp[2], mtot * dp[3]);
      const double* Minv6 = small + 144;
      for (int z = tid; z < nz; z += nthr) {
        S6 col = zero6();
        if (z < nv) {
          col = ld6(Tq + 6 * z);
          V3 dang = cross((1.0 / mtot) * lin(ld6(U + 6 * z)), mg);
          const S6 Jz = ld6(J + 6 * z);
          for (int cc = 0; cc < nkk; ++cc) {
            if (!desc[2 + cc] || !BELOW(z, (int)cfr[54 * cc])) continue;
            const V3 pf = ldv3(cfr + 54 * cc + 9);
            dang = dang + cross(lin(Jz) + cross(ang(Jz), pf), v3(u[6 * cc], u[6 * cc + 1], u[6 * cc + 2]));
          }
          col = sub6(col, mk6(v3(0, 0, 0), dang));
        } else if (z < n) {
          col = ld6(Tv + 6 * (z - nv));
        } else if (z < n + nf) {
          const int cc = (z - n) / 6, e = (z - n) % 6;
          if (desc[2 + cc]) {
            V3 ev = v3(e % 3 == 0 ? 1.0 : 0.0, e % 3 == 1 ? 1.0 : 0.0, e % 3 == 2 ? 1.0 : 0.0);
            if (e < 3) col = mk6(v3(-ev.x, -ev.y, -ev.z), cross(ev, ldv3(cfr + 54 * cc + 9)));  // -[I ; p x]
            else col = mk6(v3(0, 0, 0), v3(-ev.x, -ev.y, -ev.z));
          }
        } else {
          col = ld6(U + 6 * (6 + z - n - nf));
        }
        double jb[6];
        for (int r = 0; r < 6; ++r) jb[r] = dot6(ld6(J + 6 * r), col);
        for (int r = 0; r < 6; ++r) {
          double sacc = 0;
          for (int cc = 0; cc < 6; ++cc) sacc += Minv6[6 * r + cc] * jb[cc];
          dsol[(size_t)r * L.nz + z] = -sacc;
        }
        for (int r = 6; r < nv; ++r) dsol[(size_t)r * L.nz + z] = (z == n + nf + r - 6) ? 1.0 : 0.0;
      }
      __syncthreads();
    }
    EV_PROF(7);
    // ---- P10: R = [d r1 ; d r2] w.r.t. (q, v, u), zero padded (R overwrites the body-level blocks oY / Bc) ----
    const int n2 = 2 * nv;
    if (has_dyn) {
    for (int idx = tid; idx < (nvp + 16) * ldR; idx += nthr) Rm[idx] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < nv * nv; idx += nthr) {
      const int r = qdiv(idx, S.mg_nv), j = (idx - qdiv(idx, S.mg_nv) * nv);
      const int br = dof_body[r], bj = dof_body[j];
      double dq = 0, dv = 0;
      if ((anc[br] >> bj) & 1ull) {
        const S6 Ur = ld6(U + 6 * r), Btr = ld6(Bt + 6 * r);
        dq = dot6(Ur, ld6(Psdd + 6 * j)) + dot6(Btr, ld6(Psd + 6 * j));
        dv = dot6(Ur, ld6(Phi + 6 * j)) + dot6(Btr, ld6(J + 6 * j));
      } else if ((anc[bj] >> br) & 1ull) {
        const S6 Jr = ld6(J + 6 * r);
        dq = dot6(Jr, ld6(Tq + 6 * j));
        dv = dot6(Jr, ld6(Tv + 6 * j));
      }
      R1[r * ldR + j] = dq;
      R1[r * ldR + nv + j] = dv;
    }
    for (int i = tid; i < nu; i += nthr) R1[(nv - nu + i) * ldR + n2 + i] = -1.0;  // d r1 / du = -B
    for (int idx = tid; idx < nk * nv; idx += nthr) {
      const int cc = qdiv(idx, S.mg_nv), j = (idx - qdiv(idx, S.mg_nv) * nv);
      const int cid = desc[2 + cc], i = mcontact[cid];
      const double* cm = cd + MPC_MODEL_CONTACT_DOUBLES * cid;
      S6 rq = zero6(), rv = zero6();
      if (BELOW(j, i)) {
        const M3 Rc = ldm3(cfr + 54 * cc);
        const V3 pc = ldv3(cfr + 54 * cc + 9);
        const S6 Jj = ld6(J + 6 * j), vl = ld6(vlam + 12 * j), al = ld6(vlam + 12 * j + 6), psd = ld6(Psd + 6 * j);
        const S6 w = sub6(ld6(ov + 6 * i), vl);
        const S6 dacq = adinv(Rc, pc, add6(mcross(sub6(al, a0), Jj), mcross(psd, w)));
        const S6 dacv = adinv(Rc, pc, add6(mcross(ld6(ov + 6 * dof_body[j]), Jj), mcross(Jj, w)));
        const S6 apsd = adinv(Rc, pc, psd);
        S6 Jcj;
        for (int r = 0; r < 6; ++r) Jcj.v[r] = Jc[(6 * cc + r) * nv + j];
        const S6 jl = mat6_mul(cfr + 54 * cc + 12, Jcj);
        for (int r = 0; r < 6; ++r) {
          rq.v[r] = dacq.v[r] + cm[30 + r] * apsd.v[r] + cm[24 + r] * jl.v[r];
          rv.v[r] = dacv.v[r] + cm[30 + r] * Jcj.v[r];
        }
      }
      for (int r = 0; r < 6; ++r) { R2[(6 * cc + r) * ldR + j] = rq.v[r]; R2[(6 * cc + r) * ldR + nv + j] = rv.v[r]; }
    }
    __syncthreads();
    EV_PROF(8);
    // ---- P11: implicit differentiation, one 16-column block per wavefront (no workgroup barrier inside):
    //   W = L^-1 R1 ; T = Y^T W - R2 ; Z2 = S^-1 T ; Z1 = L^-T (W - Y Z2) ;  d a = -Z1 ,  d lambda = Z2
    trsm_fwd_blocked(M, ldm, LIm, nbm, R1, ldR, ncb, wv, nw, lane);
    for (int cj = wv; cj < ncb; cj += nw) {
      d4_t t = tile_load(R2 + cj * 16, ldR, lane);
      t = -t;
      mma_tile<false>(t, Y16, 1, 16, R1 + cj * 16, ldR, 1, nvp, lane);
      tile_store(R2 + cj * 16, ldR, t, lane);
    }
    trsm_fwd_blocked(Sp, 17, LIs, 1, R2, ldR, ncb, wv, nw, lane);
    trsm_bwd_blocked(Sp, 17, LIs, 1, R2, ldR, ncb, wv, nw, lane);
    for (int cj = wv; cj < ncb; cj += nw)
      for (int bi = 0; bi < nbm; ++bi) {
        double* Wt = R1 + (bi * 16) * ldR + cj * 16;
        d4_t w = tile_load(Wt, ldR, lane);
        mma_tile<true>(w, Y16 + (bi * 16) * 16, 16, 1, R2 + cj * 16, ldR, 1, 16, lane);
        tile_store(Wt, ldR, w, lane);
      }
    trsm_bwd_blocked(M, ldm, LIm, nbm, R1, ldR, ncb, wv, nw, lane);
    __syncthreads();  // d a = -R1, d lambda = R2 stay in LDS for the integrator and the force terms
    }  // has_dyn
  }

  EV_PROF(9);
  // ---- SE(3)-valued terms: residual and Jacobian block, one term per wavefront lane 0 (waves 2..) ---------------
  if (wv >= 2 && lane == 0) {
    int slot = 0;
    for (int t = 0; t < nterms; ++t) {
      const TermRec tr = lds_term(lterm, t);
      const bool se3_state = tr.type == MPC_TERM_STATE_ERROR && tr.i0 < 6;
      if (!se3_state && tr.type != MPC_TERM_FRAME_PLACEMENT) continue;
      const int my = slot++;
      if (my >= MB_SE3_SLOTS || (my % (nw - 2)) != wv - 2) continue;
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
        const M3 Ri = ldm3(oR + 9 * i);
        const M3 Rf = mul(Ri, ldm3(fd + 12 * fi));
        const V3 pf = mul(Ri, ldv3(fd + 12 * fi + 9)) + ldv3(op + 3 * i);
        const M3 Rr = ldm3(tp);
        const V3 pr = ldv3(tp + 9);
        log6(tmul(Rr, Rf), tmul(Rr, pf - pr), ev, ew);
        if (derivs) Jlog6(tmul(Rr, Rf), tmul(Rr, pf - pr), sl + 8);
      }
      sl[0] = ev.x; sl[1] = ev.y; sl[2] = ev.z; sl[3] = ew.x; sl[4] = ew.y; sl[5] = ew.z;
    }
  }
  // ---- P12: semi-implicit Euler, gap and its Jacobians ----------------------------------------------------
  if (has_dyn || kino) {
    const double dt = P[desc[4]];
    double* Jl6 = small;        // Jlog6(G)
    double* Je6 = small + 36;   // Jexp6(delta)
    double* Jq6 = small + 72;   // Ad(exp6(delta))^-1
    // The SE(3) pieces are single-lane work (log / exp maps and their Jacobians): spread them over the wavefronts
    // — wave 0: step, gap, Jlog6(G) ; wave 1: Jexp6, Ad^-1, E6 ; waves 2..: the SE(3)-valued cost / constraint terms
    if (tid == 0 || (tid == 64 && derivs)) {
      const V3 dl = v3(dt * (v[0] + dt * acc[0]), dt * (v[1] + dt * acc[1]), dt * (v[2] + dt * acc[2]));
      const V3 da_ = v3(dt * (v[3] + dt * acc[3]), dt * (v[4] + dt * acc[4]), dt * (v[5] + dt * acc[5]));
      M3 dR; V3 dp;
      exp6(dl, da_, dR, dp);
      const M3 Rb = quat_to_rot(q + 3);
      const M3 Rn = mul(Rb, dR);
      const V3 pn = mul(Rb, dp) + v3(q[0], q[1], q[2]);
      const M3 Rt = quat_to_rot(xn + 3);
      const M3 GR = tmul(Rt, Rn);
      const V3 Gp = tmul(Rt, pn - v3(xn[0], xn[1], xn[2]));
      if (tid == 0) {
        if (derivs) { kn[KL.oXN] = pn.x; kn[KL.oXN + 1] = pn.y; kn[KL.oXN + 2] = pn.z; rot_to_quat(Rn, kn + KL.oXN + 3); }
        V3 gv, gw;
        log6(GR, Gp, gv, gw);
        kn[KL.oF] = gv.x; kn[KL.oF + 1] = gv.y; kn[KL.oF + 2] = gv.z; kn[KL.oF + 3] = gw.x; kn[KL.oF + 4] = gw.y; kn[KL.oF + 5] = gw.z;
        if (derivs) Jlog6(GR, Gp, Jl6);
      } else {
        Jexp6(dl, da_, Je6);
        // Ad(exp6(delta))^-1 = [[dR^T, -dR^T [dp]x],[0, dR^T]]
        const M3 Sx = skew_m(dp);
        const M3 RtS = tmul(dR, Sx);
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) {
          Jq6[6 * r + cc] = dR.m[3 * cc + r]; Jq6[6 * (r + 3) + cc + 3] = dR.m[3 * cc + r];
          Jq6[6 * r + cc + 3] = -RtS.m[3 * r + cc]; Jq6[6 * (r + 3) + cc] = 0.0;
        }
        // E6 = -Jlog6(G^-1)
        double E[36];
        M3 Gi;
        for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) Gi.m[3 * r + cc] = GR.m[3 * cc + r];
        const V3 gip = mul(Gi, v3(-Gp.x, -Gp.y, -Gp.z));
        Jlog6(Gi, gip, E);
        for (int e = 0; e < 36; ++e) kn[KL.oE6 + e] = -E[e];
      }
    }
    for (int i = 6 + tid; i < n; i += nthr) {
      if (i < nv) { const double vp = v[i] + dt * acc[i]; kn[KL.oF + i] = q[i + 1] + dt * vp - xn[i + 1]; if (derivs) kn[KL.oXN + i + 1] = q[i + 1] + dt * vp; }
      else if (i >= nv) { const int j = i - nv; const double vp = v[j] + dt * acc[j]; kn[KL.oF + i] = vp - xn[nq + j]; if (derivs) kn[KL.oXN + nq + j] = vp; }
    }
    __syncthreads();
    if (derivs) {
      // dvp = dt * da + [0 I 0];  rows nv..n of AB = dvp; rows 6..nv = dt dvp + [I 0 0]; base rows below
      for (int idx = tid; idx < nv * nz; idx += nthr) {
        const int r = qdiv(idx, mg_nz), z = (idx - qdiv(idx, mg_nz) * nz);
        const double da_rz = has_dyn ? -R1[r * ldR + z] : dsol[(size_t)r * L.nz + z];
        const double dvp = dt * da_rz + ((z == nv + r) ? 1.0 : 0.0);
        kn[KL.oAB + (size_t)(nv + r) * KL.nz + z] = dvp;
        if (r >= 6) kn[KL.oAB + (size_t)r * KL.nz + z] = dt * dvp + ((z == r) ? 1.0 : 0.0);
      }
      __syncthreads();  // JL aliases rows of R1
      // base rows: Jlog6(G) ( dt Jexp6 dvp[0:6] + [Jq6 0] )
      for (int idx = tid; idx < 6 * nz; idx += nthr) {
        const int r = qdiv(idx, mg_nz), z = (idx - qdiv(idx, mg_nz) * nz);
        double s = (z < 6) ? Jq6[6 * r + z] : 0.0;
        for (int l = 0; l < 6; ++l) {
          const double da_lz = has_dyn ? -R1[l * ldR + z] : dsol[(size_t)l * L.nz + z];
          s += dt * Je6[6 * r + l] * (dt * da_lz + ((z == nv + l) ? 1.0 : 0.0));
        }
        JL[idx] = s;  // temporary (6 x nz)
      }
      __syncthreads();
      for (int idx = tid; idx < 6 * nz; idx += nthr) {
        const int r = qdiv(idx, mg_nz), z = (idx - qdiv(idx, mg_nz) * nz);
        double s = 0;
        for (int l = 0; l < 6; ++l) s += Jl6[6 * r + l] * JL[l * nz + z];
        kn[KL.oAB + (size_t)r * KL.nz + z] = s;
      }
      // the same rows in factored form (layout.h, oD12): D1_b = Jlog6(G) Jq6, Dd_b = dt Jlog6(G) Jexp6
      if (tid < 72) {
        const int e = tid % 36, r = e / 6, cc = e % 6;
        const double* Rm = (tid < 36) ? Jq6 : Je6;
        double s = 0;
        for (int l = 0; l < 6; ++l) s += Jl6[6 * r + l] * Rm[6 * l + cc];
        kn[KL.oD12 + tid] = (tid < 36) ? s : dt * s;
      } else if (tid == 72) { kn[KL.oD12 + 72] = dt; kn[KL.oD12 + 73] = 1.0; }
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
  const S6 h0 = ld6(Hc);
  double* gacc = Bt;     // nz: gradient accumulator        (the derivative vectors Bt, Tv, Phi, Tq are dead by now)
  double* hdg = Tv;      // nz: additions to diag(H)
  double* hbb = Phi;     // 36: additions to the base 6x6 block of H
  double* tcost = small; // per-term cost, summed in term order at the end (deterministic)
  int* tkind = (int*)(small + 32);  // 0 workgroup pass, 1 stacked cost (wave pass), 2 dense-weight cost (HBM pass)
  int* trow = tkind + 24;           // first stack row of a stacked term
  int* tse3 = tkind + 48;           // slot in the SE(3) table
  int* tchunk = tkind + 72;         // stack chunk (32 rows each)
  int* tmeta = tkind + 96;          // [0] number of chunks, [1] any dense-weight cost
  for (int z = tid; z < nz; z += nthr) { gacc[z] = 0.0; hdg[z] = a.opt.reg_init; }
  for (int i = tid; i < 36; i += nthr) hbb[i] = 0.0;
  for (int i = tid; i < 24; i += nthr) tcost[i] = 0.0;
  if (tid == 0) {
    int rows = 0, chunk = 0, se3n = 0, dense = 0, nst = 0;
    for (int t = 0; t < nterms; ++t) {
      const TermRec tr = lds_term(lterm, t);
      const bool se3t = tr.type == MPC_TERM_FRAME_PLACEMENT || (tr.type == MPC_TERM_STATE_ERROR && tr.i0 < 6);
      tse3[t] = se3t ? se3n++ : 0;
      const bool diag_sel = (tr.type == MPC_TERM_STATE_ERROR || tr.type == MPC_TERM_CONTROL_ERROR) && (tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT);
      int kind = 0;
      if (tr.role == MPC_ROLE_COST && !diag_sel) {
        bool wdiag = (tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT) != 0;
        if (!wdiag) { const double* W = P + tr.woff; wdiag = true; for (int e = 0; e < tr.dim * tr.dim; ++e) if ((e / tr.dim != e % tr.dim) && W[e] != 0.0) wdiag = false; }
        if (wdiag && tr.dim <= 24) {
          kind = 1;
          if (rows + tr.dim > 32) { ++chunk; rows = 0; }
          trow[t] = rows; tchunk[t] = chunk; rows += tr.dim; ++nst;
        } else { kind = 2; dense = 1; }
      }
      tkind[t] = kind;
    }
    tmeta[0] = nst ? chunk + 1 : 0; tmeta[1] = dense;
  }
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
    if (derivs) for (int idx = t0; idx < d * nz; idx += nt) Jt[idx] = 0.0;
    if (wg) __syncthreads();
    if (tr.type == MPC_TERM_STATE_ERROR) {
      const double* Jb = sl + 8;  // -Jlog6 block of the base rows
      for (int i = t0; i < d; i += nt) r[i] = state_res(tp, sl, tr.i0 + i);
      if (derivs) for (int i = t0; i < d; i += nt) {
        const int ri = tr.i0 + i;
        if (ri < 6) { for (int z = 0; z < 6; ++z) Jt[i * nz + z] = Jb[6 * ri + z]; }
        else Jt[i * nz + ri] = -1.0;
      }
    } else if (tr.type == MPC_TERM_CONTROL_ERROR) {
      for (int i = t0; i < d; i += nt) { r[i] = u[tr.i0 + i] - tp[tr.i0 + i]; if (derivs) Jt[i * nz + n + tr.i0 + i] = 1.0; }
    } else if (tr.type == MPC_TERM_FRAME_PLACEMENT || tr.type == MPC_TERM_FRAME_TRANSLATION || tr.type == MPC_TERM_FRAME_VELOCITY) {
      const int fi = tr.i0, i = mframe[fi];
      const M3 Ri = ldm3(oR + 9 * i);
      const M3 Rf = mul(Ri, ldm3(fd + 12 * fi));
      const V3 pf = mul(Ri, ldv3(fd + 12 * fi + 9)) + ldv3(op + 3 * i);
      if (tr.type == MPC_TERM_FRAME_PLACEMENT) {
        const double* Jl = sl + 8;
        if (t0 < 6) r[t0] = sl[t0];
        if (derivs) for (int j = t0; j < nv; j += nt) if (BELOW(j, i)) {
          const S6 col = mat6_mul(Jl, adinv(Rf, pf, ld6(J + 6 * j)));
          for (int rr = 0; rr < 6; ++rr) Jt[rr * nz + j] = col.v[rr];
        }
      } else if (tr.type == MPC_TERM_FRAME_TRANSLATION) {
        if (t0 < d) { const double pfa[3] = {pf.x, pf.y, pf.z}; r[t0] = pfa[tr.i1 + t0] - tp[tr.i1 + t0]; }
        if (derivs) for (int j = t0; j < nv; j += nt) if (BELOW(j, i)) {
          const S6 Jj = ld6(J + 6 * j);
          const V3 lv = lin(Jj) + cross(ang(Jj), pf);
          const double la[3] = {lv.x, lv.y, lv.z};
          for (int rr = 0; rr < d; ++rr) Jt[rr * nz + j] = la[tr.i1 + rr];
        }
      } else {
        if (t0 == 0) { const S6 vf = adinv(Rf, pf, ld6(ov + 6 * i)); for (int rr = 0; rr < 6; ++rr) r[rr] = vf.v[rr] - tp[rr]; }
        if (derivs) for (int j = t0; j < nv; j += nt) if (BELOW(j, i)) {
          const S6 cq = adinv(Rf, pf, ld6(Psd + 6 * j)), cv = adinv(Rf, pf, ld6(J + 6 * j));
          for (int rr = 0; rr < 6; ++rr) { Jt[rr * nz + j] = cq.v[rr]; Jt[rr * nz + nv + j] = cv.v[rr]; }
        }
      }
    } else if (tr.type == MPC_TERM_COM_TRANSLATION) {
      if (t0 < d) { const double ca[3] = {com.x, com.y, com.z}; r[t0] = ca[tr.i1 + t0] - tp[tr.i1 + t0]; }
      if (derivs) for (int j = t0; j < nv; j += nt) for (int rr = 0; rr < d; ++rr) Jt[rr * nz + j] = U[6 * j + tr.i1 + rr] / mtot;
    } else if (tr.type == MPC_TERM_CENTROIDAL_MOMENTUM) {
      if (t0 == 0) {
        const V3 hl = lin(h0), ha = ang(h0) - cross(com, lin(h0));
        r[0] = hl.x - tp[0]; r[1] = hl.y - tp[1]; r[2] = hl.z - tp[2]; r[3] = ha.x - tp[3]; r[4] = ha.y - tp[4]; r[5] = ha.z - tp[5];
      }
      if (derivs) for (int j = t0; j < nv; j += nt) {
        const int bj = dof_body[j];
        const S6 Uj = ld6(U + 6 * j);
        const S6 D = add6(fcross(ld6(J + 6 * j), ld6(Hc + 6 * bj)), mat6_mul(Yc + 36 * bj, ld6(Psd + 6 * j)));
        const V3 dc = (1.0 / mtot) * lin(Uj);
        const V3 dql = lin(D), dqa = ang(D) - cross(dc, lin(h0)) - cross(com, lin(D));
        const V3 dvl = lin(Uj), dva = ang(Uj) - cross(com, lin(Uj));
        const double cq[6] = {dql.x, dql.y, dql.z, dqa.x, dqa.y, dqa.z}, cv[6] = {dvl.x, dvl.y, dvl.z, dva.x, dva.y, dva.z};
        for (int rr = 0; rr < 6; ++rr) { Jt[rr * nz + j] = cq[rr]; Jt[rr * nz + nv + j] = cv[rr]; }
      }
    } else if (tr.type == MPC_TERM_CONTACT_FORCE) {
      if (t0 < 6) r[t0] = lam[6 * tr.i0 + t0] - tp[t0];
      if (derivs) for (int idx = t0; idx < 6 * nz; idx += nt) Jt[idx] = R2[(6 * tr.i0 + qdiv(idx, mg_nz)) * ldR + (idx - qdiv(idx, mg_nz) * nz)];
    } else if (tr.type == MPC_TERM_CENTROIDAL_WRENCH_CONE) {
      for (int i = t0; i < d; i += nt) { double sacc = 0; for (int j = 0; j < 6; ++j) sacc += tp[i * 6 + j] * u[6 * tr.i0 + j]; r[i] = sacc; }
      if (derivs) for (int idx = t0; idx < d * 6; idx += nt) Jt[(idx / 6) * nz + n + 6 * tr.i0 + idx % 6] = tp[idx];
    } else if (tr.type == MPC_TERM_CENTROIDAL_MOMENTUM_DER) {
      // r = [sum f + m g ; sum (p_i - c) x f_i + tau_i]   (kinodynamic_talos.py:125-127); params: g[3], states, frames
      const int nkk = tr.i0;
      if (t0 == 0) {
        V3 rl = v3(mtot * tp[0], mtot * tp[1], mtot * tp[2]), ra = v3(0, 0, 0);
        for (int cc = 0; cc < nkk; ++cc) {
          if (tp[3 + cc] == 0.0) continue;
          const int fi = (int)tp[3 + nkk + cc], i = mframe[fi];
          const V3 pf = mul(ldm3(oR + 9 * i), ldv3(fd + 12 * fi + 9)) + ldv3(op + 3 * i);
          const V3 f = v3(u[6 * cc], u[6 * cc + 1], u[6 * cc + 2]);
          rl = rl + f;
          ra = ra + cross(pf - com, f) + v3(u[6 * cc + 3], u[6 * cc + 4], u[6 * cc + 5]);
        }
        r[0] = rl.x; r[1] = rl.y; r[2] = rl.z; r[3] = ra.x; r[4] = ra.y; r[5] = ra.z;
      }
      if (derivs) {
        for (int j = t0; j < nv; j += nt) {
          V3 dang = v3(0, 0, 0);
          const S6 Jj = ld6(J + 6 * j);
          const V3 dc = (1.0 / mtot) * lin(ld6(U + 6 * j));
          for (int cc = 0; cc < nkk; ++cc) {
            if (tp[3 + cc] == 0.0) continue;
            const int fi = (int)tp[3 + nkk + cc], i = mframe[fi];
            const V3 pf = mul(ldm3(oR + 9 * i), ldv3(fd + 12 * fi + 9)) + ldv3(op + 3 * i);
            V3 dp_ = v3(0, 0, 0);
            if (BELOW(j, i)) dp_ = lin(Jj) + cross(ang(Jj), pf);
            dang = dang + cross(dp_ - dc, v3(u[6 * cc], u[6 * cc + 1], u[6 * cc + 2]));
          }
          Jt[3 * nz + j] = dang.x; Jt[4 * nz + j] = dang.y; Jt[5 * nz + j] = dang.z;
        }
        if (t0 < nkk && tp[3 + t0] != 0.0) {
          const int cc = t0, fi = (int)tp[3 + nkk + cc], i = mframe[fi];
          const V3 rr = mul(ldm3(oR + 9 * i), ldv3(fd + 12 * fi + 9)) + ldv3(op + 3 * i) - com;
          const M3 Rx = skew_m(rr);
          for (int e = 0; e < 3; ++e) {
            Jt[e * nz + n + 6 * cc + e] = 1.0;
            Jt[(3 + e) * nz + n + 6 * cc + 3 + e] = 1.0;
            for (int e2 = 0; e2 < 3; ++e2) Jt[(3 + e) * nz + n + 6 * cc + e2] = Rx.m[3 * e + e2];
          }
        }
      }
    } else if (tr.type == MPC_TERM_MB_WRENCH_CONE) {
      for (int i = t0; i < d; i += nt) { double s = 0; for (int j = 0; j < 6; ++j) s += tp[i * 6 + j] * lam[6 * tr.i0 + j]; r[i] = s; }
      if (derivs) for (int idx = t0; idx < d * nz; idx += nt) {
        const int i = qdiv(idx, mg_nz), z = (idx - qdiv(idx, mg_nz) * nz);
        double s = 0;
        for (int j = 0; j < 6; ++j) s += tp[i * 6 + j] * R2[(6 * tr.i0 + j) * ldR + z];
        Jt[idx] = s;
      }
    }
    if (wg) __syncthreads();
  };

  // ---- pass B: constraints and the diagonal state / control costs, term by term through the whole workgroup ----
  {
    int row = 0;
    double* r = red + 2 * 256 - 64;
    for (int t = 0; t < nterms; ++t) {
      const TermRec tr = lds_term(lterm, t);
      if (tkind[t] != 0) continue;
      const double* tp = P + tr.poff;
      const double* sl = se3 + 48 * tse3[t];
      const int d = tr.dim;
      const bool is_cost = tr.role == MPC_ROLE_COST;
      if (tr.type == MPC_TERM_STATE_ERROR && is_cost) {
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
      } else {
        term_rows(tr, tp, sl, r, JL, tid, nthr, true);
        emit_constraint(KL, kn, tr, P, row, r, JL, nz, nz, derivs, tid, nthr);
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
      double* r = red + 288 + 24 * wv;  // private residual scratch of this wavefront
      double* Jt = JS + trow[t] * nz;
      term_rows(tr, P + tr.poff, se3 + 48 * tse3[t], r, Jt, lane, 64, false);
      const double* W = P + tr.woff;
      const int wstride = (tr.flags & MPC_TERM_FLAG_DIAG_WEIGHT) ? 1 : d + 1;
      {
        double cst = (lane < d) ? W[lane * wstride] * r[lane] * r[lane] : 0.0;  // d <= 24
        cst = wave_sum(cst);
        if (lane == 0) tcost[t] = 0.5 * cst;
      }
      if (derivs) {
        if (lane < d) wrs[trow[t] + lane] = sqrt(W[lane * wstride]) * r[lane];
        for (int idx = lane; idx < d * nz; idx += 64) Jt[idx] *= sqrt(W[(qdiv(idx, mg_nz)) * wstride]);
      }
    }
    EV_PROF(28);
    if (derivs) {
      const int kc = (rowc + 3) & ~3;  // MFMA K granularity: zero rows up to a multiple of 4
      __syncthreads();
      for (int idx = tid; idx < (kc - rowc) * nz; idx += nthr) JS[rowc * nz + idx] = 0.0;
      __syncthreads();
      for (int z = tid; z < nz; z += nthr) {
        double g = 0;
        for (int i = 0; i < rowc; ++i) g += JS[i * nz + z] * wrs[i];
        gacc[z] += g;
      }
      // upper block triangle of 16x16 tiles, mirrored on the way out; the first chunk writes, later ones accumulate
      const int nzt = (nz + 15) >> 4;
      for (int t = wv; t < nzt * (nzt + 1) / 2; t += nw) {
        int ta = 0, rem = t;
        while (rem >= nzt - ta) { rem -= nzt - ta; ++ta; }
        const int tb = ta + rem;
        d4_t h = d4_t{0, 0, 0, 0};
        mma_tile<false>(h, JS + ta * 16, 1, nz, JS + tb * 16, nz, 1, kc, lane);
        const int zb = tb * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int za = ta * 16 + (lane >> 4) + 4 * q;
          if (za < nz && zb < nz) {
            double hv = h[q];
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
    double* r = red + 2 * 256 - 64;
    for (int t = 0; t < nterms; ++t) {
      if (tkind[t] != 2) continue;
      const TermRec tr = lds_term(lterm, t);
      term_rows(tr, P + tr.poff, se3 + 48 * tse3[t], r, JS, tid, nthr, true);
      if (derivs) for (int idx = tid; idx < tr.dim * nz; idx += nthr) JtG[idx] = JS[idx];
      __syncthreads();
      double cst = 0.0;
      accumulate_cost(KL, kn, tr, P + tr.woff, r, JtG, nz, nz, red, WJ, derivs, cst, tid, nthr);
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
#undef INSUB
}

static inline void launch_eval_multibody(hipStream_t stream, const SolverArgs& a, const Layout& LT, double* records, double* scratch,
                                         size_t scratch_stride, bool trial, int cand0 = 0, int ncand = 1, int sim_substeps = 0, double sim_dt = 0.0,
                                         bool with_derivs = false) {
  const Layout& L = a.L;
  MbArgs mb;
  mb.lds = make_mb_lds(L.nj, L.n / 2, L.nx - L.n / 2, L.m, L.nz);
  mb.scratch = scratch;
  mb.scratch_stride = scratch_stride;
  mb.sim_substeps = sim_substeps; mb.sim_dt = sim_dt;
  mb.ncand_loop = (trial && !with_derivs && ncand > 1) ? ncand : 0;
  if (mb.lds.total_bytes > 160 * 1024) throw std::runtime_error("multibody model too large for the LDS budget of the stage kernel");
  // hipFuncSetAttribute is a per-device setting: remember what was requested on every device (a process may hold handles on several
  // devices, driven from different threads)
  static std::atomic<int> attr_bytes_dev[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& attr_bytes = attr_bytes_dev[dev & 63];
  if (attr_bytes.load() != mb.lds.total_bytes + 1) {
    // the kernel also owns a few bytes of static LDS, so request exactly what the carve-out needs
    hipError_t e1 = hipFuncSetAttribute((const void*)k_eval_multibody<0>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    hipError_t e2 = hipFuncSetAttribute((const void*)k_eval_multibody<1>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)k_eval_multibody<2>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)k_eval_multibody<3>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    if (e1 != hipSuccess || e2 != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(LDS) failed: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    attr_bytes.store(mb.lds.total_bytes + 1);  // + 1: the zero-initialised slots mean "not set"
  }
  if (sim_substeps > 0) hipLaunchKernelGGL(k_eval_multibody<2>, dim3(1, L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, LT, records, mb, 0);
  else if (trial && with_derivs) hipLaunchKernelGGL(k_eval_multibody<3>, dim3(L.N + 1 + (a.spec_knot ? 1 : 0), L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, L, records, mb, 0);  // records = the knot records (+ the speculative knot)
  else if (!trial) hipLaunchKernelGGL(k_eval_multibody<0>, dim3(L.N + 1, L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, L, records, mb, 0);
  else hipLaunchKernelGGL(k_eval_multibody<1>, dim3(L.N + 1, L.B, mb.ncand_loop > 0 ? 1 : ncand), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, LT, records, mb, cand0);
}
