// eval_multibody_host.h — host side of the whole-body stage kernel (eval_multibody.h): LDS carve-out, model checks, scratch sizes and
// the launcher's prototype.  The kernel itself is compiled in its own translation unit (eval_multibody.hip).
#pragma once
#include <stdexcept>
#include "solver_args.h"

#ifndef EVAL_THREADS
#define EVAL_THREADS 256  // threads per stage workgroup: 4 wavefronts, two workgroups per CU (measured against 512 = 8 wavefronts with a 128-VGPR cap, profiles/r03_*)
#endif
#define MB_SE3_SLOTS 6
#define MB_LDY 17  // leading dimension of Y16 = L^-1 [Jc^T | r1] (16 columns)
#define MB_STAGE_CONSTRAINT_ROWS 20  // LDS staging rows for the Jacobian of one constraint term (wrench cone: 17)

// ---- LDS carve-out ------------------------------------------------------------------------------------------
// Everything per body / per dof is stored as STRUCTURE OF ARRAYS (component e of body i at A[e * nj + i], of dof k at A[e * nv + k]):
// the phases are parallel over bodies / dofs / matrix entries, so the lanes of an access read consecutive doubles — no LDS bank
// conflicts — where the array-of-6-vectors layout (stride 12 dwords, 72 for the 6 x 6 blocks) gave 2- to 4-way conflicts on every load.
// The carve-out is sized so that TWO workgroups share a CU (<= 80 KB each for the complete Talos model, 154 KB before): the stage
// kernel is a chain of ~40 dependent phases, and the only thing that hides its latency is another knot on the same CU.  What made
// the difference:
//  * the nz right-hand sides of the implicit differentiation never exist in LDS: every wavefront builds its 16-column block of
//    R = [d r1 ; d r2] in REGISTERS in MFMA fragment layout and runs the whole chain of blocked triangular solves there
//    (mma_tile_rb); the rows of [A B] that d a determines leave for the record straight from the result registers (round 6; the kinodynamic stages,
//    whose d a is a closed form, still pass it through the per-workgroup HBM scratch `dsol`), d lambda (12 rows) goes to LDS;
//  * M = L L^T is kept tile-packed (lower block triangle, inverses of the diagonal blocks in place), no separate contact Jacobian,
//    composite inertias packed symmetric (21 of 36);
//  * the derivative blocks that depend on (q, v) only (B_i, their subtree sums, Bt, Tv, Bc Psd, Yc Psd) are formed BEFORE the
//    factorisation, so the body-level 6 x 6 blocks are dead when M is built and the two share one region; the forces at the
//    solution follow from Fc += Yc da + sum U_k acc_k - contact wrenches instead of a second pass over the body inertias.
// leading dimension of the staged Jacobian rows (see eval_multibody.h, P13)
static inline constexpr __host__ __device__ int mb_ldj(int nz) { return ((nz - 16 + 31) / 32) * 32 + 16; }

struct MbLds {
  int nj, nv, nq, nl_max;
  int nvp, nbm, ncb, ldl;  // padded nv, block count of the mass matrix, 16-column blocks of the right-hand sides, leading dim of d lambda
  // persistent: body arrays (SoA, ld nj), dof arrays (SoA, ld nv), vectors
  int oR, op, ov, oa, Hc, Fc;
  int J, U, Psd;
  int x, u, xn, a, lam, gam, bias, cfr, small, se3, red, early;
  // derivative blocks alive from the pre-pass to the right-hand sides (afterwards: gradient / Hessian-diagonal accumulators)
  int Phi, Bt, Tv, BcPsd, YcPsd, Psdd, Tq;
  // time-shared region: [Yc | stage 1: oY Bc oh of  ==  stage 2: Mt Y16 Sp LIs] ; stage 3 (terms): JS from Yc on
  int Yc, oY, Bc, oh, of, Mt, Y16, Sp, LIs, JS;
  int DL, V16;  // d lambda rows [12][ldl] (R2 in, d lambda out) ; V16 (the accelerations' back-substitution) aliases it
  int total;
  int stage_rows;  // rows of Jacobian staging (ld nz) that fit in the JS region
  int anc_bytes_off, total_bytes;
  int contact_dyn;  // the carve-out holds the blocks of the contact-constrained dynamics
  unsigned mg_nv, mg_nj, mg_nz, mg_n;  // magic_div (device_common.h) of the run-time divisors nv, nj, n + nu, n of the per-element loops
};

#define MB_RED_DOUBLES (112 + 24 * (EVAL_THREADS / 64))  // [0,16) merit partials | [16,48) sqrt(W) r of the stacked rows | [48,112) residual of a workgroup term | then 24 per wavefront

// contact_dyn = false: no stage of the problem has contact-constrained dynamics (kinodynamic / kinematic problems) — the factor of M, the
// right-hand-side blocks and the d lambda rows are never touched and get no LDS (BASELINE config 4: 88.8 -> 76.7 KB, two workgroups per CU)
static inline constexpr MbLds make_mb_lds(int nj, int nv, int nq, int nu, int nz, bool contact_dyn = true) {
  MbLds s{};
  s.nj = nj; s.nv = nv; s.nq = nq; s.nl_max = 12;
  s.nvp = (nv + 15) & ~15; s.nbm = s.nvp / 16;
  s.ncb = (nz + 15) / 16; s.ldl = 16 * s.ncb + 1;
  int o = 0;
  auto take = [&](int c) { int r = o; o += (c + 1) & ~1; return r; };
  s.oR = take(9 * nj); s.op = take(3 * nj); s.ov = take(6 * nj); s.oa = take(6 * nj); s.Hc = take(6 * nj); s.Fc = take(6 * nj);
  s.J = take(6 * nv); s.U = take(6 * nv); s.Psd = take(6 * nv);
  s.x = take(nq + nv); s.u = take(nu > 0 ? nu : 1); s.xn = take(nq + nv);
  s.a = take(s.nvp); s.lam = take(16); s.gam = take(16); s.bias = take(nv);
  s.cfr = take(2 * (12 + 36 + 6));  // per contact: R(9) p(3), Jlog6(c2Mc1) (36), world wrench (6)
  s.small = take(6 * 36 + 64);      // integrator 6x6 blocks, scratch, the term table of the stage
  s.se3 = take(MB_SE3_SLOTS * 48);  // per SE(3)-valued term: residual (6), Jacobian block (36 at +8)
  s.red = take(MB_RED_DOUBLES);
  s.early = take(2 * nz + 36 + 24 + 2 + 16);  // diagonal state / control costs accumulated beside the factorisation: gradient | diag(H) | base block | per-term cost | done flag | the term classification (bytes)
  s.Phi = take(6 * nv); s.Bt = take(6 * nv); s.Tv = take(6 * nv); s.BcPsd = take(6 * nv); s.YcPsd = take(6 * nv);
  s.Psdd = take(6 * nv); s.Tq = take(6 * nv);
  s.Yc = take(21 * nj);
  const int b3 = o;
  s.oY = take(36 * nj); s.Bc = take(36 * nj); s.oh = take(6 * nj); s.of = take(6 * nj);
  const int e1 = o;
  o = b3;
  const int ntile = s.nbm * (s.nbm + 1) / 2;
  if (contact_dyn) { s.Mt = take(ntile * 272); s.Y16 = take(s.nvp * MB_LDY); s.Sp = take(272); s.LIs = take(272); }
  else s.Mt = s.Y16 = s.Sp = s.LIs = b3;  // (never dereferenced)
  if (o < e1) o = e1;
  // terms: stacked cost rows (<= 32) / the rows of the constraint being emitted, ld nz, from Yc on (Yc and the stage-2 blocks are dead)
  s.JS = s.Yc;
  const int js_rows = 32 > MB_STAGE_CONSTRAINT_ROWS ? 32 : MB_STAGE_CONSTRAINT_ROWS;
  if (o < s.JS + js_rows * mb_ldj(nz) + 8) o = (s.JS + js_rows * mb_ldj(nz) + 8 + 1) & ~1;
  s.stage_rows = (o - s.JS) / mb_ldj(nz);
  s.DL = contact_dyn ? take(12 * s.ldl > s.nvp * 16 ? 12 * s.ldl : s.nvp * 16) : take(2);
  s.V16 = s.DL;
  s.total = o;
  s.anc_bytes_off = o * 8;
  s.total_bytes = o * 8 + 4 * nj * 8 + nv * 4 + nj * 4 * 3 + 64;
  s.contact_dyn = contact_dyn ? 1 : 0;
  s.mg_nv = magic_div(nv); s.mg_nj = magic_div(nj); s.mg_nz = magic_div(nz); s.mg_n = magic_div(2 * nv);
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

struct MbArgs {
  MbLds lds;
  double* scratch;        // per-workgroup HBM scratch
  size_t scratch_stride;  // doubles
  int ncand_loop;         // TRIAL == 1: > 0 = the workgroup walks this many candidates itself (grid z = 1)
  int sim_substeps;       // TRIAL == 2 (closed-loop simulation stand-in): integration steps ...
  double sim_dt;          // ... of this length
  const double* f_ext;    // TRIAL == 2: world-frame force at the base origin per instance [B][3] (mpc_simulate_push), or nullptr
  const double* sim_u;    // TRIAL == 2: joint torques per instance [B][nu] held during the call (mpc_simulate_torque: the start state is
                          // then a.x0, no feedback law), or nullptr
  double* sim_wrench;     // TRIAL == 2: contact wrenches of the last sub-step [B][2][6], or nullptr
};


// defined in eval_multibody.hip
void launch_eval_multibody(hipStream_t stream, const SolverArgs& a, const Layout& LT, double* records, double* scratch, size_t scratch_stride,
                           bool trial, int cand0 = 0, int ncand = 1, int sim_substeps = 0, double sim_dt = 0.0, bool with_derivs = false,
                           const double* f_ext = nullptr, bool contact_dyn = true, const double* sim_u = nullptr, double* sim_wrench = nullptr);
const void* eval_multibody_kernel(int trial);  // entry point of k_eval_multibody<trial> (occupancy tooling)
