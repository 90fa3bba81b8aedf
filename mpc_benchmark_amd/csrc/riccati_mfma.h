// riccati_mfma.h — proximal Riccati backward sweep with every matrix operand
// resident in LDS and the dense products / factorisations on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64, one 16x16 output tile per wavefront instruction).
//
// One workgroup (256 threads = 4 wavefronts, one per SIMD) per MPC instance walks the horizon backwards:
//   Ph = T^T P' T                         base-frame change of the co-state (6 x 6 block)
//   L L^T = I + mu_d Ph                   blocked Cholesky: 16x16 diagonal blocks factored AND inverted in the
//                                         registers of one wavefront (readlane broadcasts), panels and trailing
//                                         updates on MFMA
//   Pt = (I + mu_d Ph)^-1 Ph              blocked triangular solves with the pre-inverted diagonal blocks: pure
//                                         MFMA, one column block per wavefront, no workgroup barrier inside
//   G = Pt [A B] ; Hh = H + [A B]^T G     16-column panels: G panel in LDS, Hh panel to L2-resident scratch
//   stage KKT (controls, then ACTIVE constraint rows), gains K, k, Knu, knu — same blocked routines
//   P = Qh + Sh K + Ca^T Knu              MFMA, result stays in LDS as next knot's P'
// dims are padded to multiples of 16 inside LDS (np, mp); u-columns start at column np.
#pragma once
#include "solver_kernels.h"
#include "mfma_blocks.h"

#ifndef RIC_SMALL_THREADS
#define RIC_SMALL_THREADS 256  // workgroup of the small-problem instantiation (np = mp = 16): 20.5 us / knot (128: 26, 64: 38, 512: 21)
#endif
#ifndef RIC_THREADS
#define RIC_THREADS 512  // 8 wavefronts (2 per SIMD): the sweep is latency-bound, a second wave per SIMD hides LDS / MFMA latency
#endif
#if RIC_THREADS <= 256
#define RIC_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(1, 1)))  // one wavefront per SIMD: the whole 512-entry register file
#else
#define RIC_WAVES_ATTR
#endif
#define RIC_MAX_SERIES 8
#define RIC_SERIES_TOL 1e-14  // truncation of the series relative to Pt (the products themselves round at ~n eps)

// phase timing (shader clock) accumulated over the knots; read back with mpc_debug_get("ric_prof")
// The sweep is one long loop over the knots with ~15 phases per knot; left alone, the compiler hoists every per-lane
// index expression of every phase out of the knot loop and then spills those ~300 loop invariants to scratch, and
// each reload (s_waitcnt vmcnt(0)) also drains the global loads in flight.  Passing the lane ids through an empty
// asm at each phase boundary makes all index arithmetic phase-local: recomputed (a few integer ops), never spilled.
#define RIC_LAUNDER() do { asm volatile("" : "+v"(tid), "+v"(lane)); wv = __builtin_amdgcn_readfirstlane(tid >> 6); } while (0)
// nv, ks, Ke of the structured products, derived inside the phase that uses them (as scalars living across the whole knot loop they
// push the SGPR spills past what the spill VGPRs hold, and every reload of those drains the loads in flight)
#define RIC_SQ_DIMS() int n_l_ = n; asm volatile("" : "+s"(n_l_)); const int nv = SQ ? (n_l_ >> 1) : 0, ks = SQ ? (nv & ~3) : 0, Ke = SQ ? ((n_l_ + 3) & ~3) - ks : np; (void)nv; (void)ks; (void)Ke
#ifndef RIC_PROF_TID
#define RIC_PROF_TID 0  // the thread whose clock is read (developer builds: 64 = wavefront 1, ...)
#endif
#ifdef RIC_SUBPROF  // developer build (tools/phase_timers.py, PHASE_SUB=1): the long phases in pieces
#define RIC_SUB(slot) RIC_PROF(slot)
#else
#define RIC_SUB(slot) do { } while (0)
#endif
#define RIC_PROF(slot) do { RIC_LAUNDER(); if (tid == RIC_PROF_TID && a.prof && leg == 0) { const long long t1_ = clock64(); a.prof[(size_t)b * 64 + (slot)] += (double)(t1_ - t0_); t0_ = t1_; } } while (0)

#include "riccati_layout.h"  // struct RicLds, make_ric_lds: the LDS plan

// 6x6 inverse by Gauss-Jordan without pivoting, fully unrolled so that the 6x12 tableau stays in registers
// (-E6 = Jlog6 of the dynamics gap is a small perturbation of the identity: no pivoting needed).
DEV void inv6_unrolled(const double* A, double* Ainv) {
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

DEV double wave_sum_r(double v) { return wave_sum(v); }  // DPP reduction of solver_kernels.h

// ============================================================================================================
// one wavefront per SIMD (LDS-bound occupancy anyway): let the register allocator use the whole 512-entry file
// RT = threads per workgroup.  RT = RIC_THREADS (8 wavefronts) is the general kernel (np <= 96, nzp <= 128); RT = RIC_SMALL_THREADS is the
// small-problem variant (np = mp = 16, at most RT constraint rows: the centroidal OCP) — the same code with fewer wavefronts.
// SQ: structured dynamics rows (whole-body problems, layout.h oD12): only the v rows of [A B] are loaded and multiplied
// LEGS: parallel-in-time sweep (legs.h).  The grid is B x nlegs workgroups; workgroup (b, leg) walks the knots of its leg only.  A
// leg other than the last starts from a ZERO value function at its end (its end co-state is a parameter handled by
// k_leg_condense / k_leg_consensus) and additionally leaves, per knot, the (u,u) and (nu,u) blocks of the inverse stage KKT
// matrix (Mu, Znu: the solve with the right-hand side [I; 0] rides along in otherwise idle wavefronts).
// FN, FM > 0: the state / control dimensions of the problem as compile-time constants (n = FN, m <= FM; layout ric_fixed_layout): every
// stride, tile count and LDS offset folds into the instructions — the generic kernel spends about half of its issue slots on index
// arithmetic with run-time dimensions.  The host launches such an instantiation only when its handle's layout is that one (ric_same_layout).
static inline constexpr RicLds ric_fixed_layout(int n, int m, bool sq, int gfull = 1, int st_lds = 1) {
  RicLds s = make_ric_lds(n, m, 0, gfull, st_lds);
  if (sq) { s.sq = 1; s.nv = n / 2; }
  return s;
}
static inline bool ric_same_layout(const RicLds& x, const RicLds& y) {  // everything but the parts that depend on the number of constraint rows
  return x.np == y.np && x.mp == y.mp && x.nzp == y.nzp && x.ldl == y.ldl && x.ldr == y.ldr && x.lw == y.lw && x.gfull == y.gfull && x.st_lds == y.st_lds &&
         x.ovl == y.ovl && x.sq == y.sq && x.nv == y.nv && x.PT == y.PT && x.R1 == y.R1 && x.LP == y.LP && x.LI == y.LI && x.AB == y.AB && x.GP == y.GP &&
         x.vec == y.vec && x.Lr == y.Lr && x.LIr == y.LIr && x.W == y.W && x.ST == y.ST && x.CT == y.CT && x.VX == y.VX && x.Y == y.Y && x.SC == y.SC &&
         x.LIs == y.LIs && x.K2 == y.K2;
}
template <int RT, int NPMAX, bool SQ = false, bool LEGS = false, int FN = 0, int FM = 0, int FGF = 1, int FST = 1>
__global__ void __launch_bounds__(RT) k_riccati_mfma(SolverArgs a, RicLds Srt) {
  constexpr bool FX = FN > 0;
  constexpr RicLds SC_ = FX ? ric_fixed_layout(FN, FM, SQ, FGF, FST) : RicLds{};
  // S: the layout — the compile-time one for a fixed-dimension instantiation, with the two members that depend on the row count from the argument
  RicLds S_ = Srt;
  if constexpr (FX) { S_ = SC_; S_.iwork = Srt.iwork; S_.total_bytes = Srt.total_bytes; }
  const RicLds& S = S_;
  constexpr int NWV = RT / 64, NBMAX = NPMAX / 16, NZTMAX = NPMAX > 16 ? 8 : 2;
  constexpr int PT_ROWS = (NPMAX + NWV - 1) / NWV;                                // rows of Pt per wavefront
  constexpr int AB_ROWS = SQ ? (NPMAX / 2 + NWV - 1) / NWV : PT_ROWS;             // register prefetch capacity: rows of [A B] per wavefront (SQ: v rows only)
  constexpr int RIC_SERIES_TILES = (NBMAX * (NBMAX + 1) / 2 + NWV - 1) / NWV;     // lower-triangle output tiles per wavefront
  constexpr int RIC_G_TILES = (NBMAX * NZTMAX + NWV - 1) / NWV;                   // tiles of G per wavefront
  constexpr int RIC_U_TILES = NPMAX > 16 ? ((FX && FM > 32) ? 3 : 2) : 1;  // u tiles of Hh a wavefront keeps in registers (m > 32: 21 tiles on 8 wavefronts)
  constexpr bool CT_PREFETCH = NPMAX <= 80;  // the largest instantiation has no registers to spare for it
  constexpr int CT_ROWS = 16 / NWV, Y_ELEMS = (NPMAX > 16 ? 48 : 16) * 16 / RT + ((NPMAX > 16 ? 48 : 16) * 16 % RT ? 1 : 0);  // per-thread shares of CT (16 rows) and Y (mp x 16)  // lower-triangle tiles of the u rows of Hh a wavefront can keep in registers
  const Layout& L = a.L;
  constexpr int nthr = RT, nw = RT / 64;  // (the launch uses RT threads)
  // legs: the last leg (the only one with the terminal node, never shorter than the others) is dispatched first
  const int b = LEGS ? (int)(blockIdx.x % L.B) : (int)blockIdx.x, leg = LEGS ? a.nlegs - 1 - (int)(blockIdx.x / L.B) : 0;
  const bool par = LEGS && leg + 1 < a.nlegs;  // parametric leg: zero value function at its end
  int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // re-derived at every phase boundary (RIC_LAUNDER)
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = FX ? FN : L.n, nz = FX ? FN + FM : L.nz, N = L.N, nr = n + 1;
  const int k_top = par ? leg_start(a, leg + 1) - 1 : N - 1, k_bot = LEGS ? leg_start(a, leg) : 0;
  const int np = S.np, ldp = S.np + 1, mp = S.mp, nzp = S.nzp, ldl = S.ldl, ldr = S.ldr, nb = S.nb, nbm = S.nbm, nzt = nzp / 16, lw = S.lw, nwb = S.nwb;
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const bool ff = L.space == MPC_SPACE_MULTIBODY;
  // u rows of Hh straight from registers into the KKT operands (step 5 / 6) when every wavefront can hold its share
  const bool ureg = S.gfull && (nzt * (nzt + 1) / 2 - nb * (nb + 1) / 2) <= RIC_U_TILES * nw;
  // Ruu factorised by wavefront 0 while the others finish the x rows of Hh (needs the overlap layout of make_ric_lds)
  const bool ovl = S.ovl && ureg && nw >= 2 && nbm * (nbm + 1) / 2 <= nw && nbm * nb <= RIC_U_TILES * (nw - 1);
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *PT = sm + S.PT, *LP = sm + S.LP, *LI = sm + S.LI, *AB = sm + S.AB, *GP = sm + S.GP, *vec = sm + S.vec;
  double *Lr = sm + S.Lr, *LIr = sm + S.LIr, *W = sm + S.W, *CTl = sm + S.CT, *VXl = sm + S.VX, *Yl = sm + S.Y, *SCl = sm + S.SC, *LIs = sm + S.LIs;
  int* act_idx = (int*)(sm + S.iwork);
  int* iflag = act_idx + L.c;  // [0] factorisation flag, [1] ca, [2..5] per-wave active counts
  double *ph = vec, *ft = vec + nzp, *vv = vec + 2 * nzp, *w = vec + 3 * nzp, *gh = vec + 4 * nzp, *pvec = vec + 5 * nzp, *dtl = vec + 6 * nzp;
  double* kvc = dtl + L.c;
  double* e6l = kvc + L.c;  // 36 doubles; vec reserves 8 (nzp + c) + 64
  double* wred = e6l + 40;  // per-wavefront partial sums (<= 16)
  double* t6l = e6l + 56;   // T6 = (-E6)^-1 (36)
  double* gpre = e6l + 92;  // gradient of the knot (nz); vec holds 7 nzp + 2 c + 92 <= 8 (nzp + c) + 64 doubles
  double* d12l = gpre + nzp;  // D1_b (36) | Dd_b (36) | dt | valid (layout.h, oD12)
  double* wk = a.work + ((size_t)b + (size_t)leg * L.B) * L.work_stride;  // legs: the scratch of (instance, leg) — the legs of an instance run side by side
  double* Hh = wk + L.wHh;  // nz x nz, leading dimension nz (L2-resident scratch)
  double* ST = S.st_lds ? sm + S.ST : wk + L.wG;  // Sh^T (mp x np)
  double* GU = (S.gfull == 2) ? wk + L.wPt : GP;   // G_u (np x mp)

  // ---- terminal node: P_N = H + Ca^T Ca / mu ; p_N = grad + Ca^T dt / mu ----
  if (par) {
    // the leg's terminal cost: 1/2 x^T Pg x with Pg = the exact Hessian the consensus found at this cut in the previous pass / tick
    // (legs.h: the consensus then solves for the difference only), or zero
    const double* pg = leg_ptr(a, b, leg) + L.lcP;
    for (int idx = tid; idx < np * ldp; idx += nthr) { const int i = idx / ldp, c0 = idx % ldp; PT[idx] = (a.leg_guess && i < n && c0 < n) ? pg[i * n + c0] : 0.0; }
    for (int r = tid; r < nzp; r += nthr) pvec[r] = 0.0;
    __syncthreads();
  } else {
    const double* kn = knot_ptr(a, b, N);
    double* g = gain_ptr(a, b, N);
    const int c = (int)kn[L.oMISC + MISC_NC];
    for (int i = tid; i < c; i += nthr) g[L.oknu + i] = kn[L.oDT + i] / mu;
    for (int idx = tid; idx < c * n; idx += nthr) {
      const int i = idx / n, z = idx % n;
      g[L.oKnu + idx] = (kn[L.oACT + i] != 0.0) ? kn[L.oCD + i * nz + z] / mu : 0.0;
    }
    for (int idx = tid; idx < np * ldp; idx += nthr) PT[idx] = 0.0;
    __syncthreads();
    for (int r = wv; r < n; r += nw)
      for (int s = lane; s < n; s += 64) {
        double t = kn[L.oH + r * nz + s];
        for (int i = 0; i < c; ++i) if (kn[L.oACT + i] != 0.0) t += kn[L.oCD + i * nz + r] * g[L.oKnu + i * n + s];
        g[L.oP + r * n + s] = t;
        PT[r * ldp + s] = t;
      }
    for (int r = tid; r < n; r += nthr) {
      double t = kn[L.oG + r];
      for (int i = 0; i < c; ++i) t += kn[L.oCD + i * nz + r] * g[L.oknu + i];
      g[L.op + r] = t;
      pvec[r] = t;
    }
    __syncthreads();
  }

  // The small per-knot vectors (row counts, active flags, E6, gap, multiplier estimate, gradient) of knot k - 1 are
  // requested from HBM in the middle of knot k and sit in registers until the next iteration starts: one value per
  // thread each, so no phase of a knot begins by waiting on a scattered HBM load.
  double pre_m, pre_c, pre_act, pre_e6, pre_f, pre_le, pre_g;
  auto prefetch_small = [&](int kk) {
    const double* kp = knot_ptr(a, b, kk);
    const int row = wv * 64 + lane;
    pre_m = kp[L.oMISC + MISC_M]; pre_c = kp[L.oMISC + MISC_NC];
    pre_act = kp[L.oACT + (row < L.c ? row : 0)];
#ifdef RIC_T6_INVERT
    pre_e6 = kp[L.oE6 + (tid < 36 ? tid : 0)];
#else
    pre_e6 = kp[L.oT6k + (tid < 36 ? tid : 0)];  // T6 = (-E6)^-1 itself, left beside E6 by the stage kernel in closed form (round 6)
#endif
    pre_f = kp[L.oF + (tid < n ? tid : 0)];
    pre_le = a.lams_e[((size_t)b * (N + 1) + kk + 1) * n + (tid < n ? tid : 0)];
    pre_g = kp[L.oG + (tid < nz ? tid : 0)];
  };
  prefetch_small(k_top);

  long long t0_ = clock64();
  for (int k = k_top; k >= k_bot; --k) {
    const double* kn = knot_ptr(a, b, k);
    double* g = gain_ptr(a, b, k);
    const int m = (int)pre_m, c = (int)pre_c;
    const double dreg = SQ ? kn[L.oD12 + (tid < 74 ? tid : 0)] : 0.0;  // in flight during step 1
    // ---- 1. active rows (ballot prefix; wave q owns rows 64q .. 64q+63, c <= 256), T6 = (-E6)^-1 ----
    const int arow = wv * 64 + lane;
    const bool is_act = (arow < c) && (pre_act != 0.0);
    const unsigned long long amask = __ballot(is_act);
    if (lane == 0) iflag[2 + wv] = __popcll(amask);
#ifdef RIC_T6_INVERT
    if (tid < 36) e6l[tid] = -pre_e6;
#else
    if (tid < 36) t6l[tid] = ff ? pre_e6 : ((tid % 7 == 0) ? 1.0 : 0.0);
#endif
    if (tid < n) ft[tid] = pre_f + mud * pre_le;
    if (tid < nz) gpre[tid] = pre_g;
    __syncthreads();
    RIC_SUB(23);
    {
      int off = 0;
      for (int q = 0; q < wv; ++q) off += iflag[2 + q];
      if (is_act) act_idx[off + __popcll(amask & ((1ull << lane) - 1ull))] = arow;
      if (tid == 0) {
        int ca_ = 0;
        for (int q = 0; q < nw; ++q) ca_ += iflag[2 + q];
        iflag[1] = ca_;
#ifdef RIC_T6_INVERT
        if (ff) inv6_unrolled(e6l, t6l);
        else for (int i2 = 0; i2 < 36; ++i2) t6l[i2] = (i2 % 7 == 0) ? 1.0 : 0.0;
#endif
      }
    }
    __syncthreads();
    RIC_PROF(0);
    const int ca = iflag[1];
    // rows of the active constraints ([Ca_x | dt], Da^T) for the stage KKT system of step 6: requested here, in registers
    // until then (an L2 round trip that nothing waits for)
    double pf_ct[CT_ROWS][2], pf_y[Y_ELEMS];
    if (CT_PREFETCH && ca <= 16) {
#pragma unroll
      for (int ii = 0; ii < CT_ROWS; ++ii) {
        const int i = wv + ii * nw;
        const int ai = (i < ca) ? act_idx[i] : 0;
#pragma unroll
        for (int zz = 0; zz < 2; ++zz) {
          const int z = lane + 64 * zz;
          const bool isx = z < n, isd = z == np;
          pf_ct[ii][zz] = kn[isd ? L.oDT + ai : L.oCD + ai * nz + (isx ? z : 0)];
        }
      }
#pragma unroll
      for (int e = 0; e < Y_ELEMS; ++e) {
        const int idx = tid + e * nthr, i = idx >> 4, j = idx & 15;
        const int aj = (j < ca) ? act_idx[j] : 0;
        pf_y[e] = kn[L.oCD + aj * nz + n + ((idx < mp * 16 && i < m) ? i : 0)];
      }
    }
    const double* T6 = t6l;  // LDS copy; the gain record gets it below
    if (tid < 36) g[L.oT6 + tid] = t6l[tid];
    if (SQ && tid < 74) d12l[tid] = dreg;
#ifndef RIC_T6_FOUR_BARRIERS
    if (ff) {
      // Ph = T^T P' T with T = diag(T6, I): only the first six rows / columns change, and P' is symmetric (step 7 symmetrises it exactly).  B = P'[:, 0:6] T6 (n x 6)
      // from the old PT ; then columns 0..5 <- B on the rows >= 6 and, mirrored, rows 0..5 <- B^T ; the 6 x 6 corner <- T6^T B[0:6].  Two barriers (round 3 - 5: the
      // column pass and the row pass one after the other through a scratch, four barriers).
      double* tmp = LP;  // scratch (LP is dead here)
      for (int idx = tid; idx < n * 6; idx += nthr) {
        const int i = idx / 6, j = idx % 6;
        double s = 0;
        for (int l = 0; l < 6; ++l) s += PT[i * ldp + l] * T6[l * 6 + j];
        tmp[idx] = s;
      }
      if (tid >= nthr - 6) { const int i = tid - (nthr - 6); double s = 0; for (int l = 0; l < 6; ++l) s += T6[l * 6 + i] * pvec[l]; ph[i] = s; }
      for (int i = 6 + tid; i < n; i += nthr) ph[i] = pvec[i];
      __syncthreads();
      for (int idx = tid; idx < n * 6; idx += nthr) {
        const int i = idx / 6, j = idx % 6;
        if (i >= 6) { const double v = tmp[idx]; PT[i * ldp + j] = v; PT[j * ldp + i] = v; }
        else { double s = 0; for (int l = 0; l < 6; ++l) s += T6[l * 6 + i] * tmp[l * 6 + j]; PT[i * ldp + j] = s; }
      }
    } else {
      for (int i = tid; i < n; i += nthr) ph[i] = pvec[i];
    }
    __syncthreads();
#else
    if (ff) {
      double* tmp = LP;  // scratch (LP is dead here)
      for (int idx = tid; idx < n * 6; idx += nthr) {
        const int i = idx / 6, j = idx % 6;
        double s = 0;
        for (int l = 0; l < 6; ++l) s += PT[i * ldp + l] * T6[l * 6 + j];
        tmp[idx] = s;
      }
      __syncthreads();
      for (int idx = tid; idx < n * 6; idx += nthr) PT[(idx / 6) * ldp + idx % 6] = tmp[idx];
      __syncthreads();
      for (int i = wv; i < 6; i += nw)
        for (int j = lane; j < n; j += 64) {
          double s = 0;
          for (int l = 0; l < 6; ++l) s += T6[l * 6 + i] * PT[l * ldp + j];
          tmp[i * n + j] = s;
        }
      if (tid < 6) { double s = 0; for (int l = 0; l < 6; ++l) s += T6[l * 6 + tid] * pvec[l]; ph[tid] = s; }
      __syncthreads();
      for (int i = wv; i < 6; i += nw) for (int j = lane; j < n; j += 64) PT[i * ldp + j] = tmp[i * n + j];
      for (int i = 6 + tid; i < n; i += nthr) ph[i] = pvec[i];
    } else {
      for (int i = tid; i < n; i += nthr) ph[i] = pvec[i];
    }
    __syncthreads();
#endif
    // Structured knot: [A B]_q = D1 [I 0 0] + Dd [A B]_v (semi-implicit Euler), so Pt [A B] and [A B]^T G need the v rows of [A B]
    // only: K = nv instead of n in both products of step 5 (ks .. ks + Ke: the v rows, aligned down to the MFMA depth of 4)
    constexpr bool sq = SQ;  // every stage knot of a whole-body problem has dynamics rows (valid flag d12l[73] = 1)
    RIC_PROF(1);
    // ---- 2. mx = ft - mu_d ph (the vector of the forward sweep's record): w of step 3c is Pt mx + ph.  [With M = (I + mu_d Ph)^-1,
    // Pt = M Ph:  w = vv - mu_d Pt vv = M vv  for  vv = Ph ft + ph,  and  M ph = ph - mu_d Pt ph.]  ||Ph||_F^2, which decides the length of
    // the series, is the trace of its first product Ph Ph (step 3a): no pass over Ph here ----
    if (tid < np) vv[tid] = (tid < n) ? ft[tid] - mud * ph[tid] : 0.0;
    RIC_PROF(2);
    // [A B] of this knot: issue the HBM loads here (the series below hides their latency), park them in
    // registers, drop them into LDS in step 4
    double abr[AB_ROWS][2];
    {
      // column of [A B] behind padded column zp (x columns 0..n-1, u columns np..np+m-1), clamped to a valid one so
      // that every load is unconditional (straight-line code: the loads of all rows are in flight together)
      RIC_SQ_DIMS();
      const int z0 = lane, z1 = lane + 64;
      const bool ok0 = z0 < n || (z0 >= np && z0 - np < m), ok1 = z1 < n || (z1 >= np && z1 - np < m);
      const int c0 = ok0 ? (z0 < n ? z0 : n + z0 - np) : 0, c1 = ok1 ? (z1 < n ? z1 : n + z1 - np) : 0;
      const double* ab0 = kn + L.oAB;
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = (SQ ? nv : 0) + wv + nw * q;  // wave-uniform row
        const double* src = ab0 + (size_t)(i < n ? i : 0) * nz;
        // mask by multiplication: with a select the compiler makes the load itself conditional (branch + s_waitcnt
        // vmcnt(0) per row: the 12 rows of a wavefront were fetched one after the other, 4 us per knot)
        const double v0 = src[c0], v1 = src[c1];
        abr[q][0] = v0 * ((ok0 && i < n) ? 1.0 : 0.0);
        abr[q][1] = v1 * ((ok1 && i < n) ? 1.0 : 0.0);
      }
    }
    RIC_PROF(18);
    if (k > k_bot) prefetch_small(k - 1);  // consumed at the top of the next iteration
    RIC_PROF(13);
    int nser = 0;
    {
      // ---- 3a. series sum_{i <= nser} (-X)^i Ph, X = mu_d Ph, as a product of factors instead of nser Horner steps:
      //   nser <= 1:  Pt = Ph - mu_d Q,  Q = Ph^2                                  1 product
      //   nser <= 3:  T = Ph - mu_d Q ;  Pt = T + mu_d^2 T Q                       2 products  (I - X)(I + X^2)
      //   nser <= 7:  ... ;              Pt = Pt + mu_d^4 Pt Q^2                   4 products  (I - X)(I + X^2)(I + X^4)
      // Every factor is a polynomial in the symmetric Ph: only the lower block triangle of 16x16 tiles is computed
      // (dealt round-robin to the wavefronts) and mirrored on the way out.  Q, then Q^2, live in LP; Ph is never copied.
      const int ntile = nb * (nb + 1) / 2;
      d4_t res[RIC_SERIES_TILES];
      int tri[RIC_SERIES_TILES], tcj[RIC_SERIES_TILES];
#pragma unroll
      for (int sidx = 0; sidx < RIC_SERIES_TILES; ++sidx) {
        const int t = wv + sidx * nw;
        int ri = 0, rem = t;
        while (rem > ri) { rem -= ri + 1; ++ri; }  // t = ri (ri + 1) / 2 + cj, cj <= ri
        tri[sidx] = ri; tcj[sidx] = rem;
      }
      auto products = [&](const double* Am, const double* Bm) {  // both with leading dimension np + 1
#pragma unroll
        for (int sidx = 0; sidx < RIC_SERIES_TILES; ++sidx) {
          res[sidx] = d4_t{0, 0, 0, 0};
          if (wv + sidx * nw < ntile) mma_tile<false>(res[sidx], Am + (tri[sidx] * 16) * ldp, ldp, 1, Bm + tcj[sidx] * 16, ldp, 1, np, lane);
        }
      };
      // PT <- PT + sc * res (pt) ; LP <- res (lp): own tile and its mirror image
      auto update = [&](bool pt, double sc, bool lp) {
#pragma unroll
        for (int sidx = 0; sidx < RIC_SERIES_TILES; ++sidx) {
          if (wv + sidx * nw < ntile) {
            const int r0 = tri[sidx] * 16 + (lane >> 4), c0 = tcj[sidx] * 16 + (lane & 15);
            const bool offd = tri[sidx] != tcj[sidx];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int r = r0 + 4 * q;
              const double rv = res[sidx][q];
              if (lp) { LP[r * ldp + c0] = rv; if (offd) LP[c0 * ldp + r] = rv; }
              if (pt) { const double v = PT[r * ldp + c0] + sc * rv; PT[r * ldp + c0] = v; if (offd) PT[c0 * ldp + r] = v; }
            }
          }
        }
      };
      products(PT, PT);  // Q = Ph Ph
      {  // ||Ph||_F^2 = trace(Q) (Ph symmetric; its padding is zero): the diagonal elements of the diagonal tiles
        double tr = 0.0;
#pragma unroll
        for (int sidx = 0; sidx < RIC_SERIES_TILES; ++sidx)
          if (wv + sidx * nw < ntile && tri[sidx] == tcj[sidx]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) tr += ((lane >> 4) + 4 * q == (lane & 15)) ? res[sidx][q] : 0.0;
          }
        tr = wave_sum_r(tr);
        if (lane == 0) wred[wv] = tr;
      }
      RIC_PROF(14);
      __syncthreads();   // PT is overwritten below: every wavefront must be done reading it
      RIC_PROF(15);
      // Pt = (I + X)^-1 Ph with X = mu_d Ph, ||X||_2 <= rho = mu_d ||Ph||_F.  mu_d = dyn_al_scale * mu is tiny, so the
      // Neumann series Ph (I - X + X^2 - ...) reaches double precision after a few terms (remainder <= rho^(nser+1)):
      // products on the matrix cores replace the n x n factorisation and the two triangular solves.  Larger rho falls back
      // to Horner steps (nser > 7) or to the Cholesky path (nser < 0); PT is still Ph at this point.
      {
        double fro = 0;
        for (int q = 0; q < nw; ++q) fro += wred[q];
        const double rho = mud * sqrt(fmax(fro, 0.0));
        double rem = rho;
        while (rem > RIC_SERIES_TOL && nser < RIC_MAX_SERIES) { rem *= rho; ++nser; }
        if (rem > RIC_SERIES_TOL) nser = -1;
        if (tid == RIC_PROF_TID && a.prof && leg == 0) {
          double* pr = a.prof + (size_t)b * 64;
          pr[20] = fmax(pr[20], rho); pr[21] += (nser >= 0) ? nser : 0; pr[22] += (nser < 0) ? 1.0 : 0.0;
        }
      }
      RIC_PROF(17);
      if (nser >= 1 && nser <= 7) {
        update(true, -mud, nser >= 2);   // PT <- Ph - mu_d Q ; LP <- Q
        __syncthreads();
        RIC_PROF(16);
      }
      if (nser >= 2 && nser <= 7) {
        products(PT, LP);  // T Q
        RIC_PROF(14);
        __syncthreads();
        RIC_PROF(15);
        update(true, mud * mud, false);
        __syncthreads();
        RIC_PROF(16);
      }
      if (nser >= 4 && nser <= 7) {
        products(LP, LP);  // Q^2
        RIC_PROF(14);
        __syncthreads();
        RIC_PROF(15);
        update(false, 0.0, true);
        __syncthreads();
        RIC_PROF(16);
        products(PT, LP);
        RIC_PROF(14);
        __syncthreads();
        RIC_PROF(15);
        update(true, mud * mud * mud * mud, false);
        __syncthreads();
        RIC_PROF(16);
      }
      RIC_PROF(3);
    }
    if (nser > 7) {
      // ---- 3a'. longer series: Horner steps T <- Ph - mu_d T Ph on the lower block triangle;
      // products first, barrier, then the in-place update of PT
      const int ntile = nb * (nb + 1) / 2;
      for (int i = wv; i < np; i += nw) for (int j = lane; j < np; j += 64) LP[i * ldl + j] = PT[i * ldp + j];
      __syncthreads();
      for (int it = 0; it < nser; ++it) {
        d4_t res[RIC_SERIES_TILES];
        int tri[RIC_SERIES_TILES], tcj[RIC_SERIES_TILES];
#pragma unroll
        for (int sidx = 0; sidx < RIC_SERIES_TILES; ++sidx) {
          const int t = wv + sidx * nw;
          res[sidx] = d4_t{0, 0, 0, 0};
          int ri = 0, rem = t;
          while (rem > ri) { rem -= ri + 1; ++ri; }  // t = ri (ri + 1) / 2 + cj, cj <= ri
          tri[sidx] = ri; tcj[sidx] = rem;
          if (t < ntile) mma_tile<false>(res[sidx], PT + (ri * 16) * ldp, ldp, 1, LP + rem * 16, ldl, 1, np, lane);
        }
        RIC_PROF(14);
        __syncthreads();
        RIC_PROF(15);
#pragma unroll
        for (int sidx = 0; sidx < RIC_SERIES_TILES; ++sidx) {
          const int t = wv + sidx * nw;
          if (t < ntile) {
            const int r0 = tri[sidx] * 16 + (lane >> 4), c0 = tcj[sidx] * 16 + (lane & 15);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const double v = LP[(r0 + 4 * q) * ldl + c0] - mud * res[sidx][q];
              PT[(r0 + 4 * q) * ldp + c0] = v;
              if (tri[sidx] != tcj[sidx]) PT[c0 * ldp + r0 + 4 * q] = v;
            }
          }
        }
        __syncthreads();
        RIC_PROF(16);
      }
      RIC_PROF(3);
    } else if (nser < 0) {
      // ---- 3b. LP <- I + mud Ph = L L^T ; PT <- (L L^T)^-1 PT ----
      for (int i = wv; i < np; i += nw)
        for (int j = lane; j < np; j += 64) LP[i * ldl + j] = mud * PT[i * ldp + j] + (i == j ? 1.0 : 0.0);
      __syncthreads();
      if (!chol_blocked(LP, ldl, nb, LI, tid, iflag)) { if (tid == 0) a.inst[b].done = 2; return; }
      RIC_PROF(3);
      trsm_fwd_blocked(LP, ldl, LI, nb, PT, ldp, nb, wv, nw, lane);
      trsm_bwd_blocked(LP, ldl, LI, nb, PT, ldp, nb, wv, nw, lane);
      __syncthreads();
    }
    RIC_PROF(5);
    // w = Pt mx + ph (step 2; vv holds mx), store Pt for the forward sweep (Pt is symmetric up to rounding).  The wavefront that owns
    // row i of Pt also holds row i of [A B] in registers: its share of gh = grad + [A B]^T w accumulates on the fly.
    double gp0 = 0.0, gp1 = 0.0;
#pragma unroll
    for (int q = 0; q < PT_ROWS; ++q) {
      const int i = wv + nw * q;
      if (i < n) {
        double s = 0;
        for (int j = lane; j < n; j += 64) { const double pv = PT[i * ldp + j]; s += pv * vv[j]; g[L.oMx + i * n + j] = pv; }
        s = wave_sum_r(s);
        const double wi = ph[i] + s;
        if (lane == 0) w[i] = wi;
        if constexpr (!SQ) { gp0 += abr[q][0] * wi; gp1 += abr[q][1] * wi; }
      }
    }
    for (int i = tid; i < n; i += nthr) g[L.omx + i] = vv[i];
    RIC_SUB(24);
    if (sq) {
      // v rows of PT, each by the wavefront that has just stored it (the q rows stay as they are): row nv + jq <- row jq of
      // Pe^T = Dd^T Pt[q rows] + Pt[v rows], then its v columns <- (. Dd + .) so that the v rows of the product of step 5 are
      // Ge = Dd^T G_q + G_v, what Hh needs, directly
      RIC_SQ_DIMS();
      const double dts = d12l[72];
      const int i0 = nv + ((wv - nv) & (nw - 1));
      // all reads of a pass first, then its writes: the rows of a wavefront are independent, but through the one pointer the compiler
      // would keep them in order (a dependent LDS round trip per row and column half)
      double t0[AB_ROWS], t1[AB_ROWS];
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = i0 + nw * q, jq = i - nv;
        t0[q] = t1[q] = 0.0;
        if (i < n) {
          const int j0 = lane, j1 = (lane + 64 < n) ? lane + 64 : lane;
          t0[q] = PT[i * ldp + j0]; t1[q] = PT[i * ldp + j1];
          if (jq >= 6) { t0[q] += dts * PT[jq * ldp + j0]; t1[q] += dts * PT[jq * ldp + j1]; }
          else for (int l = 0; l < 6; ++l) { const double dl = d12l[36 + l * 6 + jq]; t0[q] += dl * PT[l * ldp + j0]; t1[q] += dl * PT[l * ldp + j1]; }
        }
      }
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = i0 + nw * q;
        if (i < n) { PT[i * ldp + lane] = t0[q]; if (lane + 64 < n) PT[i * ldp + lane + 64] = t1[q]; }
      }
      __builtin_amdgcn_wave_barrier();
      const int lq = (lane < nv) ? lane : 0;
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = i0 + nw * q;
        t0[q] = 0.0;
        if (i < n) {
          t0[q] = PT[i * ldp + nv + lq];
          if (lq >= 6) t0[q] += dts * PT[i * ldp + lq];
          else for (int l = 0; l < 6; ++l) t0[q] += d12l[36 + l * 6 + lq] * PT[i * ldp + l];
        }
      }
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = i0 + nw * q;
        if (i < n && lane < nv) PT[i * ldp + nv + lane] = t0[q];
      }
    }
    RIC_PROF(6);
    // ---- 4. AB into LDS (zero padded; u-columns start at np) ; gh from the per-wavefront partial sums (GP is free) ----
    if constexpr (SQ) {
      RIC_SQ_DIMS();
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = nv + wv + nw * q;
        if (i < np) { if (lane < nzp) AB[i * nzp + lane] = abr[q][0]; if (lane + 64 < nzp) AB[i * nzp + lane + 64] = abr[q][1]; }
      }
      for (int idx = tid; idx < (nv - ks) * nzp; idx += nthr) AB[ks * nzp + idx] = 0.0;  // q rows inside the first MFMA group
      __syncthreads();
      // gh = grad + [A B]^T w = grad + [I 0 0]^T D1^T w_q + [A B]_v^T (Dd^T w_q + w_v), by the last two wavefronts (they have the fewest tiles below)
      const int zp = tid - (nthr - 128);
      if (zp >= 0 && zp < nzp) {
        const int z = (zp < n) ? zp : ((zp >= np && zp - np < m) ? n + zp - np : -1);
        if (z >= 0) {
          const double dts = d12l[72];
          double sacc = gpre[z];
          if (zp < 6) { for (int l = 0; l < 6; ++l) sacc += d12l[l * 6 + zp] * w[l]; }
          else if (zp < nv) sacc += w[zp];
          for (int kq = 0; kq < 6; ++kq) {
            double we = w[nv + kq];
            for (int l = 0; l < 6; ++l) we += d12l[36 + l * 6 + kq] * w[l];
            sacc += AB[(nv + kq) * nzp + zp] * we;
          }
          for (int kq = 6; kq < nv; ++kq) sacc += AB[(nv + kq) * nzp + zp] * (w[nv + kq] + dts * w[kq]);
          gh[z] = sacc;
        }
      }
    } else {
#pragma unroll
    for (int q = 0; q < AB_ROWS; ++q) {
      const int i = wv + nw * q;
      if (i < np) { if (lane < nzp) AB[i * nzp + lane] = abr[q][0]; if (lane + 64 < nzp) AB[i * nzp + lane + 64] = abr[q][1]; }
    }
    if (lane < nzp) GP[wv * nzp + lane] = gp0;
    if (lane + 64 < nzp) GP[wv * nzp + lane + 64] = gp1;
    __syncthreads();
    for (int zp = tid; zp < nzp; zp += nthr) {
      const int z = (zp < n) ? zp : ((zp >= np && zp - np < m) ? n + zp - np : -1);
      if (z >= 0) { double s = gpre[z]; for (int q = 0; q < nw; ++q) s += GP[q * nzp + zp]; gh[z] = s; }
    }
    }
    RIC_PROF(7);
    if (S.gfull) {
    // ---- 5. G = Pt [A B] : every 16x16 tile dealt round-robin to the wavefronts, kept in registers until all of
    // them are done (barrier), then G_x overwrites PT (Pt itself is no longer needed: it went to the gain record
    // above) and G_u goes to GP ; Hh = H + [A B]^T G on the lower block triangle only (Hh is symmetric) ----
    {
      RIC_SQ_DIMS();
      const int ngt = nb * nzt;
      // overlap path: the H values of this wavefront's Ruu tile and of its first Sh^T tile are requested now, behind the G products
      // (both tiles are otherwise the first thing a wavefront does after a barrier: their HBM latency was in the open)
      double hp_uu[4] = {0, 0, 0, 0}, hp_ux[4] = {0, 0, 0, 0};
      auto h_request = [&](int zi, int cj, double (&h)[4]) {
        const int col_p = cj * 16 + (lane & 15);
        const int zc = (col_p < n) ? col_p : ((col_p >= np && col_p - np < m) ? n + col_p - np : -1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int rp = zi * 16 + (lane >> 4) + 4 * q;
          const int zr = (rp < n) ? rp : ((rp >= np && rp - np < m) ? n + rp - np : -1);
          // clamped address, mask by multiplication: a select makes the load itself conditional (a branch and a wait per element)
          h[q] = kn[L.oH + (zr >= 0 ? zr : 0) * nz + (zc >= 0 ? zc : 0)] * ((zr >= 0 && zc >= 0) ? 1.0 : 0.0);
        }
      };
      if (ovl) {
        if (wv < nbm * (nbm + 1) / 2) {
          int ubi = 0, ubj = wv;
          while (ubj > ubi) { ubj -= ubi + 1; ++ubi; }
          h_request(nb + ubi, nb + ubj, hp_uu);
        }
        if (wv >= 1 && wv - 1 < nbm * nb) h_request(nb + (wv - 1) / nb, (wv - 1) % nb, hp_ux);
      }
      d4_t gres[RIC_G_TILES];
#pragma unroll
      for (int sidx = 0; sidx < RIC_G_TILES; ++sidx) {
        const int t = wv + sidx * nw;
        gres[sidx] = d4_t{0, 0, 0, 0};
        if (t < ngt) mma_tile<false>(gres[sidx], PT + ks * ldp + (t / nzt) * 16, 1, ldp, AB + ks * nzp + (t % nzt) * 16, nzp, 1, Ke, lane);  // Pt symmetric
      }
      RIC_SUB(26);
      __syncthreads();
#pragma unroll
      for (int sidx = 0; sidx < RIC_G_TILES; ++sidx) {
        const int t = wv + sidx * nw;
        if (t < ngt) {
          const int ri = t / nzt, cj = t % nzt;
          if (sq && cj * 16 < nv) {  // + Pt_q D1 on the q columns, read at the tile's own place in PT before G_x overwrites it (rows nv.. hold
            const int col = cj * 16 + (lane & 15);  // Pe^T there: the v rows of the product are Ge = Dd^T G_q + G_v, what Hh needs)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int row = ri * 16 + (lane >> 4) + 4 * q;
              double ad = 0.0;
              if (col >= 6) ad = (col < nv) ? PT[row * ldp + col] : 0.0;
              else for (int l = 0; l < 6; ++l) ad += PT[row * ldp + l] * d12l[l * 6 + col];
              gres[sidx][q] += ad;
            }
          }
          if (cj < nb) tile_store(PT + (ri * 16) * ldp + cj * 16, ldp, gres[sidx], lane);
          else tile_store(GU + (ri * 16) * mp + (cj - nb) * 16, mp, gres[sidx], lane);
        }
      }
      __syncthreads();
      RIC_SUB(27);
      const int nht = nzt * (nzt + 1) / 2, nxt = nb * (nb + 1) / 2;
      // one lower-triangle tile (zi, cj) of Hh = H + [A B]^T G
      auto hh_tile_at = [&](int zi, int cj, d4_t& out, int (&zr)[4], int& zc, const double* hp) {  // hp: the tile's H values, requested earlier (or null)
        const int col_p = cj * 16 + (lane & 15);
        zc = (col_p < n) ? col_p : ((col_p >= np && col_p - np < m) ? n + col_p - np : -1);
        double h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // H loads are in flight while the matrix cores work
          const int rp = zi * 16 + (lane >> 4) + 4 * q;
          zr[q] = (rp < n) ? rp : ((rp >= np && rp - np < m) ? n + rp - np : -1);
          h[q] = hp ? hp[q] : kn[L.oH + (zr[q] >= 0 ? zr[q] : 0) * nz + (zc >= 0 ? zc : 0)] * ((zr[q] >= 0 && zc >= 0) ? 1.0 : 0.0);  // (see h_request)
        }
        d4_t acc = d4_t{0, 0, 0, 0};
        if (cj < nb) mma_tile<false>(acc, AB + ks * nzp + zi * 16, 1, nzp, PT + ks * ldp + cj * 16, ldp, 1, Ke, lane);
        else mma_tile<false>(acc, AB + ks * nzp + zi * 16, 1, nzp, GU + ks * mp + (cj - nb) * 16, mp, 1, Ke, lane);
        if (sq && zi * 16 < nv) {  // + [I 0 0]^T D1^T G_q on the q rows (x tiles only: cj <= zi < nb)
          const int col = cj * 16 + (lane & 15);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int rp = zi * 16 + (lane >> 4) + 4 * q;
            double ad = 0.0;
            if (rp >= 6) ad = (rp < nv) ? PT[rp * ldp + col] : 0.0;
            else for (int l = 0; l < 6; ++l) ad += d12l[l * 6 + rp] * PT[l * ldp + col];
            acc[q] += ad;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = h[q] + acc[q];
      };
      auto hh_tile = [&](int t, d4_t& out, int (&zr)[4], int& zc) {
        int zi = 0, cj = t;
        while (cj > zi) { cj -= zi + 1; ++zi; }  // t = zi (zi + 1) / 2 + cj, cj <= zi
        hh_tile_at(zi, cj, out, zr, zc, nullptr);
      };
      auto x_tile = [&](int t, const double* hp = nullptr) {  // x rows: only the value update of step 7 reads them — to the L2-resident scratch
        d4_t hv; int zr[4], zc;
        int zi = 0, cj = t;
        while (cj > zi) { cj -= zi + 1; ++zi; }
        hh_tile_at(zi, cj, hv, zr, zc, hp);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (zr[q] >= 0 && zc >= 0) Hh[zr[q] * nz + zc] = hv[q];
      };
      if (!ovl) for (int t = wv; t < (ureg ? nxt : nht); t += nw) x_tile(t);
      // u rows (Sh^T, Ruu): the operands of the stage KKT system.  They stay in registers until [A B] is dead and then go
      // straight to their LDS places (Lr, W, ST) — reading them back from the scratch cost a chain of L2 round trips.
      d4_t ures[RIC_U_TILES];
#pragma unroll
      for (int sidx = 0; sidx < RIC_U_TILES; ++sidx) ures[sidx] = d4_t{0, 0, 0, 0};
      if (ovl) {
        // Overlap: the serial part of a knot is the factorisation of Ruu (16x16 Cholesky steps in the registers of ONE
        // wavefront, ~9 us).  Only its own tiles come first (one per wavefront), Lr sits at the end of the then dead G_u
        // region, and wavefront 0 factorises it WHILE the others multiply the Sh^T and x-row tiles of Hh.
        const int nuu = nbm * (nbm + 1) / 2, nux = nbm * nb, nwx = nw - 1;
        const int xt0 = (wv - 1 + nwx - nux % nwx) % nwx;  // the x tiles continue the round-robin of the Sh^T tiles: 25 tiles on 7 wavefronts are 4 rounds, not 5
        d4_t uu = d4_t{0, 0, 0, 0};
        int ubi = 0, ubj = wv;  // lower block (ubi, ubj) of Ruu
        while (ubj > ubi) { ubj -= ubi + 1; ++ubi; }
        if (wv < nuu) {
          int zr[4], zc;
          hh_tile_at(nb + ubi, nb + ubj, uu, zr, zc, hp_uu);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (zr[q] >= 0 && zc >= 0) Hh[zr[q] * nz + zc] = uu[q];  // kept for the inertia-correction path
        }
        __syncthreads();  // G_u is dead
        if (wv < nuu) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = ubi * 16 + (lane >> 4) + 4 * q, j = ubj * 16 + (lane & 15);
            Lr[i * ldr + j] = (i == j && i >= m) ? 1.0 : uu[q];
          }
        }
        __syncthreads();
        RIC_SUB(28);
        if (wv == 0) { const bool ok = chol_blocked_wave(Lr, ldr, nbm, LIr, lane, m); if (lane == 0) iflag[0] = ok ? 1 : 0; }  // (rows >= m of Lr: identity)
        else {
#pragma unroll
          for (int sidx = 0; sidx < RIC_U_TILES; ++sidx) {
            const int e = wv - 1 + sidx * nwx;
            if (e < nux) {
              const int zi = nb + e / nb, cj = e % nb;
              int zr[4], zc;
              hh_tile_at(zi, cj, ures[sidx], zr, zc, sidx == 0 ? hp_ux : nullptr);
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if (zr[q] >= 0 && zc >= 0) Hh[zr[q] * nz + zc] = ures[sidx][q];
            }
          }
          for (int t = xt0; t < nxt; t += nwx) x_tile(t);
        }
        RIC_SUB(29);
        __syncthreads();  // [A B] is dead from here on: W = -[Sh^T | rh], ST = Sh^T lie over it
        if (wv > 0) {
#pragma unroll
          for (int sidx = 0; sidx < RIC_U_TILES; ++sidx) {
            const int e = wv - 1 + sidx * nwx;
            if (e < nux) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int i = (e / nb) * 16 + (lane >> 4) + 4 * q, cp = (e % nb) * 16 + (lane & 15);
                W[i * lw + cp] = -ures[sidx][q]; ST[i * np + cp] = ures[sidx][q];
              }
            }
          }
        }
      } else {
#pragma unroll
      for (int sidx = 0; sidx < RIC_U_TILES; ++sidx) {
        const int t = nxt + wv + sidx * nw;
        if (ureg && t < nht) {
          int zr[4], zc;
          hh_tile(t, ures[sidx], zr, zc);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (zr[q] >= 0 && zc >= 0) Hh[zr[q] * nz + zc] = ures[sidx][q];  // kept for the inertia-correction path
        }
      }
      __syncthreads();
      // ---- 6. stage KKT: Lr = Ruu (lower block triangle, identity padding) ; W = -[Sh^T | rh] (mp x lw, zero padded) ;
      // ST = Sh^T (mp x np)
#pragma unroll
      for (int sidx = 0; sidx < RIC_U_TILES; ++sidx) {
        const int t = nxt + wv + sidx * nw;
        if (ureg && t < nht) {
          int zi = 0, cj = t;
          while (cj > zi) { cj -= zi + 1; ++zi; }
          const int col = lane & 15;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = (zi - nb) * 16 + (lane >> 4) + 4 * q;  // u row (padded index)
            const double v = ures[sidx][q];
            if (cj < nb) { W[i * lw + cj * 16 + col] = -v; ST[i * np + cj * 16 + col] = v; }
            else { const int j = (cj - nb) * 16 + col; Lr[i * ldr + j] = (i == j && i >= m) ? 1.0 : v; }
          }
        }
      }
      }
      if (ureg) for (int idx = tid; idx < mp * 16; idx += nthr) {  // feed-forward column of W and its padding
        const int i = idx >> 4, cc = idx & 15;
        W[i * lw + np + cc] = (cc == 0 && i < m) ? -gh[n + i] : 0.0;
      }
    }
    } else {
    // ---- 5. panels: G_j = Pt AB_j ; Hh[:, j] = H[:, j] + AB^T G_j ----
    for (int cj = 0; cj < nzt; ++cj) {
      for (int r0 = wv; r0 < nb; r0 += 2 * nw) {
        const int r1 = r0 + nw;
        d4_t acc0 = d4_t{0, 0, 0, 0}, acc1 = d4_t{0, 0, 0, 0};
        if (r1 < nb) {
          mma_tile2(acc0, acc1, PT + r0 * 16, PT + r1 * 16, 1, ldp, AB + cj * 16, nzp, 1, np, lane);  // Pt symmetric
          tile_store(GP + (r1 * 16) * 16, 16, acc1, lane);
        } else {
          mma_tile<false>(acc0, PT + r0 * 16, 1, ldp, AB + cj * 16, nzp, 1, np, lane);
        }
        tile_store(GP + (r0 * 16) * 16, 16, acc0, lane);
      }
      __syncthreads();
      const int col_p = cj * 16 + (lane & 15);
      const int zc = (col_p < n) ? col_p : ((col_p >= np && col_p - np < m) ? n + col_p - np : -1);
      for (int zi0 = wv; zi0 < nzt; zi0 += 2 * nw) {
        const int zi1 = zi0 + nw;
        d4_t acc0 = d4_t{0, 0, 0, 0}, acc1 = d4_t{0, 0, 0, 0};
        int zr0[4], zr1[4];
        double h0[4], h1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // H loads are in flight while the matrix cores work
          const int rp0 = zi0 * 16 + (lane >> 4) + 4 * q, rp1 = zi1 * 16 + (lane >> 4) + 4 * q;
          zr0[q] = (rp0 < n) ? rp0 : ((rp0 >= np && rp0 - np < m) ? n + rp0 - np : -1);
          zr1[q] = (zi1 < nzt) ? ((rp1 < n) ? rp1 : ((rp1 >= np && rp1 - np < m) ? n + rp1 - np : -1)) : -1;
          h0[q] = (zr0[q] >= 0 && zc >= 0) ? kn[L.oH + zr0[q] * nz + zc] : 0.0;
          h1[q] = (zr1[q] >= 0 && zc >= 0) ? kn[L.oH + zr1[q] * nz + zc] : 0.0;
        }
        if (zi1 < nzt) mma_tile2(acc0, acc1, AB + zi0 * 16, AB + zi1 * 16, 1, nzp, GP, 16, 1, np, lane);
        else mma_tile<false>(acc0, AB + zi0 * 16, 1, nzp, GP, 16, 1, np, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (zr0[q] >= 0 && zc >= 0) Hh[zr0[q] * nz + zc] = h0[q] + acc0[q];
          if (zr1[q] >= 0 && zc >= 0) Hh[zr1[q] * nz + zc] = h1[q] + acc1[q];
        }
      }
      __syncthreads();
    }
    }
    RIC_PROF(8);
    // ---- 6. stage KKT (AB is dead: R1 is reused) ----
    const bool small_ca = ca <= 16;
    // panel path / many u tiles: Lr = sym(Hh_uu) padded with identity ; W = -[Sh^T | rh] (mp x lw, zero padded) ; ST = Sh^T (mp x np)
    if (!ureg) for (int i = wv; i < mp; i += nw) {
      for (int j = lane; j < mp; j += 64) Lr[i * ldr + j] = (i < m && j < m) ? Hh[(n + (i > j ? i : j)) * nz + n + (i > j ? j : i)] : (i == j ? 1.0 : 0.0);  // lower triangle of Hh_uu
      for (int z = lane; z < lw; z += 64) {
        double sv = 0.0;
        if (i < m) { if (z < n) sv = Hh[(n + i) * nz + z]; else if (z == np) sv = gh[n + i]; }
        W[i * lw + z] = -sv;
        if (z < np) ST[i * np + z] = (z < n) ? sv : 0.0;
      }
    }
    if (small_ca) {
      if (CT_PREFETCH) {
      // CT = [Ca_x | . | dt | .] (16 x lw) ; Y = Da^T (mp x 16): from the registers filled at the top of the knot
#pragma unroll
      for (int ii = 0; ii < CT_ROWS; ++ii) {
        const int i = wv + ii * nw;
#pragma unroll
        for (int zz = 0; zz < 2; ++zz) {
          const int z = lane + 64 * zz;
          if (i < 16 && z < lw) CTl[i * lw + z] = (i < ca && (z < n || z == np)) ? pf_ct[ii][zz] : 0.0;
        }
      }
#pragma unroll
      for (int e = 0; e < Y_ELEMS; ++e) {
        const int idx = tid + e * nthr, i = idx >> 4, j = idx & 15;
        if (idx < mp * 16) Yl[i * RIC_LDY + j] = (i < m && j < ca) ? pf_y[e] : 0.0;
      }
      } else {
      // CT = [Ca_x | . | dt | .] (16 x lw) ; Y = Da^T (mp x 16)
      for (int i = wv; i < 16; i += nw)
        for (int z = lane; z < lw; z += 64) {
          // clamped address, mask by multiplication: the loads of all rows are in flight together (see step 3)
          const int ai = (i < ca) ? act_idx[i] : 0;
          const bool isx = z < n, isd = z == np;
          const double cv = kn[isd ? L.oDT + ai : L.oCD + ai * nz + (isx ? z : 0)];
          CTl[i * lw + z] = cv * ((i < ca && (isx || isd)) ? 1.0 : 0.0);
        }
      for (int idx = tid; idx < mp * 16; idx += nthr) {
        const int i = idx >> 4, j = idx & 15;
        const int aj = (j < ca) ? act_idx[j] : 0;
        Yl[i * RIC_LDY + j] = kn[L.oCD + aj * nz + n + (i < m ? i : 0)] * ((i < m && j < ca) ? 1.0 : 0.0);
      }
      }
    }
    __syncthreads();
    RIC_PROF(9);
    // An indefinite reduced control Hessian (far from the solution the Gauss-Newton + penalty model can lose definiteness
    // through the AL terms) is regularised: Ruu + rho I with rho = max(1e-8, 1e-6 max|diag|) x 10^t until the Cholesky
    // succeeds (same rule in oracle/solver.hpp) — the damped Newton step of ProxDDP's inertia correction.
    bool chol_ok = ovl ? (iflag[0] != 0) : chol_blocked(Lr, ldr, nbm, LIr, tid, iflag);  // ovl: wavefront 0 did it during step 5
    for (int attempt = 1; !chol_ok; ++attempt) {
      if (attempt > 10) { if (tid == 0) a.inst[b].done = 3; return; }
      if (tid == 0) {
        double dmax = 0.0;
        for (int i = 0; i < m; ++i) dmax = fmax(dmax, fabs(Hh[(n + i) * nz + n + i]));
        double rho = fmax(1e-8, 1e-6 * dmax);
        for (int t = 1; t < attempt; ++t) rho *= 10.0;
        wred[0] = rho;
      }
      __syncthreads();
      const double rho = wred[0];
      for (int i = wv; i < mp; i += nw)
        for (int j = lane; j < mp; j += 64)
          Lr[i * ldr + j] = (i < m && j < m) ? Hh[(n + (i > j ? i : j)) * nz + n + (i > j ? j : i)] + (i == j ? rho : 0.0) : (i == j ? 1.0 : 0.0);
      __syncthreads();
      chol_ok = chol_blocked(Lr, ldr, nbm, LIr, tid, iflag);
    }
    RIC_PROF(10);
    // legs: the inverse stage KKT matrix applied to [I; 0] -> Mu (m x m), Znu (ca x m).  Column block j2 of the identity is one more
    // column block next to those of W, solved by wavefront (nwb + j2) % nw from start to end; its operands W2 (mp x mp), VX2 (16 x mp)
    // live in the PT region, which is dead between step 5 (G_x consumed) and step 7 (next P)
    double *W2 = sm + S.K2, *VX2 = W2 + mp * (mp + 1);
    const int ldw2 = mp + 1;
    const bool kkt2 = par && small_ca;
    if (LEGS && kkt2)
      for (int j2 = 0; j2 < nbm; ++j2)
        if (wv == (nwb + j2) % nw) {
          for (int idx = lane; idx < mp * 16; idx += 64) { const int i = idx >> 4, cc = j2 * 16 + (idx & 15); W2[i * ldw2 + cc] = (i == cc && i < m) ? 1.0 : 0.0; }
          trsm_fwd_blocked(Lr, ldr, LIr, nbm, W2 + j2 * 16, ldw2, 1, 0, 1, lane);
        }
    trsm_fwd_blocked(Lr, ldr, LIr, nbm, W, lw, nwb, wv, nw, lane);  // W = L^-1 T
    if (ca > 0 && small_ca && wv == nw - 1) trsm_fwd_blocked(Lr, ldr, LIr, nbm, Yl, RIC_LDY, 1, 0, 1, lane);  // Y = L^-1 Da^T
    __syncthreads();
    RIC_SUB(30);
    if (ca > 0) {
      if (small_ca) {
        // Sc = mu I + Y^T Y (pad identity) ; V = [Ca | dt] + Y^T W
        if (wv == 0) {
          d4_t acc = d4_t{0, 0, 0, 0};
          mma_tile<false>(acc, Yl, 1, RIC_LDY, Yl, RIC_LDY, 1, mp, lane);
          const int col = lane & 15;
          for (int q = 0; q < 4; ++q) {
            const int row = (lane >> 4) + 4 * q;
            double v = acc[q];
            if (row == col) v += (row < ca) ? mu : 1.0;
            SCl[row * 17 + col] = v;
          }
        }
        for (int cj = wv; cj < nwb; cj += nw) {
          d4_t acc = tile_load(CTl + cj * 16, lw, lane);
          mma_tile<false>(acc, Yl, 1, RIC_LDY, W + cj * 16, lw, 1, mp, lane);
          tile_store(VXl + cj * 16, lw, acc, lane);
        }
        if (LEGS && kkt2)
          for (int j2 = 0; j2 < nbm; ++j2)
            if (wv == (nwb + j2) % nw) {
              d4_t acc = d4_t{0, 0, 0, 0};
              mma_tile<false>(acc, Yl, 1, RIC_LDY, W2 + j2 * 16, ldw2, 1, mp, lane);
              tile_store(VX2 + j2 * 16, ldw2, acc, lane);
            }
        __syncthreads();
        if (tid == 0) iflag[0] = 1;
        __syncthreads();
        if (wv == 0) { if (!chol16_wave(SCl, 17, LIs, lane, ca) && lane == 0) iflag[0] = 0; }  // (rows / columns >= ca of Sc are identity)
        __syncthreads();
        if (iflag[0] == 0) { if (tid == 0) a.inst[b].done = 4; return; }
        trsm_fwd_blocked(SCl, 17, LIs, 1, VXl, lw, nwb, wv, nw, lane);
        trsm_bwd_blocked(SCl, 17, LIs, 1, VXl, lw, nwb, wv, nw, lane);
        if (LEGS && kkt2)
          for (int j2 = 0; j2 < nbm; ++j2)
            if (wv == (nwb + j2) % nw) {
              trsm_fwd_blocked(SCl, 17, LIs, 1, VX2 + j2 * 16, ldw2, 1, 0, 1, lane);
              trsm_bwd_blocked(SCl, 17, LIs, 1, VX2 + j2 * 16, ldw2, 1, 0, 1, lane);
              for (int ri = 0; ri < nbm; ++ri) {  // W2 -= Y VX2 (own column block: in order within the wavefront)
                double* Wt = W2 + (ri * 16) * ldw2 + j2 * 16;
                d4_t acc = tile_load(Wt, ldw2, lane);
                mma_tile<true>(acc, Yl + (ri * 16) * RIC_LDY, RIC_LDY, 1, VX2 + j2 * 16, ldw2, 1, 16, lane);
                tile_store(Wt, ldw2, acc, lane);
              }
            }
        __syncthreads();
        // W -= Y V
        for (int t = wv; t < nbm * nwb; t += nw) {
          const int ri = t / nwb, cj = t % nwb;
          double* Wt = W + (ri * 16) * lw + cj * 16;
          d4_t acc = tile_load(Wt, lw, lane);
          mma_tile<true>(acc, Yl + (ri * 16) * RIC_LDY, RIC_LDY, 1, VXl + cj * 16, lw, 1, 16, lane);
          tile_store(Wt, lw, acc, lane);
        }
        __syncthreads();
      } else {
        // many active rows: unblocked path on the L2-resident scratch (rare: more than 16 active rows at one knot)
        double *Ct = wk + L.wCt, *V = wk + L.wV, *Y = wk + L.wY, *Sc = wk + L.wSc;
        for (int idx = tid; idx < ca * n; idx += nthr) Ct[(idx / n) * nz + idx % n] = kn[L.oCD + act_idx[idx / n] * nz + idx % n];
        for (int idx = tid; idx < m * ca; idx += nthr) Y[idx] = kn[L.oCD + act_idx[idx % ca] * nz + n + idx / ca];
        for (int i = tid; i < ca; i += nthr) dtl[i] = kn[L.oDT + act_idx[i]];
        __syncthreads();
        for (int j = tid; j < ca; j += nthr)
          for (int i = 0; i < m; ++i) { double s = Y[i * ca + j]; for (int q = 0; q < i; ++q) s -= Lr[i * ldr + q] * Y[q * ca + j]; Y[i * ca + j] = s / Lr[i * ldr + i]; }
        __syncthreads();
        for (int idx = tid; idx < ca * ca; idx += nthr) {
          const int i = idx / ca, j = idx % ca;
          double s = (i == j) ? mu : 0.0;
          for (int l = 0; l < m; ++l) s += Y[l * ca + i] * Y[l * ca + j];
          Sc[idx] = s;
        }
        for (int idx = tid; idx < ca * nr; idx += nthr) {
          const int i = idx / nr, z = idx % nr;
          double s = (z < n) ? Ct[i * nz + z] : dtl[i];
          for (int l = 0; l < m; ++l) s += Y[l * ca + i] * W[l * lw + ((z < n) ? z : np)];
          V[i * nr + z] = s;
        }
        __syncthreads();
        if (!chol_block(Sc, ca, ca, tid, nthr, iflag)) { if (tid == 0) a.inst[b].done = 4; return; }
        potrs_block(Sc, ca, ca, V, nr, nr, tid, nthr);
        for (int idx = tid; idx < m * nr; idx += nthr) {
          const int l = idx / nr, z = idx % nr;
          double s = 0;
          for (int i = 0; i < ca; ++i) s += Y[l * ca + i] * V[i * nr + z];
          W[l * lw + ((z < n) ? z : np)] -= s;
        }
        __syncthreads();
      }
    }
    RIC_SUB(31);
    trsm_bwd_blocked(Lr, ldr, LIr, nbm, W, lw, nwb, wv, nw, lane);  // U = L^-T (W - Y V)
    if (LEGS && kkt2)
      for (int j2 = 0; j2 < nbm; ++j2)
        if (wv == (nwb + j2) % nw) trsm_bwd_blocked(Lr, ldr, LIr, nbm, W2 + j2 * 16, ldw2, 1, 0, 1, lane);
    __syncthreads();
    RIC_SUB(4);
    if (LEGS && par) {
      const int mpd = L.mpad;  // == mp
      if (small_ca) {
        for (int idx = tid; idx < mp * mp; idx += nthr) { const int i = qdiv(idx, S.mg_mp); g[L.oMu + idx] = W2[i * ldw2 + idx - i * mp]; }
        for (int idx = tid; idx < ca * mp; idx += nthr) { const int i = qdiv(idx, S.mg_mp); g[L.oZnu + idx] = VX2[i * ldw2 + idx - i * mp]; }
      } else {
        // many active rows (rare): the same solve by the unblocked routines on the L2-resident scratch
        double *W2s = wk + L.wAcl, *V2s = wk + L.wLp;  // m x m, ca x m (both free in this kernel)
        const double *Y = wk + L.wY, *Sc = wk + L.wSc;
        for (int idx = tid; idx < m * m; idx += nthr) W2s[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
        __syncthreads();
        for (int j = tid; j < m; j += nthr)
          for (int i = 0; i < m; ++i) { double s2 = W2s[i * m + j]; for (int q = 0; q < i; ++q) s2 -= Lr[i * ldr + q] * W2s[q * m + j]; W2s[i * m + j] = s2 / Lr[i * ldr + i]; }
        __syncthreads();
        for (int idx = tid; idx < ca * m; idx += nthr) {
          const int i = idx / m, z = idx % m;
          double s2 = 0;
          for (int l = 0; l < m; ++l) s2 += Y[l * ca + i] * W2s[l * m + z];
          V2s[idx] = s2;
        }
        __syncthreads();
        potrs_block(Sc, ca, ca, V2s, m, m, tid, nthr);
        for (int idx = tid; idx < m * m; idx += nthr) {
          const int l = idx / m, z = idx % m;
          double s2 = 0;
          for (int i = 0; i < ca; ++i) s2 += Y[l * ca + i] * V2s[i * m + z];
          W2s[idx] -= s2;
        }
        __syncthreads();
        for (int j = tid; j < m; j += nthr)
          for (int i = m - 1; i >= 0; --i) { double s2 = W2s[i * m + j]; for (int q = i + 1; q < m; ++q) s2 -= Lr[q * ldr + i] * W2s[q * m + j]; W2s[i * m + j] = s2 / Lr[i * ldr + i]; }
        __syncthreads();
        for (int idx = tid; idx < mpd * mpd; idx += nthr) { const int i = idx / mpd, j = idx % mpd; g[L.oMu + idx] = (i < m && j < m) ? W2s[i * m + j] : 0.0; }
        for (int idx = tid; idx < ca * mpd; idx += nthr) { const int i = idx / mpd, j = idx % mpd; g[L.oZnu + idx] = (j < m) ? V2s[i * m + j] : 0.0; }
      }
    }
    RIC_PROF(11);
    // gains out: K, k, Knu, knu ; p = qh + Sh k + Ca^T kv
    for (int i = wv; i < m; i += nw) {
      for (int z = lane; z < n; z += 64) g[L.oK + i * n + z] = W[i * lw + z];
      if (lane == 0) g[L.ok + i] = W[i * lw + np];
    }
    // rows of Knu of inactive constraints are left as they are: k_duals reads Knu only where the knot's active flag is set
    for (int i = tid; i < c; i += nthr) g[L.oknu + i] = 0.0;
    __syncthreads();
    if (small_ca) {
      for (int i = wv; i < ca; i += nw) {
        for (int z = lane; z < n; z += 64) g[L.oKnu + act_idx[i] * n + z] = VXl[i * lw + z];
        if (lane == 0) { g[L.oknu + act_idx[i]] = VXl[i * lw + np]; kvc[i] = VXl[i * lw + np]; }
      }
    } else {
      const double* V = wk + L.wV;
      for (int idx = tid; idx < ca * n; idx += nthr) g[L.oKnu + act_idx[idx / n] * n + idx % n] = V[(idx / n) * nr + idx % n];
      for (int i = tid; i < ca; i += nthr) { g[L.oknu + act_idx[i]] = V[i * nr + n]; kvc[i] = V[i * nr + n]; }
    }
    __syncthreads();
    for (int r = tid; r < n; r += nthr) {
      double t = gh[r];
#pragma unroll 8
      for (int i = 0; i < m; ++i) t += ST[i * np + r] * W[i * lw + np];  // (operands of eight terms in flight)
      if (small_ca) {
#pragma unroll 4
        for (int i = 0; i < ca; ++i) t += CTl[i * lw + r] * kvc[i];
      }
      else { const double* Ct = wk + L.wCt; for (int i = 0; i < ca; ++i) t += Ct[i * nz + r] * kvc[i]; }
      g[L.op + r] = t;
      pvec[r] = t;
    }
    RIC_SUB(19);
    // ---- 7. P = Qh + Sh K + Ca^T Kv  ->  PT (next knot's P'): lower block triangle on the MFMA, mirrored into the upper
    // one (P is symmetric); diagonal tiles are symmetrised in place by the wavefront that produced them ----
    if (small_ca) {
      const int kc = (ca + 3) & ~3;
      for (int t = wv; t < nb * (nb + 1) / 2; t += nw) {
        int ri = 0, cj = t;
        while (cj > ri) { cj -= ri + 1; ++ri; }  // t = ri (ri + 1) / 2 + cj, cj <= ri
        const int col = cj * 16 + (lane & 15);
        double qh[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // Hh: lower block triangle (clamped address, mask by multiplication: unconditional loads, all in flight together)
          const int row = ri * 16 + (lane >> 4) + 4 * q;
          qh[q] = Hh[(row < n ? row : 0) * nz + (col < n ? col : 0)] * ((row < n && col < n) ? 1.0 : 0.0);
        }
        d4_t acc = d4_t{0, 0, 0, 0};
        mma_tile<false>(acc, ST + ri * 16, 1, np, W + cj * 16, lw, 1, mp, lane);
        if (kc > 0) mma_tile<false>(acc, CTl + ri * 16, 1, lw, VXl + cj * 16, lw, 1, kc, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = ri * 16 + (lane >> 4) + 4 * q;
          const double pv = (row < n && col < n) ? qh[q] + acc[q] : 0.0;
          PT[row * ldp + col] = pv;
          if (ri != cj) PT[col * ldp + row] = pv;
        }
        if (ri == cj) {
          double tv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; tv[q] = 0.5 * (PT[row * ldp + col] + PT[col * ldp + row]); }
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; PT[row * ldp + col] = tv[q]; }
        }
      }
      __syncthreads();
    } else {
      const double *Ct = wk + L.wCt, *V = wk + L.wV;
      for (int r = wv; r < np; r += nw)
        for (int s = lane; s < np; s += 64) {
          double t = 0.0;
          if (r < n && s < n) {
            t = Hh[(r >= s ? r : s) * nz + (r >= s ? s : r)];  // lower triangle of Hh
            for (int i = 0; i < m; ++i) t += ST[i * np + r] * W[i * lw + s];
            for (int i = 0; i < ca; ++i) t += Ct[i * nz + r] * V[i * nr + s];
          }
          PT[r * ldp + s] = t;
        }
      __syncthreads();
      // symmetrise in LDS (leading dimension np + 1: the transposed read is conflict-free)
      double sv[AB_ROWS][2];
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = wv + nw * q;
        if (i < n) {
          if (lane < n) sv[q][0] = 0.5 * (PT[i * ldp + lane] + PT[lane * ldp + i]);
          if (lane + 64 < n) sv[q][1] = 0.5 * (PT[i * ldp + lane + 64] + PT[(lane + 64) * ldp + i]);
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < AB_ROWS; ++q) {
        const int i = wv + nw * q;
        if (i < n) {
          if (lane < n) PT[i * ldp + lane] = sv[q][0];
          if (lane + 64 < n) PT[i * ldp + lane + 64] = sv[q][1];
        }
      }
      __syncthreads();
    }
    RIC_SUB(25);
    // gain record: P of this knot, row by row (coalesced)
    for (int i = wv; i < n; i += nw)
      for (int j = lane; j < n; j += 64) g[L.oP + i * n + j] = PT[i * ldp + j];
    __syncthreads();
    RIC_PROF(12);
  }
}
