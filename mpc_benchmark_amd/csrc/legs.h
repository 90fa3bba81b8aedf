// legs.h — parallel-in-time proximal Riccati: linear_solver_choice = LQ_SOLVER_PARALLEL + setNumThreads(n) of the reference scripts
// (fulldynamic_talos.py:383-385) mapped on the GPU.  The horizon is cut into `nlegs` legs; workgroup (instance, leg) of the
// Riccati kernel (riccati_mfma.h, LEGS) sweeps its own leg, a leg other than the last from a ZERO value function at its end with
// the co-state theta of the cut state as a parameter (Jallet et al., "Parallel and proximal constrained LQ", 2024).  The
// parametric part is factored so that the sequential kernels stay small (algebra pinned in tests/test_oracle_legs.py):
//
//   per knot, independent of the recursion (k_leg_knot, one workgroup per knot):
//     Bc = T (I - mu_d Pt) B ;  Phi = T (I - mu_d Pt)(A + B K) ;  phi = T (I - mu_d Pt)(B k + mx)
//     Ku = -Mu Bc^T = dk/dp' ;  Knup = -Znu Bc^T = dknu/dp' ;  Gamma = Bc Ku - mu_d T (I - mu_d Pt) T^T = d dx'/dp'
//     (Mu, Znu: (u,u) and (nu,u) blocks of the inverse stage KKT matrix, left by the Riccati kernel)
//   per leg, backwards over its knots (k_leg_condense):  Lm = dp/dtheta,  dx_cut = Lm^T dx + Sg theta + sg
//     Lm_k = Phi_k^T Lm' ;  Sg += Lm'^T Gamma_k Lm' ;  sg += Lm'^T phi_k        (Lm' = I, Sg = 0, sg = 0 at the end of the leg)
//   per instance, over the cuts (k_leg_consensus): exact value function (calP, calp) at every cut, last to first,
//     x_cut = (I - Sg dP)^-1 (Lm^T x + Sg calp + sg) = Zx x + zc ;  then first to last: cut states and theta = dP x_cut + calp
//     (both from the same solve: the stationarity condition at a cut is met like at any other knot) ; exact gain K_0.
//     dP = calP - Pg, Pg = the terminal Hessian the leg carried: with Pg = 0, I - Sg calP mixes the stiffest directions of calP
//     (constraint penalties, 1/mu) with the most controllable ones of the leg and the cut states lose up to ten digits; with
//     Pg = calP of the previous pass / MPC tick the consensus solves for a small correction.  The first pass of a handle has no
//     guess: it sweeps twice (mpc_hip.hip)
//   per knot (k_leg_apply): the affine terms take their final values,  p += Lm theta, k += Ku p', knu += Knup p', phi += Gamma p'
//     with p' = Lm' theta ; the forward sweeps of the legs (closed_loop.h k_forward_phi) and k_duals are then the plain ones.
// Same KKT system as the serial sweep: identical results up to round-off (tests/test_gpu_legs.py).
#pragma once
#include "mfma_blocks.h"
#include "solver_kernels.h"

// workgroup barrier that waits for the LDS traffic only: a __syncthreads() also drains the global loads in flight (the prefetch of the
// next knot) and waits for every global store to be acknowledged.  Used wherever the data exchanged lives in LDS.
#define LEG_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// (per-lane index expressions are kept phase-local by passing the lane ids through an empty asm, as in riccati_mfma.h: hoisted out of
// the knot loop they would be spilled)
#define LEG_LAUNDER() do { asm volatile("" : "+v"(tid), "+v"(lane)); wv = __builtin_amdgcn_readfirstlane(tid >> 6); } while (0)

#define LK_THREADS 512
#define LK_U 20      // global loads in flight per thread in the LDS fills
#define LK_PT 13     // matrix elements per thread parked in registers (np <= 80: 6400 / 512)
#define LK_TILES 5   // output tiles per wavefront: nb (nb + nbm) <= 35 on 8 wavefronts

struct LkLds {
  int np, mp, nzp, lda, ldp, ldm, nb, nbm;  // lda: leading dimension of the [A B] buffer (odd: rows AND columns are read with a lane stride)
  int AB, PR, TM, ZN, vec, total_bytes;
  unsigned mg_nzp, mg_np, mg_mp;  // magic_div of the run-time divisors of the fills
};
// LDS plan (np = 80, mp = 48: 153 KB; mp = 32: 140 KB).  PR is one n x n region with three lives: K (stage 1), Pt (stages 2-3),
// U1 = Mu Bc^T next to Mu (stages 4-5).  TM: rows 0..5 of [M | Bl] (6 x nzp) | columns 0..5 of Pt T^T (np x 6) | 6 x 6 corner of T Pt T^T.
static inline constexpr LkLds make_lk_lds(int n, int m) {
  LkLds s{};
  s.np = (n + 15) & ~15; s.mp = (m + 15) & ~15; s.nzp = s.np + s.mp; s.lda = s.nzp + 1; s.ldp = s.np + 1; s.ldm = s.mp + 1;
  s.nb = s.np / 16; s.nbm = s.mp / 16;
  s.mg_nzp = magic_div(s.nzp); s.mg_np = magic_div(s.np); s.mg_mp = magic_div(s.mp);
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
  const int pr = s.np * s.ldp > s.mp * s.np + s.mp * s.ldm ? s.np * s.ldp : s.mp * s.np + s.mp * s.ldm;
  s.AB = take(s.np * s.lda); s.PR = take(pr); s.TM = take(6 * s.nzp + 6 * s.np + 40); s.ZN = take(16 * s.ldm); s.vec = take(s.mp + 3 * s.np + 80 + 8);
  s.total_bytes = o * 8;
  return s;
}

// grid (ceil(N / chunk), B): workgroup (ck, b) walks knots ck * chunk .. of instance b.  Reads K, k, Pt, mx, T6, Mu, Znu (gain record) and
// [A B] (knot record); writes Phi, phi for every knot and Gamma, Ku, Knup for the knots of parametric legs.  Everything the NEXT knot of
// the chunk needs is requested into registers while this one computes (a workgroup owns its CU: 140 - 153 KB of LDS), so only the first
// knot of a chunk waits for HBM.  MP: padded control dimension (sizes the prefetch registers).
// FN, FM > 0: state / control dimensions as compile-time constants (see k_riccati_mfma)
template <int MP, int FN = 0, int FM = 0>
__global__ void __launch_bounds__(LK_THREADS) k_leg_knot(SolverArgs a, LkLds Srt, int chunk) {
  constexpr bool FX = FN > 0;
  constexpr LkLds SC_ = FX ? make_lk_lds(FN, FM) : LkLds{};
  LkLds S_ = Srt;
  if constexpr (FX) S_ = SC_;
  const LkLds& S = S_;
  constexpr int NAB = (80 * (80 + MP) + LK_THREADS - 1) / LK_THREADS, NK = (MP * 80 + LK_THREADS - 1) / LK_THREADS;
  constexpr int NMU = (MP * MP + LK_THREADS - 1) / LK_THREADS, NZN = (16 * MP + LK_THREADS - 1) / LK_THREADS;
  const Layout& L = a.L;
  const int b = blockIdx.y;
  int tid = threadIdx.x, lane = tid & 63;
  constexpr int nthr = LK_THREADS, nw = LK_THREADS / 64;
  int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = FX ? FN : L.n, nz = FX ? FN + FM : L.nz, np = S.np, mp = S.mp, nzp = S.nzp, lda = S.lda, ldp = S.ldp, ldm = S.ldm, nb = S.nb, nbm = S.nbm;
  const int kbeg = blockIdx.x * chunk, kend = (kbeg + chunk < L.N) ? kbeg + chunk : L.N;
  const double mud = st.mu * a.opt.dyn_al_scale;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *AB = sm + S.AB, *PR = sm + S.PR, *ZN = sm + S.ZN, *vec = sm + S.vec;
  double *KM = PR, *PT = PR, *U1 = PR, *MU = PR + mp * np;  // the three lives of PR
  double *TMP = sm + S.TM, *PT6 = TMP + 6 * nzp, *c6 = PT6 + 6 * np;
  double *kf = vec, *y0 = vec + mp, *z0 = y0 + np, *t6 = z0 + np, *g6 = t6 + 36, *mxs = g6 + 36;  // k (mp), B k + mx (np), (I - mu_d Pt) y0 (np), T6, T6 T6^T, mx (np)
  int* cnt = (int*)(mxs + np);  // active rows counted per wavefront (nw ints)
  // developer phase timers (mpc_profile(3)): the workgroup of (knot 1, instance 1) -> slots 32.. of instance 1's counter block
  long long tk0_ = clock64();
#define LK_PROF(slot) do { if (a.prof && k == 1 && b == 1 && tid == 0) { const long long t1_ = clock64(); a.prof[64 + 32 + (slot)] += (double)(t1_ - tk0_); tk0_ = t1_; } } while (0)

  // ---- the operands of a knot, as the registers of the prefetch hold them ----
  double pab[NAB], pkm[NK], ptv[LK_PT], pmu[NMU], pzn[NZN], psm = 0.0, pact = 0.0;  // psm: k / T6 / mx element of this thread
  auto request_ab = [&](int k) {  // [A B], K, the small vectors, the active-row flags
    const double* kn = knot_ptr(a, b, k);
    const double* g = gain_ptr(a, b, k);
    const int m = (int)kn[L.oMISC + MISC_M], c = (int)kn[L.oMISC + MISC_NC];
#pragma unroll
    for (int u = 0; u < NAB; ++u) {
      const int idx = tid + u * nthr, i = qdiv(idx, S.mg_nzp), zp = (idx - qdiv(idx, S.mg_nzp) * nzp);
      const int z = (zp < n) ? zp : ((zp >= np && zp - np < m) ? n + zp - np : -1);
      pab[u] = (idx < np * nzp && i < n && z >= 0) ? kn[L.oAB + (size_t)i * nz + z] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < NK; ++u) {
      const int idx = tid + u * nthr, l = qdiv(idx, S.mg_np), j = (idx - qdiv(idx, S.mg_np) * np);
      pkm[u] = (idx < mp * np && l < m && j < n) ? g[L.oK + l * n + j] : 0.0;
    }
    // threads 0..mp-1: k ; 64..99: T6 ; 128..128+np-1: mx
    psm = (tid < mp) ? ((tid < m) ? g[L.ok + tid] : 0.0) : ((tid >= 64 && tid < 100) ? g[L.oT6 + tid - 64] : ((tid >= 128 && tid < 128 + n) ? g[L.omx + tid - 128] : 0.0));
    pact = (tid < c) ? kn[L.oACT + tid] : 0.0;  // c <= LK_THREADS (checked by the host)
  };
  auto request_pt = [&](int k) {  // Pt, Mu, Znu
    const double* g = gain_ptr(a, b, k);
#pragma unroll
    for (int u = 0; u < LK_PT; ++u) {
      const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), j = (idx - qdiv(idx, S.mg_np) * np);
      ptv[u] = (idx < np * np && i < n && j < n) ? g[L.oMx + i * n + j] : 0.0;
    }
  };
  auto request_mu = [&](int k) {
    const double* g = gain_ptr(a, b, k);
#pragma unroll
    for (int u = 0; u < NMU; ++u) { const int idx = tid + u * nthr; pmu[u] = (idx < mp * mp) ? g[L.oMu + idx] : 0.0; }
#pragma unroll
    for (int u = 0; u < NZN; ++u) { const int idx = tid + u * nthr; pzn[u] = (idx < 16 * mp) ? g[L.oZnu + idx] : 0.0; }
  };

  request_ab(kbeg);
  request_pt(kbeg);
  if (leg_of_knot(a, kbeg) + 1 < a.nlegs) request_mu(kbeg);
  for (int k = kbeg; k < kend; ++k) {
    const double* kn = knot_ptr(a, b, k);
    double* g = gain_ptr(a, b, k);
    const int m = (int)kn[L.oMISC + MISC_M];
    const bool par = leg_of_knot(a, k) + 1 < a.nlegs;
    const bool more = k + 1 < kend, par_next = more && leg_of_knot(a, k + 1) + 1 < a.nlegs;
    LEG_BARRIER();  // the previous knot of the chunk is done with every buffer
    LEG_LAUNDER();
    if (a.prof) tk0_ = clock64();
    // ---- stage 1: [A B] and K into LDS (zero padded; u-columns of [A B] start at np) ; A_cl = A + B K in place ; y0 = B k + mx ----
#pragma unroll
    for (int u = 0; u < NAB; ++u) { const int idx = tid + u * nthr; if (idx < np * nzp) AB[(qdiv(idx, S.mg_nzp)) * lda + (idx - qdiv(idx, S.mg_nzp) * nzp)] = pab[u]; }
#pragma unroll
    for (int u = 0; u < NK; ++u) { const int idx = tid + u * nthr; if (idx < mp * np) KM[idx] = pkm[u]; }
    if (tid < mp) kf[tid] = psm;
    else if (tid >= 64 && tid < 100) t6[tid - 64] = psm;
    else if (tid >= 128 && tid < 128 + np) mxs[tid - 128] = psm;
    int ca = 0;
    if (par) {
      const unsigned long long act = __ballot(pact != 0.0);
      if (lane == 0) cnt[wv] = __popcll(act);
    }
    LEG_BARRIER();
    if (par) for (int w = 0; w < nw; ++w) ca += cnt[w];
    LEG_LAUNDER();
    if (more) request_ab(k + 1);  // lands behind stages 1 - 5
    LEG_LAUNDER();
    LK_PROF(0);
    for (int t = wv; t < nb * nb; t += nw) {
      const int ri = t / nb, cj = t % nb;
      double* At = AB + (ri * 16) * lda + cj * 16;
      d4_t acc = tile_load(At, lda, lane);                                                           // A tile
      mma_tile<false>(acc, AB + (ri * 16) * lda + np, lda, 1, KM + cj * 16, np, 1, mp, lane);        // + B K (own tile only: in place)
      tile_store(At, lda, acc, lane);
    }
    for (int i = wv; i < np; i += nw) {  // one row per wavefront, lanes over the controls
      double s = (lane < m) ? AB[i * lda + np + lane] * kf[lane] : 0.0;
      s = wave_sum(s);
      if (lane == 0) y0[i] = s + mxs[i];
    }
    if (tid < 36) { const int i = tid / 6, j = tid % 6; double s = 0; for (int l = 0; l < 6; ++l) s += t6[i * 6 + l] * t6[j * 6 + l]; g6[tid] = s; }
    LEG_BARRIER();  // K is dead: Pt takes its place
    LEG_LAUNDER();
    // ---- stage 2: Pt into LDS ; [M | Bl] = (I - mu_d Pt) [A_cl | B] (products to registers, then in place) ; z0 = y0 - mu_d Pt y0 ----
#pragma unroll
    for (int u = 0; u < LK_PT; ++u) { const int idx = tid + u * nthr; if (idx < np * np) PT[(qdiv(idx, S.mg_np)) * ldp + (idx - qdiv(idx, S.mg_np) * np)] = ptv[u]; }
    LEG_BARRIER();
    LEG_LAUNDER();
    if (more) request_pt(k + 1);
    LEG_LAUNDER();
    LK_PROF(1);
    const int nct = par ? nb + nbm : nb;  // the u columns are only needed for the parametric quantities
    d4_t res[LK_TILES];
#pragma unroll
    for (int sidx = 0; sidx < LK_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      res[sidx] = d4_t{0, 0, 0, 0};
      if (t < nb * nct) mma_tile<false>(res[sidx], PT + ((t / nct) * 16) * ldp, ldp, 1, AB + (t % nct) * 16, lda, 1, np, lane);
    }
    for (int i = wv; i < n; i += nw) {
      double s = 0;
      for (int j = lane; j < n; j += 64) s += PT[i * ldp + j] * y0[j];
      s = wave_sum(s);
      if (lane == 0) z0[i] = y0[i] - mud * s;
    }
    LEG_BARRIER();
#pragma unroll
    for (int sidx = 0; sidx < LK_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nb * nct) {
        double* At = AB + ((t / nct) * 16) * lda + (t % nct) * 16;
        const int row = lane >> 4, col = lane & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) At[(row + 4 * q) * lda + col] -= mud * res[sidx][q];
      }
    }
    LEG_BARRIER();
    LK_PROF(2);
    // ---- stage 3: base rows through T6 (rows 0..5 of [M | Bl] -> [Phi | Bc]) ; Phi, phi out ; columns 0..5 of Pt T^T kept for Gamma ----
    for (int idx = tid; idx < 6 * nzp; idx += nthr) TMP[idx] = AB[(qdiv(idx, S.mg_nzp)) * lda + (idx - qdiv(idx, S.mg_nzp) * nzp)];
    if (par) for (int idx = tid; idx < np * 6; idx += nthr) {  // (Pt T^T)[i][j] = sum_l Pt[i][l] T6[j][l]
      const int i = idx / 6, j = idx % 6;
      double s = 0;
      for (int l = 0; l < 6; ++l) s += PT[i * ldp + l] * t6[j * 6 + l];
      PT6[idx] = s;
    }
    LEG_BARRIER();
    for (int idx = tid; idx < 6 * nzp; idx += nthr) {
      const int i = qdiv(idx, S.mg_nzp), z = (idx - qdiv(idx, S.mg_nzp) * nzp);
      double s = 0;
      for (int l = 0; l < 6; ++l) s += t6[i * 6 + l] * TMP[l * nzp + z];
      AB[i * lda + z] = s;
    }
    if (par && tid < 36) {  // 6 x 6 corner of T (Pt T^T)
      const int i = tid / 6, j = tid % 6;
      double s = 0;
      for (int l = 0; l < 6; ++l) s += t6[i * 6 + l] * PT6[l * 6 + j];
      c6[tid] = s;
    }
    LEG_BARRIER();
    for (int i = wv; i < n; i += nw)
      for (int j = lane; j < n; j += 64) g[L.oPhi + i * n + j] = AB[i * lda + j];
    for (int i = tid; i < n; i += nthr) {
      double s = z0[i];
      if (i < 6) { s = 0; for (int l = 0; l < 6; ++l) s += t6[i * 6 + l] * z0[l]; }
      g[L.ophi + i] = s;
    }
    LK_PROF(3);
    LEG_LAUNDER();
    if (par) {
      // ---- stage 4: Mu (into PR behind the place of U1: Pt is dead), Znu ; U1 = Mu Bc^T (-> Ku = -U1, kept for stage 5) ; V1 = Znu Bc^T (-> Knup = -V1) ----
#pragma unroll
      for (int u = 0; u < NMU; ++u) { const int idx = tid + u * nthr; if (idx < mp * mp) MU[(qdiv(idx, S.mg_mp)) * ldm + (idx - qdiv(idx, S.mg_mp) * mp)] = pmu[u]; }
#pragma unroll
      for (int u = 0; u < NZN; ++u) { const int idx = tid + u * nthr; if (idx < 16 * mp) ZN[(qdiv(idx, S.mg_mp)) * ldm + (idx - qdiv(idx, S.mg_mp) * mp)] = (qdiv(idx, S.mg_mp) < ca && ca <= 16) ? pzn[u] : 0.0; }
      LEG_BARRIER();
      LEG_LAUNDER();
      if (par_next) request_mu(k + 1);
      LEG_LAUNDER();
      LK_PROF(4);
      d4_t ures[2];
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int t = wv + sidx * nw;
        ures[sidx] = d4_t{0, 0, 0, 0};
        if (t < nbm * nb) mma_tile<false>(ures[sidx], MU + ((t / nb) * 16) * ldm, ldm, 1, AB + ((t % nb) * 16) * lda + np, 1, lda, mp, lane);
      }
#ifdef LK_SUBPROF
      LK_PROF(7);
#endif
      if (ca <= 16) {
        for (int cj = wv; cj < nb; cj += nw) {
          d4_t acc = d4_t{0, 0, 0, 0};
          mma_tile<false>(acc, ZN, ldm, 1, AB + (cj * 16) * lda + np, 1, lda, mp, lane);
          const int col = cj * 16 + (lane & 15);
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = (lane >> 4) + 4 * q; if (row < ca && col < n) g[L.oKnup + row * n + col] = -acc[q]; }
        }
      } else {
        for (int idx = tid; idx < ca * n; idx += nthr) {
          const int i = idx / n, j = idx % n;
          double s = 0;
          for (int l = 0; l < m; ++l) s += g[L.oZnu + i * mp + l] * AB[j * lda + np + l];
          g[L.oKnup + idx] = -s;
        }
      }
#ifdef LK_SUBPROF
      LK_PROF(8);
#endif
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int t = wv + sidx * nw;
        if (t < nbm * nb) {
          const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = ri * 16 + (lane >> 4) + 4 * q;
            U1[row * np + col] = ures[sidx][q];  // beside Mu (other wavefronts may still read it), not over it
            if (row < m && col < n) g[L.oKu + row * n + col] = -ures[sidx][q];
          }
        }
      }
#ifdef LK_SUBPROF
      LK_PROF(9);
#endif
      LEG_BARRIER();
      LK_PROF(5);
      // ---- stage 5: Gamma = -Bc U1 - mu_d (T T^T - mu_d T Pt T^T) ; T Pt T^T: rows / columns 0..5 from PT6 / c6 (symmetric), the rest is Pt
      // itself, read again from the gain record (L2) while the matrix cores work ----
      for (int t = wv; t < nb * nb; t += nw) {
        const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
        double ptt[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = ri * 16 + (lane >> 4) + 4 * q;
          const bool in = row < n && col < n;
          ptt[q] = (in && row >= 6 && col >= 6) ? g[L.oMx + row * n + col] : 0.0;
        }
        d4_t acc = d4_t{0, 0, 0, 0};
        mma_tile<false>(acc, AB + (ri * 16) * lda + np, lda, 1, U1 + cj * 16, np, 1, mp, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = ri * 16 + (lane >> 4) + 4 * q;
          if (row < n && col < n) {
            const double tt = (row < 6 && col < 6) ? g6[row * 6 + col] : (row == col ? 1.0 : 0.0);
            const double tp = (row < 6 && col < 6) ? c6[row * 6 + col] : (col < 6 ? PT6[row * 6 + col] : (row < 6 ? PT6[col * 6 + row] : ptt[q]));
            g[L.oGam + row * n + col] = -acc[q] - mud * (tt - mud * tp);
          }
        }
      }
      LK_PROF(6);
    } else if (par_next) request_mu(k + 1);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_leg_condense: grid (nlegs - 1, B), leg j of instance b, backwards over its knots.  Lm (this knot's dp/dtheta) stays in LDS, Sg in the
// accumulator registers of the matrix cores over the whole leg; Phi and Gamma of the next knot are requested while this one computes.
// ---------------------------------------------------------------------------------------------------------------------
struct LcLds { int np, ldp, nb; int LM, BA, BB, vec, total_bytes; unsigned mg_np, mg_ldp; };
// Leading dimension of the three operands: every matrix-core read of this kernel takes 16 CONSECUTIVE doubles of 4 rows per instruction
// (ds_read_b64: two groups of 32 lanes, 64 banks of 4 bytes), so two rows of a group must lie half a bank row apart: ld = 16 mod 32
// doubles (np = 80 itself).  With np + 1 (round 3) row k + 1 began two banks after row k ended its wrap: a 2-way conflict on EVERY operand
// read (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 30 %, profiles/r03_sq_counters.txt).  Gamma is read through its transpose for that
// (it is symmetric: -Bc Mu Bc^T - mu_d T (I - mu_d Pt) T^T).
static inline constexpr int lc_ld(int np) { int ld = np; while ((ld & 31) != 16) ++ld; return ld; }
static inline constexpr LcLds make_lc_lds(int n) {
  LcLds s{};
  s.np = (n + 15) & ~15; s.ldp = lc_ld(s.np); s.nb = s.np / 16;
  s.mg_np = magic_div(s.np); s.mg_ldp = magic_div(s.ldp);
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
  s.LM = take(s.np * s.ldp); s.BA = take(s.np * s.ldp); s.BB = take(s.np * s.ldp); s.vec = take(3 * s.np + 16);
  s.total_bytes = o * 8;
  return s;
}
#define LC_TILES 4   // nb^2 <= 25 output tiles on 8 wavefronts
#define LC_STILES 2  // nb (nb + 1) / 2 <= 15 lower-triangle tiles of Sg

template <int FN = 0>
__global__ void __launch_bounds__(LK_THREADS) k_leg_condense(SolverArgs a, LcLds Srt) {
  constexpr bool FX = FN > 0;
  constexpr LcLds SC_ = FX ? make_lc_lds(FN) : LcLds{};
  LcLds S_ = Srt;
  if constexpr (FX) S_ = SC_;
  const LcLds& S = S_;
  const Layout& L = a.L;
  constexpr int nthr = LK_THREADS, nw = LK_THREADS / 64;
  const int j = blockIdx.x, b = blockIdx.y;
  int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = FX ? FN : L.n, np = S.np, ldp = S.ldp, nb = S.nb;
  const int ks = leg_start(a, j), ke = leg_start(a, j + 1) - 1;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *LM = sm + S.LM, *BA = sm + S.BA, *BB = sm + S.BB, *sg = sm + S.vec, *phi = sg + np;
  for (int idx = tid; idx < np * ldp; idx += nthr) { const int i = qdiv(idx, S.mg_ldp), c0 = (idx - qdiv(idx, S.mg_ldp) * ldp); LM[idx] = (i == c0 && i < n) ? 1.0 : 0.0; }
  for (int i = tid; i < np; i += nthr) sg[i] = 0.0;
  d4_t sacc[LC_STILES];
  int tri[LC_STILES], tcj[LC_STILES];
#pragma unroll
  for (int sidx = 0; sidx < LC_STILES; ++sidx) {
    sacc[sidx] = d4_t{0, 0, 0, 0};
    int ri = 0, rem = wv + sidx * nw;
    while (rem > ri) { rem -= ri + 1; ++ri; }  // t = ri (ri + 1) / 2 + cj, cj <= ri
    tri[sidx] = ri; tcj[sidx] = rem;
  }
  const int nst = nb * (nb + 1) / 2;
  double pa[LK_PT], pb[LK_PT], pphi;
  auto prefetch = [&](int kk) {
    const double* gk = gain_ptr(a, b, kk);
#pragma unroll
    for (int u = 0; u < LK_PT; ++u) {
      const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), c0 = (idx - qdiv(idx, S.mg_np) * np);
      const bool ok = idx < np * np && i < n && c0 < n;
      const int src = ok ? i * n + c0 : 0;
      pa[u] = gk[L.oPhi + src] * (ok ? 1.0 : 0.0);
      pb[u] = gk[L.oGam + src] * (ok ? 1.0 : 0.0);
    }
    pphi = gk[L.ophi + (tid < n ? tid : 0)];
  };
  prefetch(ke);
  long long tc0_ = clock64();
#define LC_PROF(slot) do { if (a.prof && j == 0 && b == 2 && tid == 0) { const long long t1_ = clock64(); a.prof[128 + 32 + (slot)] += (double)(t1_ - tc0_); tc0_ = t1_; } } while (0)
  for (int k = ke; k >= ks; --k) {
    LEG_LAUNDER();
#pragma unroll
    for (int u = 0; u < LK_PT; ++u) {
      const int idx = tid + u * nthr;
      if (idx < np * np) { BA[(qdiv(idx, S.mg_np)) * ldp + (idx - qdiv(idx, S.mg_np) * np)] = pa[u]; BB[(qdiv(idx, S.mg_np)) * ldp + (idx - qdiv(idx, S.mg_np) * np)] = pb[u]; }
    }
    if (tid < np) phi[tid] = (tid < n) ? pphi : 0.0;
    LEG_BARRIER();
    LC_PROF(0);
    LEG_LAUNDER();
    if (k > ks) prefetch(k - 1);
    LEG_LAUNDER();
    // Lm_k = Phi^T Lm' ; Z = Gamma Lm' (both to registers) ; sg += Lm'^T phi
    d4_t lres[LC_TILES], zres[LC_TILES];
#pragma unroll
    for (int sidx = 0; sidx < LC_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      lres[sidx] = d4_t{0, 0, 0, 0}; zres[sidx] = d4_t{0, 0, 0, 0};
      if (t < nb * nb)  // both products read the same columns of Lm': one pass, two accumulator chains
        mma_tile_2a(lres[sidx], zres[sidx], BA + (t / nb) * 16, 1, ldp, BB + (t / nb) * 16, 1, ldp, LM + (t % nb) * 16, ldp, 1, np, lane);  // (Gamma^T = Gamma)
    }
    LC_PROF(1);
    // (by the last two wavefronts — three tiles each above, the first has four — and over the zero-padded np rows with sixteen operands in
    // flight: a loop of n dependent LDS round trips on wavefronts 0 and 1 kept the other six at the barrier for 2.4 us per knot)
    {
      const int ts = tid - (nthr - 128);
      if (ts >= 0 && ts < n) {
        double s = 0;
#pragma unroll 16
        for (int l = 0; l < np; ++l) s += LM[l * ldp + ts] * phi[l];
        sg[ts] += s;
      }
    }
    LC_PROF(2);
    LEG_BARRIER();
    LC_PROF(3);
    LEG_LAUNDER();
#pragma unroll
    for (int sidx = 0; sidx < LC_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nb * nb) tile_store(BB + ((t / nb) * 16) * ldp + (t % nb) * 16, ldp, zres[sidx], lane);
    }
    LEG_BARRIER();
    LC_PROF(4);
    LEG_LAUNDER();
    // Sg += Lm'^T Z on the lower block triangle
#pragma unroll
    for (int sidx = 0; sidx < LC_STILES; ++sidx)
      if (wv + sidx * nw < nst) mma_tile<false>(sacc[sidx], LM + tri[sidx] * 16, 1, ldp, BB + tcj[sidx] * 16, ldp, 1, np, lane);
    LC_PROF(5);
    LEG_BARRIER();
    LC_PROF(6);
    LEG_LAUNDER();
    double* gk = gain_ptr(a, b, k);
#pragma unroll
    for (int sidx = 0; sidx < LC_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nb * nb) {
        const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
        tile_store(LM + (ri * 16) * ldp + cj * 16, ldp, lres[sidx], lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; if (row < n && col < n) gk[L.oLm + row * n + col] = lres[sidx][q]; }
      }
    }
    LEG_BARRIER();
    LC_PROF(7);
  }
  double* lr = leg_ptr(a, b, j);
#pragma unroll
  for (int sidx = 0; sidx < LC_STILES; ++sidx)
    if (wv + sidx * nw < nst) {
      const int col = tcj[sidx] * 16 + (lane & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = tri[sidx] * 16 + (lane >> 4) + 4 * q;
        if (row < n && col < n) {
          if (tri[sidx] != tcj[sidx]) { lr[L.lSg + row * n + col] = sacc[sidx][q]; lr[L.lSg + col * n + row] = sacc[sidx][q]; }
          else if (col <= row) { lr[L.lSg + row * n + col] = sacc[sidx][q]; lr[L.lSg + col * n + row] = sacc[sidx][q]; }  // diagonal tile: its lower triangle, mirrored
        }
      }
    }
  for (int i = tid; i < n; i += nthr) lr[L.lsg + i] = sg[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// k_leg_consensus: grid B.  Backward over the cuts: exact value function at the start of every leg and the linear map of the cut
// state, x_{j+1} = Zx_j x_j + zc_j (Gauss-Jordan with partial pivoting on I - Sg calP, implicit row permutation); then forward over
// the cuts: cut states (straight into dxs) and co-states theta ; last, the exact feedback gain of knot 0 (controlFeedbacks()[0]).
// ---------------------------------------------------------------------------------------------------------------------
struct LxLds { int np, mp, ldp, nb, nbm; int PC, MA, RB, vec, iw, total_bytes; unsigned mg_np, mg_ldp; };
static inline constexpr LxLds make_lx_lds(int n, int m) {
  LxLds s{};
  s.np = (n + 15) & ~15; s.mp = (m + 15) & ~15; s.ldp = s.np + 1; s.nb = s.np / 16; s.nbm = s.mp / 16;
  s.mg_np = magic_div(s.np); s.mg_ldp = magic_div(s.ldp);
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += (cnt + 1) & ~1; return r; };
  s.PC = take(s.np * s.ldp); s.MA = take(s.np * s.ldp); s.RB = take(s.np * s.ldp);
  s.vec = take(8 * s.np + 16 + 128);  // (+ 112: with the free tail of the vectors, the panel / pivot-block scratch of the blocked elimination of k_leg_compose)
  s.iw = o;
  s.total_bytes = o * 8 + (2 * s.np + 8) * 4;
  return s;
}

// n x n matrix (row-major, leading dimension n) from global memory into an LDS buffer (np x np used, leading dimension ldp, zero padded),
// optionally transposed; every load of a thread is in flight before the first LDS write (a plain strided loop waits per element).
// `add`: an LDS matrix of the same shape added on the way (dst = src + add) — may be dst itself.
template <bool TR>
DEV void leg_load_mat(double* dst, int ldp, int np, const double* src, int n, int tid, int nthr, unsigned mg_np) {
  const struct { unsigned mg_np; } S = {mg_np};
  double v[LK_PT];
#pragma unroll
  for (int u = 0; u < LK_PT; ++u) {
    const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), c0 = (idx - qdiv(idx, S.mg_np) * np);
    const bool ok = idx < np * np && i < n && c0 < n;
    v[u] = src[ok ? i * n + c0 : 0] * (ok ? 1.0 : 0.0);  // clamped address, mask by multiplication: unconditional loads
  }
#pragma unroll
  for (int u = 0; u < LK_PT; ++u) {
    const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), c0 = (idx - qdiv(idx, S.mg_np) * np);
    if (idx < np * np) dst[TR ? c0 * ldp + i : i * ldp + c0] = v[u];
  }
}

// the same in two halves, for a load whose latency is to hide behind other work: request into registers now, store into LDS later
DEV void leg_request_mat(double (&v)[LK_PT], int np, const double* src, int n, int tid, int nthr, unsigned mg_np) {
#pragma unroll
  for (int u = 0; u < LK_PT; ++u) {
    const int idx = tid + u * nthr, i = qdiv(idx, mg_np), c0 = (idx - qdiv(idx, mg_np) * np);
    const bool ok = idx < np * np && i < n && c0 < n;
    v[u] = src[ok ? i * n + c0 : 0] * (ok ? 1.0 : 0.0);
  }
}
DEV void leg_store_mat(double* dst, int ldp, int np, const double (&v)[LK_PT], int tid, int nthr, unsigned mg_np) {
#pragma unroll
  for (int u = 0; u < LK_PT; ++u) {
    const int idx = tid + u * nthr, i = qdiv(idx, mg_np), c0 = (idx - qdiv(idx, mg_np) * np);
    if (idx < np * np) dst[i * ldp + c0] = v[u];
  }
}

// broadcast of a double from a wave-uniform lane (v_readlane with a scalar lane index)
DEV double readlane_dyn(double v, int src_lane) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// phase timers of the consensus kernel (developer tooling, mpc_profile(3)): slots 23..29 of the instance's counter block
#define LEG_PROF(slot) do { LEG_LAUNDER(); if (tid == 0 && a.prof) { const long long t1_ = clock64(); a.prof[(size_t)b * 64 + (slot)] += (double)(t1_ - t0_); t0_ = t1_; } } while (0)
// NP: padded state dimension (S.np), a template parameter so that the rows of the elimination are unrolled without guards
template <int NP>
__global__ void __launch_bounds__(LK_THREADS) k_leg_consensus(SolverArgs a, LxLds S) {
  const Layout& L = a.L;
  const int b = blockIdx.x, nthr = blockDim.x, nw = nthr >> 6;
  int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, N = L.N, np = S.np, mp = S.mp, ldp = S.ldp, nb = S.nb, nbm = S.nbm, J = a.nlegs;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double *PC = sm + S.PC, *MA = sm + S.MA, *RB = sm + S.RB, *vec = sm + S.vec;
  double *pc = vec, *rv = vec + np, *ev = vec + 2 * np, *fcol = vec + 4 * np, *xv = vec + 5 * np;  // calp | right-hand side / zc | scratch | 1 / pivots | cut state
  int* perm = (int*)(sm + S.iw);  // perm[col] = row that was the pivot of column col
  int* used = perm + np;          // used[row] != 0: row already served as a pivot
  d4_t res[LC_TILES];
  long long t0_ = clock64();
  // value function at the start of the last leg
  {
    const double* gl = gain_ptr(a, b, leg_start(a, J - 1));
    leg_load_mat<false>(PC, ldp, np, gl + L.oP, n, tid, nthr, S.mg_np);
    for (int i = tid; i < np; i += nthr) pc[i] = (i < n) ? gl[L.op + i] : 0.0;
  }
  LEG_BARRIER();
  for (int j = J - 2; j >= 0; --j) {
    LEG_LAUNDER();
    double* lr = leg_ptr(a, b, j);
    double* gs = gain_ptr(a, b, leg_start(a, j));
    // calP_{j+1} out (the next pass starts leg j from it) ; PC <- dP = calP_{j+1} - (the guess leg j carried in this pass): the leg's
    // terminal gradient is Pg x + theta, so theta = dP x_cut + calp ; dP, calp out (the forward part computes theta from them) ;
    // MA <- Sg ; RB <- Lm^T
    {
      double po[LK_PT];
#pragma unroll
      for (int u = 0; u < LK_PT; ++u) {
        const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), c0 = (idx - qdiv(idx, S.mg_np) * np);
        const bool ok = a.leg_guess && idx < np * np && i < n && c0 < n;
        po[u] = lr[L.lcP + (ok ? i * n + c0 : 0)] * (ok ? 1.0 : 0.0);
      }
      leg_load_mat<false>(MA, ldp, np, lr + L.lSg, n, tid, nthr, S.mg_np);
      leg_load_mat<true>(RB, ldp, np, gs + L.oLm, n, tid, nthr, S.mg_np);  // coalesced read, transposed write (odd ld: no conflicts)
#pragma unroll
      for (int u = 0; u < LK_PT; ++u) {
        const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), c0 = (idx - qdiv(idx, S.mg_np) * np);
        if (idx < np * np && i < n && c0 < n) {
          const double pnew = PC[i * ldp + c0], d = pnew - po[u];
          lr[L.lcP + i * n + c0] = pnew; lr[L.ldP + i * n + c0] = d; PC[i * ldp + c0] = d;
        }
      }
    }
    for (int i = tid; i < n; i += nthr) lr[L.lcp + i] = pc[i];
    LEG_BARRIER();
    LEG_PROF(23);
    // rv = Sg calp + sg ; Mt = I - Sg calP (to registers, then over Sg)
    for (int i = wv; i < np; i += nw) {
      double s = 0;
      for (int c0 = lane; c0 < n; c0 += 64) s += MA[i * ldp + c0] * pc[c0];
      s = wave_sum(s);
      if (lane == 0) rv[i] = (i < n) ? s + lr[L.lsg + i] : 0.0;
    }
#pragma unroll
    for (int sidx = 0; sidx < LC_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      res[sidx] = d4_t{0, 0, 0, 0};
      if (t < nb * nb) mma_tile<true>(res[sidx], MA + ((t / nb) * 16) * ldp, ldp, 1, PC + (t % nb) * 16, ldp, 1, np, lane);
    }
    LEG_BARRIER();
#pragma unroll
    for (int sidx = 0; sidx < LC_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nb * nb) {
        const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; MA[row * ldp + col] = res[sidx][q] + (row == col ? 1.0 : 0.0); }  // pad rows: identity
      }
    }
    LEG_BARRIER();
    LEG_PROF(24);
    // Gauss-Jordan on T = [Mt | R | rv] (n rows, 2 n + 1 columns) with partial pivoting, the whole tableau in REGISTERS (through LDS
    // the elimination is bound by LDS bandwidth: every column rewrites the tableau).  A lane is a ROW (lane l: rows l and l + 64), a
    // wavefront owns every 8th COLUMN (wavefront w: columns w, w + 8, ...: GJ_SLOTS of them).  Per pivot column: its owner finds the
    // pivot with a DPP max + ballot over its own lanes, scales the column to the elimination factors and leaves them (one per row)
    // and the pivot row index in LDS; ONE barrier; every wavefront fetches its factor(s) and broadcasts the pivot row's entries of its
    // own columns with v_readlane (lane p), rank-one update in registers.  Implicit row permutation, the pivot row is not scaled: at
    // the end unknown `col` sits in row perm[col], scaled by dinv[col].  Dead columns keep round-off residues: never read again.
    // (An earlier layout — wavefronts own rows, lanes own columns — needed the column entries of ten rows per wavefront, a search
    // through LDS and two barriers per column: 165 us per cut against this one's, see DESIGN.md.)
    constexpr int GJ_SLOTS = (2 * NP + 1 + 7) / 8;
    double* dinv = fcol;        // 1 / pivot of every column
    int* iperm = used;          // iperm[row] = unknown that row holds
    double* fbuf = MA;          // [2][128] elimination factors of the current column, double-buffered ; MA and RB are dead while the
    int* pbuf = (int*)(MA + 256);  // tableau is in registers (first written after every wavefront has loaded its share: barrier below)
    double tq[2][GJ_SLOTS];
#pragma unroll
    for (int sl = 0; sl < GJ_SLOTS; ++sl) {
      const int cc = 8 * sl + wv;  // column of the tableau: [0, n) Mt, [n, 2 n) R, 2 n: rv
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + 64 * h;
        double v = 0.0;
        if (r < NP) v = (cc < n) ? MA[r * ldp + cc] : ((cc < 2 * n) ? RB[r * ldp + (cc - n)] : ((cc == 2 * n) ? rv[r] : 0.0));
        tq[h][sl] = v;
      }
    }
    bool used0 = false, used1 = false;  // rows lane, lane + 64 already served as pivots (every wavefront keeps the same flags)
    LEG_BARRIER();
#pragma unroll
    for (int so = 0; so < NP / 8; ++so) {
      for (int ow = 0; ow < 8; ++ow) {  // nw == 8
        const int col = 8 * so + ow;
        if (col >= n) break;
        double* fb = fbuf + (col & 1) * 128;
        if (wv == ow) {
          const double e0 = tq[0][so], e1 = tq[1][so];
          const double v0 = (lane < n && !used0) ? fabs(e0) : -1.0, v1 = (lane + 64 < n && !used1) ? fabs(e1) : -1.0;
          const bool second = v1 > v0;
          const double vl = second ? v1 : v0;
          const double vmax = wave_max_nonneg(fmax(vl, 0.0));
          const unsigned long long mk = __ballot(vl == vmax);
          const int src = __builtin_amdgcn_readfirstlane(mk ? __ffsll((long long)mk) - 1 : 0);
          const int ph = __builtin_amdgcn_readlane(second ? 1 : 0, src);
          const int p = src + 64 * ph;
          const double piv = readlane_dyn(ph ? e1 : e0, src);
          double inv = __builtin_amdgcn_rcp(piv);  // v_rcp_f64 + two Newton steps: the IEEE division is ~40 dependent instructions on the critical path of every column
          inv = inv * (2.0 - piv * inv);
          inv = inv * (2.0 - piv * inv);
          fb[lane] = (lane == p || lane >= n) ? 0.0 : e0 * inv;
          fb[lane + 64] = (lane + 64 == p || lane + 64 >= n) ? 0.0 : e1 * inv;
          if (lane == 0) { pbuf[col & 1] = p; perm[col] = p; iperm[p] = col; dinv[col] = inv; }
        }
        LEG_BARRIER();
        const int p = __builtin_amdgcn_readfirstlane(pbuf[col & 1]);
        const double f0 = fb[lane], f1 = fb[lane + 64];
        if (lane == (p & 63)) { if (p >> 6) used1 = true; else used0 = true; }
        // slots below `so` hold columns of Mt that are dead already (all their columns are < col): skipped
        if (p < 64) {
#pragma unroll
          for (int sl = so; sl < GJ_SLOTS; ++sl) { const double pr = readlane_dyn(tq[0][sl], p); tq[0][sl] -= f0 * pr; tq[1][sl] -= f1 * pr; }
        } else {
#pragma unroll
          for (int sl = so; sl < GJ_SLOTS; ++sl) { const double pr = readlane_dyn(tq[1][sl], p - 64); tq[0][sl] -= f0 * pr; tq[1][sl] -= f1 * pr; }
        }
      }
    }
    LEG_BARRIER();
    // solution in natural order back into RB (rows of the unknowns) and rv
#pragma unroll
    for (int sl = 0; sl < GJ_SLOTS; ++sl) {
      const int cc = 8 * sl + wv;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + 64 * h;
        if (r < n && cc >= n && cc <= 2 * n) {
          const int u = iperm[r];
          const double v = tq[h][sl] * dinv[u];
          if (cc < 2 * n) RB[u * ldp + (cc - n)] = v; else rv[u] = v;
        }
      }
    }
    LEG_BARRIER();
    LEG_PROF(25);
    // Zx, zc out (the forward part and the exact K_0 read them) ; padding of RB / rv stays zero
    for (int i = wv; i < n; i += nw) for (int c0 = lane; c0 < n; c0 += 64) lr[L.lZx + i * n + c0] = RB[i * ldp + c0];
    for (int i = tid; i < n; i += nthr) lr[L.lzc + i] = rv[i];
    LEG_PROF(26);
    // D = calP Zx (to registers, then into MA) ; ev = calP zc + calp
#pragma unroll
    for (int sidx = 0; sidx < LC_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      res[sidx] = d4_t{0, 0, 0, 0};
      if (t < nb * nb) mma_tile<false>(res[sidx], PC + ((t / nb) * 16) * ldp, ldp, 1, RB + (t % nb) * 16, ldp, 1, np, lane);
    }
    for (int i = wv; i < np; i += nw) {
      double s = 0;
      for (int c0 = lane; c0 < n; c0 += 64) s += PC[i * ldp + c0] * rv[c0];
      s = wave_sum(s);
      if (lane == 0) ev[i] = (i < n) ? s + pc[i] : 0.0;
    }
    LEG_BARRIER();
#pragma unroll
    for (int sidx = 0; sidx < LC_TILES; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nb * nb) tile_store(MA + ((t / nb) * 16) * ldp + (t % nb) * 16, ldp, res[sidx], lane);
    }
    LEG_PROF(27);
    if (j > 0) {
      // calP_j = P_j + Lm_j D ; calp_j = p_j + Lm_j ev   (RB <- Lm_j)
      leg_load_mat<false>(RB, ldp, np, gs + L.oLm, n, tid, nthr, S.mg_np);
      LEG_BARRIER();
      const int nst = nb * (nb + 1) / 2;
      d4_t pres[LC_STILES];
#pragma unroll
      for (int sidx = 0; sidx < LC_STILES; ++sidx) {
        const int t = wv + sidx * nw;
        pres[sidx] = d4_t{0, 0, 0, 0};
        if (t < nst) {
          int ri = 0, rem = t;
          while (rem > ri) { rem -= ri + 1; ++ri; }
          const int col = rem * 16 + (lane & 15);
#pragma unroll
          for (int q = 0; q < 4; ++q) {  // P_j tile: in flight while the matrix cores work
            const int row = ri * 16 + (lane >> 4) + 4 * q;
            pres[sidx][q] = (row < n && col < n) ? gs[L.oP + row * n + col] : 0.0;
          }
          mma_tile<false>(pres[sidx], RB + (ri * 16) * ldp, ldp, 1, MA + rem * 16, ldp, 1, np, lane);
        }
      }
      for (int i = wv; i < np; i += nw) {
        double s = 0;
        for (int c0 = lane; c0 < n; c0 += 64) s += RB[i * ldp + c0] * ev[c0];
        s = wave_sum(s);
        if (lane == 0) pc[i] = (i < n) ? s + gs[L.op + i] : 0.0;
      }
      LEG_BARRIER();
      // lower block triangle + P_j, mirrored; diagonal tiles symmetrised (0.5 (a + a^T)) through LDS
#pragma unroll
      for (int sidx = 0; sidx < LC_STILES; ++sidx) {
        const int t = wv + sidx * nw;
        if (t < nst) {
          int ri = 0, rem = t;
          while (rem > ri) { rem -= ri + 1; ++ri; }
          const int col = rem * 16 + (lane & 15);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int row = ri * 16 + (lane >> 4) + 4 * q;
            const double v = (row < n && col < n) ? pres[sidx][q] : 0.0;
            PC[row * ldp + col] = v;
            if (ri != rem) PC[col * ldp + row] = v;
          }
          if (ri == rem) {
            double tv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; tv[q] = 0.5 * (PC[row * ldp + col] + PC[col * ldp + row]); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; PC[row * ldp + col] = tv[q]; }
          }
        }
      }
      LEG_BARRIER();
    } else {
      // exact K_0 = K_0 + Kth_0 D_0 with Kth_0 = Ku_0 Lm_1 (Lm_1 = I when leg 0 is a single knot) ; MA = D_0
      double* g0 = gain_ptr(a, b, 0);
      const bool single = leg_start(a, 1) == 1;
      const double* g1 = gain_ptr(a, b, 1);
      LEG_BARRIER();
      for (int idx = tid; idx < mp * ldp; idx += nthr) { const int i = qdiv(idx, S.mg_ldp), c0 = (idx - qdiv(idx, S.mg_ldp) * ldp); RB[idx] = (i < L.m && c0 < n) ? g0[L.oKu + i * n + c0] : 0.0; }
      if (single) { for (int idx = tid; idx < np * ldp; idx += nthr) { const int i = qdiv(idx, S.mg_ldp), c0 = (idx - qdiv(idx, S.mg_ldp) * ldp); PC[idx] = (i == c0 && i < n) ? 1.0 : 0.0; } }
      else leg_load_mat<false>(PC, ldp, np, g1 + L.oLm, n, tid, nthr, S.mg_np);
      LEG_BARRIER();
      d4_t kres[2];
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int t = wv + sidx * nw;
        kres[sidx] = d4_t{0, 0, 0, 0};
        if (t < nbm * nb) mma_tile<false>(kres[sidx], RB + ((t / nb) * 16) * ldp, ldp, 1, PC + (t % nb) * 16, ldp, 1, np, lane);
      }
      LEG_BARRIER();
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int t = wv + sidx * nw;
        if (t < nbm * nb) tile_store(RB + ((t / nb) * 16) * ldp + (t % nb) * 16, ldp, kres[sidx], lane);
      }
      LEG_BARRIER();
#pragma unroll
      for (int sidx = 0; sidx < 2; ++sidx) {
        const int t = wv + sidx * nw;
        if (t < nbm * nb) {
          d4_t acc = d4_t{0, 0, 0, 0};
          mma_tile<false>(acc, RB + ((t / nb) * 16) * ldp, ldp, 1, MA + (t % nb) * 16, ldp, 1, np, lane);
          const int col = (t % nb) * 16 + (lane & 15);
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = (t / nb) * 16 + (lane >> 4) + 4 * q; if (row < L.m && col < n) g0[L.oK + row * n + col] += acc[q]; }
        }
      }
      LEG_BARRIER();
    }
  }
  __syncthreads();  // full barrier: Zx, zc, dP, calp of every cut were written to global memory by other threads
  LEG_PROF(28);
  // forward over the cuts: x_{j+1} = Zx_j x_j + zc_j (x_0 = 0: forced initial condition) ; theta_{j+1} = calP_{j+1} x_{j+1} + calp_{j+1}
  for (int i = tid; i < np; i += nthr) xv[i] = 0.0;
  LEG_BARRIER();
  constexpr int FW_ROWS = NP / 8;  // rows of a mat-vec per wavefront: all their loads in flight before the reductions
  for (int j = 0; j + 1 < J; ++j) {
    double* lr = leg_ptr(a, b, j);
    const int cut = leg_start(a, j + 1);
    const int c0 = lane < n ? lane : 0, c1 = lane + 64 < n ? lane + 64 : 0;
    const double m0 = lane < n ? 1.0 : 0.0, m1 = lane + 64 < n ? 1.0 : 0.0;
    double va[FW_ROWS][2];
    if (j > 0) {  // x_0 = 0: the first cut state is zc_0
#pragma unroll
      for (int i = 0; i < FW_ROWS; ++i) { const int r = wv + i * nw, rr = r < n ? r : 0; va[i][0] = lr[L.lZx + rr * n + c0] * m0; va[i][1] = lr[L.lZx + rr * n + c1] * m1; }
    }
#pragma unroll
    for (int i = 0; i < FW_ROWS; ++i) {
      const int r = wv + i * nw;
      double sx = 0.0;
      if (j > 0) sx = wave_sum(va[i][0] * xv[c0] + va[i][1] * xv[c1]);
      if (lane == 0 && r < n) { ev[r] = sx + lr[L.lzc + r]; a.dxs[((size_t)b * (N + 1) + cut) * n + r] = ev[r]; }
    }
#pragma unroll
    for (int i = 0; i < FW_ROWS; ++i) { const int r = wv + i * nw, rr = r < n ? r : 0; va[i][0] = lr[L.ldP + rr * n + c0] * m0; va[i][1] = lr[L.ldP + rr * n + c1] * m1; }
    LEG_BARRIER();
#pragma unroll
    for (int i = 0; i < FW_ROWS; ++i) {
      const int r = wv + i * nw;
      const double sx = wave_sum(va[i][0] * ev[c0] + va[i][1] * ev[c1]);
      if (lane == 0 && r < n) lr[L.lth + r] = sx + lr[L.lcp + r];
    }
    for (int i = tid; i < n; i += nthr) xv[i] = ev[i];
    LEG_BARRIER();
  }
  LEG_PROF(29);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_leg_apply: grid (N, B), block 256.  Knot k of a parametric leg with end co-state theta: p' = Lm_{k+1} theta (theta itself at
// the last knot of the leg) ; p += Lm_k theta ; k += Ku p' ; knu[active] += Knup p' ; phi += Gamma p'.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_leg_apply(SolverArgs a) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int j = leg_of_knot(a, k);
  if (j + 1 >= a.nlegs) return;
  const int n = L.n, ke = leg_start(a, j + 1) - 1;
  const double* kn = knot_ptr(a, b, k);
  double* g = gain_ptr(a, b, k);
  const int m = (int)kn[L.oMISC + MISC_M], c = (int)kn[L.oMISC + MISC_NC];
  const double* th_g = leg_ptr(a, b, j) + L.lth;
  __shared__ double th[128], pn[128];
  __shared__ int act_idx[256];
  __shared__ int wcnt[4];
  for (int i = tid; i < n; i += nthr) th[i] = th_g[i];
  // active rows in order (ballot prefix; c <= 256 = block size, checked by the host)
  const bool is_act = tid < c && kn[L.oACT + (tid < c ? tid : 0)] != 0.0;
  const unsigned long long amask = __ballot(is_act);
  if (lane == 0) wcnt[wv] = __popcll(amask);
  __syncthreads();
  int ca = 0;
  {
    int off = 0;
    for (int q = 0; q < nw; ++q) { if (q < wv) off += wcnt[q]; ca += wcnt[q]; }
    if (is_act) act_idx[off + __popcll(amask & ((1ull << lane) - 1ull))] = tid;
  }
  // mat-vecs: a wavefront takes four rows at a time, all their loads in flight before the four reductions
  const int c0 = lane < n ? lane : 0, c1 = lane + 64 < n ? lane + 64 : 0;
  const double m0 = lane < n ? 1.0 : 0.0, m1 = lane + 64 < n ? 1.0 : 0.0;
  if (k == ke) { for (int i = tid; i < n; i += nthr) pn[i] = th[i]; }
  else {
    const double* gn = gain_ptr(a, b, k + 1) + L.oLm;
    const double t0 = th[c0] * m0, t1 = th[c1] * m1;
    for (int r0 = 4 * wv; r0 < n; r0 += 4 * nw) {
      double v[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int r = r0 + q < n ? r0 + q : 0; v[q][0] = gn[r * n + c0]; v[q][1] = gn[r * n + c1]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const double sx = wave_sum(v[q][0] * t0 + v[q][1] * t1); if (lane == 0 && r0 + q < n) pn[r0 + q] = sx; }
    }
  }
  __syncthreads();
  // rows: [0, n) p += Lm_k theta ; [n, 2n) phi += Gamma p' ; [2n, 2n + m) k += Ku p' ; then the active constraint rows: knu += Knup p'
  const double t0 = th[c0] * m0, t1 = th[c1] * m1, q0 = pn[c0] * m0, q1 = pn[c1] * m1;
  const int total = 2 * n + m + ca;
  for (int r0 = 4 * wv; r0 < total; r0 += 4 * nw) {
    double v[4][2];
    double* dst[4];
    bool use_th[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q < total ? r0 + q : 0;
      const double* row;
      if (r < n) { row = g + L.oLm + r * n; dst[q] = g + L.op + r; use_th[q] = true; }
      else if (r < 2 * n) { row = g + L.oGam + (r - n) * n; dst[q] = g + L.ophi + (r - n); use_th[q] = false; }
      else if (r < 2 * n + m) { row = g + L.oKu + (r - 2 * n) * n; dst[q] = g + L.ok + (r - 2 * n); use_th[q] = false; }
      else { row = g + L.oKnup + (r - 2 * n - m) * n; dst[q] = g + L.oknu + act_idx[r - 2 * n - m]; use_th[q] = false; }
      v[q][0] = row[c0]; v[q][1] = row[c1];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double sx = wave_sum(use_th[q] ? v[q][0] * t0 + v[q][1] * t1 : v[q][0] * q0 + v[q][1] * q1);
      if (lane == 0 && r0 + q < total) *dst[q] += sx;
    }
  }
}
