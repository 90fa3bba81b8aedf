// legs_tree.h — a TREE over the cuts of the parallel-in-time sweep instead of the chain of k_leg_consensus (legs.h).
// The chain solves one n x n system per cut, one after the other: with J legs the sweep takes N / J knots and the consensus J - 1
// solves, which is what bounds the latency of a single instance (batch 1: 8 legs, 0.9 ms of sweep + 0.9 ms of consensus).  Here the
// condensed forms of adjacent (groups of) legs are composed pairwise, all pairs of a level side by side on their own CUs:
// ceil(log2 J) rounds.  A node covering legs lo..hi is, like a leg,
//     lambda_in = P x_in + Lm theta_out + p ,   x_out = Lm^T x_in + Sg theta_out + sg      (theta_out: co-state parameter at its end)
// and a followed by b, with D = P_b - Pg (Pg = the terminal Hessian the last leg of a carried), W = (I - Sg_a D)^-1:
//     T1 = W Lm_a^T, T2 = W Sg_a, t3 = W (Sg_a p_b + sg_a)                     (one Gauss-Jordan with 2 n + 1 right-hand sides)
//     Zt = T2 Lm_b ;  Lm_ab = T1^T Lm_b ;  Sg_ab = Sg_b + Lm_b^T Zt ;  sg_ab = sg_b + Lm_b^T t3
//     F = D T1 ;  E = D Zt + Lm_b ;  u = D t3 + p_b ;  P_ab = P_a + Lm_a F ;  p_ab = p_a + Lm_a u
//     x_mid = T1 x_in + Zt theta_out + t3 ;  theta_mid = F x_in + E theta_out + u          (down-sweep: k_leg_tree_down)
// Same KKT system as the chain and the serial sweep: identical steps up to round-off (oracle/solver.hpp backward_legs_tree,
// tests/test_oracle_legs.py, tests/test_gpu_legs.py).  The guess of the value-function Hessian at a cut (kept in the leg record for the
// next pass) is the Hessian of the node that STARTS there, given its own end guess: exact for the nodes that hold the last leg, the
// others catch up one level per pass (a handle's first pass sweeps depth + 1 times).  The exact feedback gain of knot 0 follows the
// leftmost path of the tree: d theta / d x_0 = F + E (d theta_out / d x_0).
#pragma once
#include "legs.h"

#define MPC_TREE_MAX_NODES (2 * MPC_MAX_LEGS - 1)
#define MPC_TREE_MAX_LEVELS 8
struct TreeDesc {
  int J, nnodes, nlev;
  int lo[MPC_TREE_MAX_NODES], hi[MPC_TREE_MAX_NODES], left[MPC_TREE_MAX_NODES], right[MPC_TREE_MAX_NODES];
  int lev_first[MPC_TREE_MAX_LEVELS], lev_cnt[MPC_TREE_MAX_LEVELS];  // inner nodes created by level l: lev_first[l] .. + lev_cnt[l]
};
// leaves 0 .. J-1 (the legs), then the inner nodes level by level: adjacent nodes paired, an odd one carried up (oracle: backward_legs_tree)
static inline TreeDesc make_tree_desc(int J) {
  TreeDesc T;
  T.J = J; T.nnodes = J; T.nlev = 0;
  int level[MPC_MAX_LEGS], cnt = J;
  for (int j = 0; j < J; ++j) { T.lo[j] = T.hi[j] = j; T.left[j] = T.right[j] = -1; level[j] = j; }
  while (cnt > 1) {
    int next[MPC_MAX_LEGS], nn = 0;
    T.lev_first[T.nlev] = T.nnodes;
    for (int i = 0; i + 1 < cnt; i += 2) {
      const int id = T.nnodes++;
      T.left[id] = level[i]; T.right[id] = level[i + 1]; T.lo[id] = T.lo[level[i]]; T.hi[id] = T.hi[level[i + 1]];
      next[nn++] = id;
    }
    T.lev_cnt[T.nlev] = T.nnodes - T.lev_first[T.nlev];
    ++T.nlev;
    if (cnt % 2) next[nn++] = level[cnt - 1];
    for (int i = 0; i < nn; ++i) level[i] = next[i];
    cnt = nn;
  }
  return T;
}

DEV double* tree_node_ptr(const SolverArgs& a, int b, int inner) { return a.treebuf + ((size_t)b * a.leg_cap + inner) * a.L.tree_stride; }
DEV double* tree_scratch_ptr(const SolverArgs& a, int b) { return tree_node_ptr(a, b, a.leg_cap - 1); }  // G of the K_0 path (mp x n)

// Exact K_0 = K_0 + Ku_0 Lm_1 d theta_1 / d x_0, d theta_1 / d x_0 = S_1, S_i = F_i + E_i S_{i+1} along the leftmost path (node 1: the
// parent of leg 0 ... the root), evaluated from the left with the thin matrix G (m x n):  G_1 = Ku_0 Lm_1 ;  K_0 += G_i F_i ;
// G_{i+1} = G_i E_i.  One step per launch of the up-sweep, by a workgroup of its own beside the compositions (node i is complete when
// level i + 1 starts), the root's step beside the first level of the down-sweep: nothing of it is on the critical path.
// node < 0: G_1 only.
template <int NP, int FN = 0, int FM = 0>
DEV void tree_k0_step(const SolverArgs& a, const LxLds& S, const TreeDesc& T, int b, int node, double* sm, int tid, int nthr) {
  Layout L_ = a.L;
  if constexpr (FN > 0) { L_.n = FN; L_.m = FM; L_.nz = FN + FM; }
  const Layout& L = L_;
  const int n = L.n, np = S.np, mp = S.mp, ldp = S.ldp, nb = S.nb, nbm = S.nbm, nw = nthr >> 6, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *PC = sm + S.PC, *MA = sm + S.MA, *RB = sm + S.RB;
  double* G = tree_scratch_ptr(a, b);
  double* g0 = gain_ptr(a, b, 0);
  d4_t r0[2], r1[2];
  if (node < 0) {  // G_1 = Ku_0 Lm_1 (Lm_1 = I when leg 0 is a single knot)
    const bool single = leg_start(a, 1) == 1;
    const double* g1 = gain_ptr(a, b, 1);
    for (int idx = tid; idx < mp * ldp; idx += nthr) { const int i = qdiv(idx, S.mg_ldp), cc = idx - i * ldp; RB[idx] = (i < L.m && cc < n) ? g0[L.oKu + i * n + cc] : 0.0; }
    if (single) { for (int idx = tid; idx < np * ldp; idx += nthr) { const int i = qdiv(idx, S.mg_ldp), cc = idx - i * ldp; MA[idx] = (i == cc && i < n) ? 1.0 : 0.0; } }
    else leg_load_mat<false>(MA, ldp, np, g1 + L.oLm, n, tid, nthr, S.mg_np);
    LEG_BARRIER();
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
      const int tt = wv + sidx * nw;
      if (tt < nbm * nb) {
        d4_t acc = d4_t{0, 0, 0, 0};
        mma_tile<false>(acc, RB + ((tt / nb) * 16) * ldp, ldp, 1, MA + (tt % nb) * 16, ldp, 1, np, lane);
        const int col = (tt % nb) * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = (tt / nb) * 16 + (lane >> 4) + 4 * q; if (row < L.m && col < n) G[row * n + col] = acc[q]; }
      }
    }
    return;
  }
  const double* t = tree_node_ptr(a, b, node - T.J);
  const bool has_t = T.hi[node] + 1 < T.J;
  for (int idx = tid; idx < mp * ldp; idx += nthr) { const int i = qdiv(idx, S.mg_ldp), cc = idx - i * ldp; RB[idx] = (i < L.m && cc < n) ? G[i * n + cc] : 0.0; }
  leg_load_mat<false>(PC, ldp, np, t + L.tF, n, tid, nthr, S.mg_np);
  if (has_t) leg_load_mat<false>(MA, ldp, np, t + L.tE, n, tid, nthr, S.mg_np);
  LEG_BARRIER();
#pragma unroll
  for (int sidx = 0; sidx < 2; ++sidx) {
    const int tt = wv + sidx * nw;
    r0[sidx] = d4_t{0, 0, 0, 0}; r1[sidx] = d4_t{0, 0, 0, 0};
    if (tt < nbm * nb) {
      mma_tile<false>(r0[sidx], RB + ((tt / nb) * 16) * ldp, ldp, 1, PC + (tt % nb) * 16, ldp, 1, np, lane);
      if (has_t) mma_tile<false>(r1[sidx], RB + ((tt / nb) * 16) * ldp, ldp, 1, MA + (tt % nb) * 16, ldp, 1, np, lane);
    }
  }
#pragma unroll
  for (int sidx = 0; sidx < 2; ++sidx) {
    const int tt = wv + sidx * nw;
    if (tt < nbm * nb) {
      const int col = (tt % nb) * 16 + (lane & 15);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (tt / nb) * 16 + (lane >> 4) + 4 * q;
        if (row < L.m && col < n) { g0[L.oK + row * n + col] += r0[sidx][q]; if (has_t) G[row * n + col] = r1[sidx][q]; }  // (G was read into LDS by everybody: barrier above)
      }
    }
  }
}
#ifndef MPC_TREE_GROWTH
#define MPC_TREE_GROWTH 1e6  // bound on the multipliers of the blocked elimination of k_leg_compose (beyond it: the pivoted Gauss-Jordan)
#endif
// condensed form of a node; Lm == nullptr: the node holds the last leg (no end parameter: Lm = Sg = 0, sg = 0)
struct NodeRef { const double *P, *p, *Lm, *Sg, *sg; };
DEV NodeRef tree_node_ref(const SolverArgs& a, const TreeDesc& T, int b, int node) {
  const Layout& L = a.L;
  NodeRef r;
  const bool last = T.hi[node] + 1 == T.J;
  if (node < T.J) {
    const double* g = gain_ptr(a, b, leg_start(a, node));
    r.P = g + L.oP; r.p = g + L.op;
    r.Lm = last ? nullptr : g + L.oLm;
    const double* lr = last ? nullptr : leg_ptr(a, b, node);
    r.Sg = last ? nullptr : lr + L.lSg; r.sg = last ? nullptr : lr + L.lsg;
  } else {
    const double* t = tree_node_ptr(a, b, node - T.J);
    r.P = t + L.tP; r.p = t + L.tp;
    r.Lm = last ? nullptr : t + L.tLm; r.Sg = last ? nullptr : t + L.tSg; r.sg = last ? nullptr : t + L.tsg;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_leg_compose: grid (nodes of the level + 1, B, 2) — the extra workgroup (z = 0) takes a step of the K_0 path.  LDS: the three n x n buffers of the consensus kernel (X, Y, Z below).
// The composition splits into two halves that share nothing but their inputs, one workgroup each (blockIdx.z):
//   role 0: Gauss-Jordan on [Mt | Lm_a^T | rv] -> T1, t3 ;  Zx, zc, Lm_ab, sg_ab, F, u, P_ab, p_ab
//   role 1: Gauss-Jordan on [Mt | Sg_a]        -> T2     ;  Zt, Sg_ab, E            (nothing to do when b holds the last leg)
// (both eliminate the same Mt: the elimination is two thirds of a composition, and its cost is the number of columns a wavefront owns)
// ---------------------------------------------------------------------------------------------------------------------
#ifndef LCMP_THREADS
#define LCMP_THREADS 512  // 8 wavefronts (measured with 1024 — every 16th column, 11 slots per wavefront, 128 VGPRs: 128 B of spills, dearer barriers: 0.502 against 0.507 ms for four levels, the centroidal problem slower)
#endif
template <int NP, int FN = 0, int FM = 0>  // FN, FM > 0: state / control dimensions as compile-time constants (see k_riccati_mfma)
__global__ void __launch_bounds__(LCMP_THREADS) k_leg_compose(SolverArgs a, LxLds Srt, TreeDesc T, int level) {
  constexpr int NWC = LCMP_THREADS / 64, LCT = (25 + NWC - 1) / NWC, LCS = (15 + NWC - 1) / NWC;  // wavefronts ; tiles / lower-triangle tiles per wavefront (nb <= 5)
  constexpr bool FX = FN > 0;
  constexpr LxLds SC_ = FX ? make_lx_lds(FN, FM) : LxLds{};
  LxLds S_ = Srt;
  if constexpr (FX) S_ = SC_;
  const LxLds& S = S_;
  Layout L_ = a.L;
  if constexpr (FX) { L_.n = FN; L_.m = FM; L_.nz = FN + FM; }
  const Layout& L = L_;
  const int b = blockIdx.y;
  constexpr int nthr = LCMP_THREADS, nw = LCMP_THREADS / 64;
  int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, np = S.np, ldp = S.ldp, nb = S.nb;
  if ((int)blockIdx.x == T.lev_cnt[level]) {  // the workgroup of the K_0 path (tree_k0_step): the leftmost node of the level below
    extern __shared__ __attribute__((aligned(16))) double smk[];
    if (blockIdx.z == 0) tree_k0_step<NP, FN, FM>(a, S, T, b, level == 0 ? -1 : T.lev_first[level - 1], smk, tid, nthr);
    return;
  }
  const int node = T.lev_first[level] + blockIdx.x;
  const NodeRef A = tree_node_ref(a, T, b, T.left[node]), Bn = tree_node_ref(a, T, b, T.right[node]);
  double* out = tree_node_ptr(a, b, node - T.J);
  double* lrc = leg_ptr(a, b, T.hi[T.left[node]]);  // the leg that ends at the cut: its record keeps the guess (lcP) and D (ldP)
  const bool bpar = Bn.Lm != nullptr;              // b has an end parameter (it does not hold the last leg)
  const int role = blockIdx.z;
  if (role == 1 && !bpar) {  // Zt = 0, Sg_ab = 0, E = 0: never read (the node has no end parameter either)
    return;
  }
  extern __shared__ __attribute__((aligned(16))) double sm[];
  // developer phase timers (mpc_profile(3)): role 0 of the first composition of level 0, instance 0 -> slots 24 .. 28 of its counter block
  long long tc0_ = clock64();
#define LCMP_PROF(slot) do { if (a.prof && b == 0 && level == 0 && blockIdx.x == 0 && role == 0 && tid == 0) { const long long t1_ = clock64(); a.prof[(slot)] += (double)(t1_ - tc0_); tc0_ = t1_; } } while (0)
  double *X = sm + S.PC, *Y = sm + S.MA, *Z = sm + S.RB, *vec = sm + S.vec;
  double *pb = vec, *rv = vec + np, *uu = vec + 2 * np, *fcol = vec + 4 * np;  // p_b | right-hand side / t3 | u | 1 / pivots
  int* perm = (int*)(sm + S.iw);
  int* used = perm + np;
  d4_t res[LCT];
  // ---- X <- Sg_a ; Y <- D = P_b - Pg (Pg: lcP of the leg that ends at the cut, read by both roles) ; ldP <- D ----
  double dreg[LK_PT];  // D, kept in registers by role 0 for the products after the elimination (its buffer is reused in between)
  {
    double pv[LK_PT], po[LK_PT];
#pragma unroll
    for (int u = 0; u < LK_PT; ++u) {
      const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), c0 = idx - i * np;
      const bool ok = idx < np * np && i < n && c0 < n;
      pv[u] = Bn.P[ok ? i * n + c0 : 0] * (ok ? 1.0 : 0.0);
      const double pg = lrc[L.lcP + (ok ? i * n + c0 : 0)];
      po[u] = (ok && a.leg_guess) ? pg : 0.0;
    }
    leg_load_mat<false>(X, ldp, np, A.Sg, n, tid, nthr, S.mg_np);
#pragma unroll
    for (int u = 0; u < LK_PT; ++u) {
      const int idx = tid + u * nthr, i = qdiv(idx, S.mg_np), c0 = idx - i * np;
      if (idx < np * np) {
        const double d = pv[u] - po[u];
        dreg[u] = d;
        Y[i * ldp + c0] = d;
        if (role == 0 && i < n && c0 < n) lrc[L.ldP + i * n + c0] = d;  // (the guess itself is refreshed by k_leg_tree_down: the other role reads it too)
      }
    }
  }
  for (int i = tid; i < np; i += nthr) pb[i] = (i < n) ? Bn.p[i] : 0.0;
  LEG_BARRIER();
  LCMP_PROF(24);
  // ---- rv = Sg_a p_b + sg_a ; Mt = I - Sg_a D (to registers, then into Z) ----
  for (int i = wv; i < np; i += nw) {
    double s = 0;
    for (int c0 = lane; c0 < n; c0 += 64) s += X[i * ldp + c0] * pb[c0];
    s = wave_sum(s);
    if (lane == 0) rv[i] = (i < n) ? s + A.sg[i] : 0.0;
  }
  auto build_mt = [&]() {  // Z <- I - Sg_a D  (Sg_a in X, D in Y)
#pragma unroll
    for (int sidx = 0; sidx < LCT; ++sidx) {
      const int t = wv + sidx * nw;
      res[sidx] = d4_t{0, 0, 0, 0};
      if (t < nb * nb) mma_tile<true>(res[sidx], X + ((t / nb) * 16) * ldp, ldp, 1, Y + (t % nb) * 16, ldp, 1, np, lane);
    }
#pragma unroll
    for (int sidx = 0; sidx < LCT; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nb * nb) {
        const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; Z[row * ldp + col] = res[sidx][q] + (row == col ? 1.0 : 0.0); }  // pad rows: identity
      }
    }
  };
  build_mt();
  LEG_BARRIER();  // role 0: D in Y is dead (it reads D back from the leg record later) ; role 1 keeps D in Y
  LEG_LAUNDER();
  if (role == 0) leg_load_mat<true>(Y, ldp, np, A.Lm, n, tid, nthr, S.mg_np);  // Y <- Lm_a^T
  LEG_BARRIER();
  LCMP_PROF(25);
  // ---- Elimination on the matrix cores first (round 4): Gauss-Jordan on [Mt | R] by PANELS OF FOUR COLUMNS with the pivots on the diagonal.
  // The tableau lives in accumulator registers as 16 x 16 tiles, a wavefront owns whole column tiles (<= 2: 5 tiles each), so the pivot rows
  // of a column tile and its update are in ONE wavefront: per panel the owner of the pivot column tile publishes the panel's four columns
  // (np x 4) and the inverse of its 4 x 4 pivot block through LDS, then every column tile gets  U = Pinv * (its pivot rows)  (one MFMA, the
  // result lands in the lanes that supply it as the B operand next) and  T -= Panel * U  (one MFMA per tile), the pivot rows become U.
  // 19 panels of ~2 500 cycles against 38 two-column steps of ~4 600 of the pivoted form below (74 us per level, profiles/r04_legs_phase_timers.txt).
  // Pivots are NOT searched for outside the 4 x 4 block: a block whose multipliers pass MPC_TREE_GROWTH abandons the attempt — nothing
  // has been written, Mt / R / rv are as they were — and the pivoted form below does the job.  (rv rides in column n of R: needs n < np.)
  // Lm_b, the operand of the products after the elimination, is requested from HBM now (13 doubles per thread) and dropped into Z when the
  // elimination is done: its latency was in the open
  double lmb[LK_PT];
  if (bpar) leg_request_mat(lmb, np, Bn.Lm, n, tid, nthr, S.mg_np);
  bool eliminated = false;
  if ((n & 3) == 0 && n < np && 8 * np + 96 <= np * ldp && !a.tree_pivoted) {
    constexpr int NBT = NP / 16, SL = (2 * NBT + NWC - 1) / NWC;  // row tiles ; column tiles per wavefront
    // scratch of the panels, double-buffered (ONE barrier per panel: the owner of the next panel publishes it while the others still update
    // with this one): a whole n x n buffer that is dead by now — X (Sg_a) for role 0 ; Z (Mt, in registers by then) for role 1, which
    // rebuilds Mt from Sg_a and D if the attempt is abandoned
    double* scr = role == 0 ? X : Z;
    int* failS = (int*)(vec + 5 * np);  // [2]: one flag per panel buffer, written (0 or 1) whenever the buffer is published
    const double* Rsrc = role == 0 ? Y : X;
    const int g = lane >> 4, c = lane & 15;
    d4_t col[SL][NBT];
    int ct[SL];
#pragma unroll
    for (int sl = 0; sl < SL; ++sl) {
      ct[sl] = (wv + nb) % NWC + sl * NWC;
      if (ct[sl] >= 2 * nb) ct[sl] = -1;
#pragma unroll
      for (int ri = 0; ri < NBT; ++ri) {
        col[sl][ri] = d4_t{0, 0, 0, 0};
        if (ct[sl] >= 0 && ri < nb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = ri * 16 + g + 4 * r;
            double v;
            if (ct[sl] < nb) v = Z[row * ldp + ct[sl] * 16 + c];
            else { const int cc = (ct[sl] - nb) * 16 + c; v = (role == 0 && cc == n) ? rv[row] : Rsrc[row * ldp + cc]; }
            col[sl][ri][r] = v;
          }
        }
      }
    }
    if (tid == 0) { failS[0] = 0; failS[1] = 0; }
    LEG_BARRIER();  // the tableau is in registers: the scratch buffer may be written
    // the owner of panel (rt, q) — the wavefront whose first slot is column tile rt — publishes the panel's four columns and the inverse of
    // its 4 x 4 pivot block into buffer `buf`
    auto publish = [&](int rt, int q, int buf) {
      const long long tp0_ = clock64();
      double* panel = scr + buf * (4 * np);     // [np][4]
      double* ppS = scr + 8 * np + buf * 40;   // [16] pivot block, [16] its inverse, [4] the terms of its determinant
      double pmax = 0.0;  // largest entry of the panel: with the largest entry of the inverse of the pivot block it bounds the multipliers
      if ((c >> 2) == q) {
#pragma unroll
        for (int ri = 0; ri < NBT; ++ri)
          if (ri < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) panel[(ri * 16 + g + 4 * r) * 4 + (c & 3)] = col[0][ri][r];
            pmax = fmax(pmax, fmax(fmax(fabs(col[0][ri][0]), fabs(col[0][ri][1])), fmax(fabs(col[0][ri][2]), fabs(col[0][ri][3]))));
          }
        // the pivot block: rows j0 + g (register q of tile (rt, rt)), columns j0 + (c & 3)
        double pv = 0.0;
#pragma unroll
        for (int ri = 0; ri < NBT; ++ri) if (ri == rt) pv = col[0][ri][q];
        ppS[g * 4 + (c & 3)] = pv;
      }
      pmax = wave_max_nonneg(pmax);
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wavefront's own LDS writes
      __builtin_amdgcn_wave_barrier();
      // inverse of the 4 x 4 pivot block, a lane per entry: lane 4 j + i forms the cofactor of element (j, i) — the 3 x 3 minor without row j
      // and column i — which over the determinant is entry (i, j) of the inverse; the determinant is the expansion along row 0 (lanes 0 .. 3)
      const int ii = lane & 3, jj = (lane >> 2) & 3;
      const int r0 = jj == 0 ? 1 : 0, r1 = jj <= 1 ? 2 : 1, r2 = jj <= 2 ? 3 : 2;
      const int c0 = ii == 0 ? 1 : 0, c1 = ii <= 1 ? 2 : 1, c2 = ii <= 2 ? 3 : 2;
      const double a00 = ppS[r0 * 4 + c0], a01 = ppS[r0 * 4 + c1], a02 = ppS[r0 * 4 + c2];
      const double a10 = ppS[r1 * 4 + c0], a11 = ppS[r1 * 4 + c1], a12 = ppS[r1 * 4 + c2];
      const double a20 = ppS[r2 * 4 + c0], a21 = ppS[r2 * 4 + c1], a22 = ppS[r2 * 4 + c2];
      const double pji = ppS[jj * 4 + ii];
      const double minor = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
      const double cof = ((ii + jj) & 1) ? -minor : minor;
      if (lane < 4) ppS[32 + lane] = pji * cof;  // (row 0: lanes 0 .. 3)
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      const double det = (ppS[32] + ppS[33]) + (ppS[34] + ppS[35]);
      const double pinv = cof / det;
      const double imax = wave_max_nonneg(lane < 16 ? fabs(pinv) : 0.0);
      // no pivot is looked for outside the 4 x 4 block: allowed while the multipliers Panel * Pinv stay below MPC_TREE_GROWTH (partial pivoting keeps
      // them below 1; six digits of the sixteen are what the bound gives away)
      const bool ok = isfinite(det) && det != 0.0 && isfinite(imax) && imax * pmax < MPC_TREE_GROWTH;
      if (lane < 16) ppS[16 + ii * 4 + jj] = pinv;
      if (lane == 0) failS[buf] = ok ? 0 : 1;
      if (a.prof && b == 0 && level == 0 && blockIdx.x == 0 && role == 0 && lane == 0) atomicAdd(&a.prof[28], (double)(clock64() - tp0_));  // (developer timer: the owner's serial piece)
    };
    // panel (rt, q) applied to column-tile slot sl of this wavefront
    auto update = [&](int sl, int rt, int q, int buf) {
      if (ct[sl] < 0 || (ct[sl] < nb && ct[sl] < rt)) return;  // no such tile / a column tile of Mt that is already eliminated
      const double* panel = scr + buf * (4 * np);
      const double* ppS = scr + 8 * np + buf * 40;
      const double aop = (c < 4) ? ppS[16 + c * 4 + g] : 0.0;  // A operand of U = Pinv * rows: A(i, k) = Pinv[i][k], i < 4
      double prow = 0.0;
#pragma unroll
      for (int ri = 0; ri < NBT; ++ri) if (ri == rt) prow = col[sl][ri][q];
      d4_t u = d4_t{0, 0, 0, 0};
      u = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, prow, u, 0, 0, 0);
      const double ub = u[0];  // U[g][column c of the tile]
#pragma unroll
      for (int ri = 0; ri < NBT; ++ri) {
        if (ri < nb) {
          double av = panel[(ri * 16 + c) * 4 + g];  // A(i = c, k = g) = Panel[row ri * 16 + i][k]
          if (ri == rt && (c >> 2) == q) av = 0.0;      // the pivot rows are not eliminated from themselves
          col[sl][ri] = __builtin_amdgcn_mfma_f64_16x16x4f64(-av, ub, col[sl][ri], 0, 0, 0);
          if (ri == rt) col[sl][ri][q] = ub;
        }
      }
    };
    const int npan = n >> 2;
    if (wv == (NWC - nb % NWC) % NWC) publish(0, 0, 0);  // (the owner of column tile 0)
    LEG_BARRIER();
    bool fail = failS[0] != 0;
    for (int rt = 0; rt * 16 < n && !fail; ++rt) {  // (the four panels of a column tile unrolled: the register index q of the pivot rows is then a constant)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pidx = rt * 4 + q;
        if (pidx >= npan || fail) continue;
        const int buf = pidx & 1;
        // the next panel and its owner
        const int qn = (q + 1) & 3;
        const int rtn = (q == 3) ? rt + 1 : rt;
        const bool has_next = pidx + 1 < npan;
        const bool own_next = has_next && wv == (rtn + NWC - nb % NWC) % NWC;
        update(0, rt, q, buf);                 // (the pivot column tile of the next panel is a first slot: updated first ...)
        if (own_next) publish(rtn, qn, buf ^ 1);  // (... and published while the other wavefronts are still updating)
#pragma unroll
        for (int sl = 1; sl < SL; ++sl) update(sl, rt, q, buf);
        LEG_BARRIER();
        // the flag of the panel about to be applied (buffer buf ^ 1, published before this barrier): the owner of the panel after it writes
        // the OTHER flag, so a wavefront that is slow to read cannot see a verdict that belongs to a later panel
        if (has_next && failS[buf ^ 1]) fail = true;
      }
    }
    if (a.prof && tid == 0) atomicAdd(&a.prof[(size_t)b * 64 + (fail ? 30 : 29)], 1.0);  // developer counters (mpc_profile(3)): compositions that took the blocked elimination / fell back to the pivoted one
    if (!fail) {
      eliminated = true;
      double* Rw = role == 0 ? Y : X;
#pragma unroll
      for (int sl = 0; sl < SL; ++sl) {
        if (ct[sl] < nb) continue;
#pragma unroll
        for (int ri = 0; ri < NBT; ++ri) {
          if (ri < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = ri * 16 + g + 4 * r, cc = (ct[sl] - nb) * 16 + c;
              const double v = col[sl][ri][r];
              if (role == 0 && cc == n) { rv[row] = (row < n) ? v : 0.0; Rw[row * ldp + cc] = 0.0; }
              else Rw[row * ldp + cc] = (row < n && cc < n) ? v : 0.0;
            }
          }
        }
      }
      LEG_BARRIER();
    } else if (role == 1) {  // the attempt used Z as its scratch: Mt again (role 0 used X, which nobody reads any more)
      build_mt();
      LEG_BARRIER();
    }
  }
  // ---- Gauss-Jordan on [Mt | R | rv] (n rows, 2 n + 1 columns; role 0: R = Lm_a^T in Y, role 1: R = Sg_a in X, no rv), tableau in
  // registers: as in k_leg_consensus (a lane is a row, a wavefront owns every 8th column, the owner of the pivot column leaves the
  // elimination factors in LDS, one barrier per column) ----
  if (!eliminated) {
    // (TWO columns per barrier: a wavefront owns the column pairs (2 (NWC t + wv), + 1) ; the owner of the pivot pair eliminates its first
    // column from its second one in registers, picks the second pivot from that, and leaves both factor columns in LDS — the other
    // wavefronts apply the two eliminations one after the other.  Same pivots, same operations in the same order as one column per
    // barrier: bit-identical results with half the barriers.)
    constexpr int GJ_SLOTS = 2 * ((2 * NP + 1 + 2 * NWC - 1) / (2 * NWC));
    double* dinv = fcol;
    int* iperm = used;
    double* fbuf = Z;                 // [2 pairs][2 columns][NP], double-buffered: Mt is dead once the tableau is in registers (Y keeps D for role 1)
    int* pbuf = (int*)(Z + 4 * NP);   // [2 pairs][2]   (Z holds np x (np + 1) doubles: 272 for np = 16)
    const double* R = role == 0 ? Y : X;
    double tq[2][GJ_SLOTS];
    auto col_of = [&](int sl) { return 2 * NWC * (sl >> 1) + 2 * wv + (sl & 1); };
#pragma unroll
    for (int sl = 0; sl < GJ_SLOTS; ++sl) {
      const int cc = col_of(sl);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + 64 * h;
        double v = 0.0;
        if (r < NP) v = (cc < n) ? Z[r * ldp + cc] : ((cc < 2 * n) ? R[r * ldp + (cc - n)] : ((cc == 2 * n && role == 0) ? rv[r] : 0.0));
        tq[h][sl] = v;
      }
    }
    bool used0 = false, used1 = false;
    LEG_BARRIER();
    // pivot of one column held as (e0, e1) by the lanes of a wavefront, rows already used excluded: row, value, 1 / value
    auto pick = [&](double e0, double e1, bool u0, bool u1, int& p, double& inv) {
      const double v0 = (lane < n && !u0) ? fabs(e0) : -1.0, v1 = (lane + 64 < n && !u1) ? fabs(e1) : -1.0;
      const bool second = v1 > v0;
      const double vl = second ? v1 : v0;
      const double vmax = wave_max_nonneg(fmax(vl, 0.0));
      const unsigned long long mk = __ballot(vl == vmax);
      const int src = __builtin_amdgcn_readfirstlane(mk ? __ffsll((long long)mk) - 1 : 0);
      const int ph = __builtin_amdgcn_readlane(second ? 1 : 0, src);
      p = src + 64 * ph;
      const double piv = readlane_dyn(ph ? e1 : e0, src);
      inv = __builtin_amdgcn_rcp(piv);
      inv = inv * (2.0 - piv * inv);
      inv = inv * (2.0 - piv * inv);
    };
#pragma unroll
    for (int tp = 0; tp < (NP + 2 * NWC - 1) / (2 * NWC); ++tp) {
      for (int ow = 0; ow < NWC; ++ow) {  // nw == NWC
        const int col = 2 * (NWC * tp + ow);
        if (col >= n) break;
        const bool two = col + 1 < n;
        const int pb = (NWC * tp + ow) & 1;
        double* fb = fbuf + pb * 2 * NP;
        if (wv == ow) {
          int pA, pB = 0;
          double invA, invB = 0.0;
          const double a0 = tq[0][2 * tp], a1 = tq[1][2 * tp];
          pick(a0, a1, used0, used1, pA, invA);
          const double fA0 = (lane == pA || lane >= n) ? 0.0 : a0 * invA, fA1 = (lane + 64 == pA || lane + 64 >= n) ? 0.0 : a1 * invA;
          if (lane < NP) fb[lane] = fA0;
          if (lane + 64 < NP) fb[lane + 64] = fA1;
          if (two) {
            // the second column after the first elimination (what every wavefront will compute for its own columns below)
            const double b0 = tq[0][2 * tp + 1], b1 = tq[1][2 * tp + 1];
            const double prB = (pA < 64) ? readlane_dyn(b0, pA) : readlane_dyn(b1, pA - 64);
            const double c0 = b0 - fA0 * prB, c1 = b1 - fA1 * prB;
            const bool uA0 = used0 || lane == pA, uA1 = used1 || lane + 64 == pA;
            pick(c0, c1, uA0, uA1, pB, invB);
            if (lane < NP) fb[NP + lane] = (lane == pB || lane >= n) ? 0.0 : c0 * invB;
            if (lane + 64 < NP) fb[NP + lane + 64] = (lane + 64 == pB || lane + 64 >= n) ? 0.0 : c1 * invB;
          }
          if (lane == 0) {
            pbuf[2 * pb] = pA; perm[col] = pA; iperm[pA] = col; dinv[col] = invA;
            if (two) { pbuf[2 * pb + 1] = pB; perm[col + 1] = pB; iperm[pB] = col + 1; dinv[col + 1] = invB; }
          }
        }
        LEG_BARRIER();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (q == 1 && !two) break;
          const int p = __builtin_amdgcn_readfirstlane(pbuf[2 * pb + q]);
          const double f0 = (lane < NP) ? fb[NP * q + lane] : 0.0, f1 = (lane + 64 < NP) ? fb[NP * q + lane + 64] : 0.0;
          if (lane == (p & 63)) { if (p >> 6) used1 = true; else used0 = true; }
          if (p < 64) {
#pragma unroll
            for (int sl = 2 * tp; sl < GJ_SLOTS; ++sl) { const double pr = readlane_dyn(tq[0][sl], p); tq[0][sl] -= f0 * pr; tq[1][sl] -= f1 * pr; }
          } else {
#pragma unroll
            for (int sl = 2 * tp; sl < GJ_SLOTS; ++sl) { const double pr = readlane_dyn(tq[1][sl], p - 64); tq[0][sl] -= f0 * pr; tq[1][sl] -= f1 * pr; }
          }
        }
      }
    }
    LEG_BARRIER();
    // solution in natural order over R (role 0: T1 -> Y, t3 -> rv ; role 1: T2 -> X): rows / columns >= n of the buffer cleared first
    double* Rw = role == 0 ? Y : X;
    for (int idx = tid; idx < np * ldp; idx += nthr) Rw[idx] = 0.0;
    for (int i = tid; i < np; i += nthr) rv[i] = 0.0;
    LEG_BARRIER();
#pragma unroll
    for (int sl = 0; sl < GJ_SLOTS; ++sl) {
      const int cc = col_of(sl);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = lane + 64 * h;
        if (r < n && cc >= n && cc <= 2 * n) {
          const int u = iperm[r];
          const double v = tq[h][sl] * dinv[u];
          if (cc < 2 * n) Rw[u * ldp + (cc - n)] = v; else if (role == 0) rv[u] = v;
        }
      }
    }
    LEG_BARRIER();
  }
  LCMP_PROF(26);
  LEG_LAUNDER();
  // Z <- Lm_b (both roles)
  if (bpar) leg_store_mat(Z, ldp, np, lmb, tid, nthr, S.mg_np);
  else for (int idx = tid; idx < np * ldp; idx += nthr) Z[idx] = 0.0;
  if (role == 1) {
    // ======== role 1: X = T2, Y = D, Z = Lm_b ;  Zt = T2 Lm_b (out, then over T2) ; Sg_ab = Sg_b + Lm_b^T Zt ; E = D Zt + Lm_b ========
    LEG_BARRIER();
#pragma unroll
    for (int sidx = 0; sidx < LCT; ++sidx) {
      const int t = wv + sidx * nw;
      res[sidx] = d4_t{0, 0, 0, 0};
      if (t < nb * nb) mma_tile<false>(res[sidx], X + ((t / nb) * 16) * ldp, ldp, 1, Z + (t % nb) * 16, ldp, 1, np, lane);
    }
    LEG_BARRIER();
#pragma unroll
    for (int sidx = 0; sidx < LCT; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nb * nb) {
        const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
        tile_store(X + (ri * 16) * ldp + cj * 16, ldp, res[sidx], lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; if (row < n && col < n) out[L.tZt + row * n + col] = res[sidx][q]; }
      }
    }
    LEG_BARRIER();
    const int nst = nb * (nb + 1) / 2;
#pragma unroll
    for (int sidx = 0; sidx < LCS; ++sidx) {  // Sg_ab: lower block triangle, mirrored
      const int t = wv + sidx * nw;
      if (t < nst) {
        int ri = 0, rem = t;
        while (rem > ri) { rem -= ri + 1; ++ri; }
        const int col = rem * 16 + (lane & 15);
        d4_t acc = d4_t{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; acc[q] = (row < n && col < n) ? Bn.Sg[row * n + col] : 0.0; }
        mma_tile<false>(acc, Z + ri * 16, 1, ldp, X + rem * 16, ldp, 1, np, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = ri * 16 + (lane >> 4) + 4 * q;
          if (row < n && col < n) { out[L.tSg + row * n + col] = acc[q]; if (ri != rem) out[L.tSg + col * n + row] = acc[q]; }
        }
      }
    }
#pragma unroll
    for (int sidx = 0; sidx < LCT; ++sidx) {  // E = Lm_b + D Zt
      const int t = wv + sidx * nw;
      if (t < nb * nb) {
        const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
        d4_t acc = d4_t{0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; acc[q] = (row < n && col < n) ? Bn.Lm[row * n + col] : 0.0; }
        mma_tile<false>(acc, Y + (ri * 16) * ldp, ldp, 1, X + cj * 16, ldp, 1, np, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; if (row < n && col < n) out[L.tE + row * n + col] = acc[q]; }
      }
    }
    return;
  }
  // ======== role 0: Y = T1, rv = t3, Z = Lm_b ========
  // T1 (= Zx of the down-sweep), t3 out
  for (int i = wv; i < n; i += nw) for (int c0 = lane; c0 < n; c0 += 64) out[L.tZx + i * n + c0] = Y[i * ldp + c0];
  for (int i = tid; i < n; i += nthr) out[L.tzc + i] = rv[i];
  LEG_BARRIER();
  // Lm_ab = T1^T Lm_b (out) ; sg_ab = sg_b + Lm_b^T t3
#pragma unroll
  for (int sidx = 0; sidx < LCT; ++sidx) {
    const int t = wv + sidx * nw;
    if (t < nb * nb) {
      const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
      d4_t acc = d4_t{0, 0, 0, 0};
      if (bpar) mma_tile<false>(acc, Y + ri * 16, 1, ldp, Z + cj * 16, ldp, 1, np, lane);  // T1^T: rows of the product are columns of Y
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; if (row < n && col < n) out[L.tLm + row * n + col] = acc[q]; }
    }
  }
  for (int i = wv; i < n; i += nw) {  // (Lm_b^T t3)_i = sum_l Lm_b[l][i] t3[l]: lanes over l
    double s = 0;
    for (int l = lane; l < n; l += 64) s += Z[l * ldp + i] * rv[l];
    s = wave_sum(s);
    if (lane == 0) out[L.tsg + i] = s + (bpar ? Bn.sg[i] : 0.0);
  }
  LEG_BARRIER();  // all reads of Z done
  LEG_LAUNDER();
  // ---- Z <- D (from the registers that formed it) ; F = D T1 (out, then over T1 in Y) ; u = D t3 + p_b ; Lm_a requested for the step after ----
  leg_store_mat(Z, ldp, np, dreg, tid, nthr, S.mg_np);
  double lma[LK_PT];
  leg_request_mat(lma, np, A.Lm, n, tid, nthr, S.mg_np);
  LEG_BARRIER();
#pragma unroll
  for (int sidx = 0; sidx < LCT; ++sidx) {
    const int t = wv + sidx * nw;
    res[sidx] = d4_t{0, 0, 0, 0};
    if (t < nb * nb) mma_tile<false>(res[sidx], Z + ((t / nb) * 16) * ldp, ldp, 1, Y + (t % nb) * 16, ldp, 1, np, lane);
  }
  for (int i = wv; i < np; i += nw) {
    double s = 0;
    for (int c0 = lane; c0 < n; c0 += 64) s += Z[i * ldp + c0] * rv[c0];
    s = wave_sum(s);
    if (lane == 0) { uu[i] = (i < n) ? s + pb[i] : 0.0; if (i < n) out[L.tu + i] = uu[i]; }
  }
  LEG_BARRIER();
#pragma unroll
  for (int sidx = 0; sidx < LCT; ++sidx) {
    const int t = wv + sidx * nw;
    if (t < nb * nb) {
      const int ri = t / nb, cj = t % nb, col = cj * 16 + (lane & 15);
      tile_store(Y + (ri * 16) * ldp + cj * 16, ldp, res[sidx], lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; if (row < n && col < n) out[L.tF + row * n + col] = res[sidx][q]; }
    }
  }
  LEG_LAUNDER();
  // ---- Z <- Lm_a ; P_ab = P_a + Lm_a F (lower block triangle, mirrored, diagonal tiles symmetrised through X) ; p_ab = p_a + Lm_a u ----
  leg_store_mat(Z, ldp, np, lma, tid, nthr, S.mg_np);
  LEG_BARRIER();
  {
    const int nst = nb * (nb + 1) / 2;
    d4_t pres[LCS];
#pragma unroll
    for (int sidx = 0; sidx < LCS; ++sidx) {
      const int t = wv + sidx * nw;
      pres[sidx] = d4_t{0, 0, 0, 0};
      if (t < nst) {
        int ri = 0, rem = t;
        while (rem > ri) { rem -= ri + 1; ++ri; }
        const int col = rem * 16 + (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; pres[sidx][q] = (row < n && col < n) ? A.P[row * n + col] : 0.0; }
        mma_tile<false>(pres[sidx], Z + (ri * 16) * ldp, ldp, 1, Y + rem * 16, ldp, 1, np, lane);
      }
    }
    for (int i = wv; i < n; i += nw) {
      double s = 0;
      for (int c0 = lane; c0 < n; c0 += 64) s += Z[i * ldp + c0] * uu[c0];
      s = wave_sum(s);
      if (lane == 0) out[L.tp + i] = s + A.p[i];
    }
#pragma unroll
    for (int sidx = 0; sidx < LCS; ++sidx) {
      const int t = wv + sidx * nw;
      if (t < nst) {
        int ri = 0, rem = t;
        while (rem > ri) { rem -= ri + 1; ++ri; }
        const int col = rem * 16 + (lane & 15);
        if (ri == rem) {  // diagonal tile: 0.5 (a + a^T) through its own place in X (X is dead: every tile writes and reads its own block only)
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; X[row * ldp + col] = pres[sidx][q]; }
          __builtin_amdgcn_wave_barrier();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int q = 0; q < 4; ++q) { const int row = ri * 16 + (lane >> 4) + 4 * q; pres[sidx][q] = 0.5 * (pres[sidx][q] + X[col * ldp + row]); }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = ri * 16 + (lane >> 4) + 4 * q;
          if (row < n && col < n) { out[L.tP + row * n + col] = pres[sidx][q]; if (ri != rem) out[L.tP + col * n + row] = pres[sidx][q]; }
        }
      }
    }
  }
  LCMP_PROF(27);
}
#undef LCMP_PROF

// ---------------------------------------------------------------------------------------------------------------------
// k_leg_tree_down: one launch per level, top-down (a node is created at a lower level than its parent): grid (nodes of the level, B),
// plus one workgroup for the last step of the K_0 path in the first launch.  A node: state at the cut between its children and the
// co-state parameter of the left child, x_mid = Zx x_in + Zt theta_out + zc, theta_mid = F x_in + E theta_out + u — four mat-vecs, the
// rows dealt to the wavefronts with all their loads in flight.  Cut states go straight into dxs, theta into the leg records (what
// k_leg_apply and the forward sweeps of the legs read), x_in / theta_out of the children into their records ; last, the guess of the
// value-function Hessian at the cut for the next pass (the leg record's lcP).  Launched after every sweep (the guesses), also the ones
// whose cut states are not used (first pass of a handle).
// ---------------------------------------------------------------------------------------------------------------------
template <int NP, int FN = 0, int FM = 0>
__global__ void __launch_bounds__(LK_THREADS) k_leg_tree_down(SolverArgs a, LxLds Srt, TreeDesc T, int level) {
  constexpr bool FX = FN > 0;
  constexpr LxLds SC_ = FX ? make_lx_lds(FN, FM) : LxLds{};
  LxLds S_ = Srt;
  if constexpr (FX) S_ = SC_;
  const LxLds& S = S_;
  Layout L_ = a.L;
  if constexpr (FX) { L_.n = FN; L_.m = FM; L_.nz = FN + FM; }
  const Layout& L = L_;
  const int b = blockIdx.y;
  constexpr int nthr = LK_THREADS, nw = LK_THREADS / 64;
  int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, N = L.N, J = T.J;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  if ((int)blockIdx.x == T.lev_cnt[level]) { tree_k0_step<NP, FN, FM>(a, S, T, b, T.lev_first[level], sm, tid, nthr); return; }  // (first launch only: the root)
  const int node = T.lev_first[level] + blockIdx.x;
  double* t = tree_node_ptr(a, b, node - J);
  const int lc = T.left[node], rc = T.right[node], cutleg = T.lo[rc];
  const bool has_x = T.lo[node] > 0;            // x_in = 0 on the leftmost path (forced initial condition)
  const bool has_t = T.hi[node] + 1 < J;        // no end parameter on the rightmost path
  constexpr int FW_ROWS = NP / 8;
  const int c0 = lane < n ? lane : 0, c1 = lane + 64 < n ? lane + 64 : 0;
  const double m0 = lane < n ? 1.0 : 0.0, m1 = lane + 64 < n ? 1.0 : 0.0;
  double x0 = 0, x1 = 0, t0 = 0, t1 = 0;
  if (has_x) { x0 = t[L.txin + c0] * m0; x1 = t[L.txin + c1] * m1; }
  if (has_t) { t0 = t[L.ttho + c0] * m0; t1 = t[L.ttho + c1] * m1; }
  double xm[FW_ROWS], tm[FW_ROWS], va[FW_ROWS][2], vb[FW_ROWS][2];
#pragma unroll
  for (int i = 0; i < FW_ROWS; ++i) { xm[i] = 0.0; tm[i] = 0.0; }
  if (has_x) {
#pragma unroll
    for (int i = 0; i < FW_ROWS; ++i) {
      const int r = wv + i * nw, rr = r < n ? r : 0;
      va[i][0] = t[L.tZx + rr * n + c0]; va[i][1] = t[L.tZx + rr * n + c1]; vb[i][0] = t[L.tF + rr * n + c0]; vb[i][1] = t[L.tF + rr * n + c1];
    }
#pragma unroll
    for (int i = 0; i < FW_ROWS; ++i) { xm[i] += wave_sum(va[i][0] * x0 + va[i][1] * x1); tm[i] += wave_sum(vb[i][0] * x0 + vb[i][1] * x1); }
  }
  if (has_t) {
#pragma unroll
    for (int i = 0; i < FW_ROWS; ++i) {
      const int r = wv + i * nw, rr = r < n ? r : 0;
      va[i][0] = t[L.tZt + rr * n + c0]; va[i][1] = t[L.tZt + rr * n + c1]; vb[i][0] = t[L.tE + rr * n + c0]; vb[i][1] = t[L.tE + rr * n + c1];
    }
#pragma unroll
    for (int i = 0; i < FW_ROWS; ++i) { xm[i] += wave_sum(va[i][0] * t0 + va[i][1] * t1); tm[i] += wave_sum(vb[i][0] * t0 + vb[i][1] * t1); }
  }
  double* lr = leg_ptr(a, b, cutleg - 1);
  const int cut = leg_start(a, cutleg);
  double* tl = lc >= J ? tree_node_ptr(a, b, lc - J) : nullptr;
  double* tr = rc >= J ? tree_node_ptr(a, b, rc - J) : nullptr;
#pragma unroll
  for (int i = 0; i < FW_ROWS; ++i) {
    const int r = wv + i * nw;
    if (lane == 0 && r < n) {
      const double xv = xm[i] + t[L.tzc + r], tv = tm[i] + t[L.tu + r];
      a.dxs[((size_t)b * (N + 1) + cut) * n + r] = xv;
      lr[L.lth + r] = tv;
      if (tr) { tr[L.txin + r] = xv; if (has_t) tr[L.ttho + r] = t[L.ttho + r]; }
      if (tl) { tl[L.ttho + r] = tv; if (has_x) tl[L.txin + r] = t[L.txin + r]; }
    }
  }
  // the guess of the next pass at this cut: the Hessian of the node that starts there (not in k_leg_compose: both workgroups of a
  // composition read the old guess, at their own pace)
  const double* Pr = tree_node_ref(a, T, b, rc).P;
  for (int idx = tid; idx < n * n; idx += nthr) lr[L.lcP + idx] = Pr[idx];
}
