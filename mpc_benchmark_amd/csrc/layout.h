// layout.h — HBM data layout of the MI355X ProxDDP solver (see DESIGN.md §"Data layout in HBM").
// Everything is fp64, row-major, batch-major: [instance][knot][block].  One LQ knot record holds what the
// per-knot evaluation kernel produces and the Riccati kernel consumes; one gain record
// holds what the backward sweep produces and the forward sweep consumes (K8/K9).
#pragma once
#include <stdint.h>

struct Layout {
  int N, B, space, nx, n, m, c, nz;
  int nj;  // moving joints of the multibody model (0 for vector spaces)
  int model_mask_off;  // int32 offset of the 64-bit tree masks (ancestors | subtree | path dofs | dofs strictly below, nj each) in the device model table
  // offsets inside one knot record (doubles)
  int oH, oG, oAB, oF, oE6, oT6k, oD12, oCV, oCD, oLO, oHI, oDT, oACT, oCT, oMISC, oXD, oWR, oXN, knot_stride;  // (oT6k: T6 = (-E6)^-1, written beside E6 by the whole-body stage kernel)
  // offsets inside one gain record
  int oP, op, oK, ok, oKnu, oknu, oMx, omx, oT6, oPhi, ophi, gain_stride;
  // parallel-in-time legs (legs.h), knots of a parametric leg: (u,u) and (nu,u) blocks of the inverse stage KKT matrix (Riccati sweep),
  // Gamma = d dx'/dp', Ku = dk/dp', Knup = dknu/dp' (k_leg_knot), Lm = dp/dtheta (k_leg_condense)
  int oMu, oZnu, oGam, oKu, oKnup, oLm, mpad;
  // per (instance, parametric leg j) record: Sg (n x n) | sg | Zx (n x n) | zc | calP (n x n: exact value-function Hessian at the start of
  // leg j + 1 — kept across passes: the terminal cost of leg j in the next pass) | calp | theta | dP (n x n: calP minus the guess the leg carried)
  int lSg, lsg, lZx, lzc, lcP, lcp, lth, ldP, leg_stride;
  // tree over the cuts (legs_tree.h), per (instance, inner node): condensed form P | Lm | Sg | p | sg of the legs it covers and the maps of
  // the down-sweep Zx | Zt | F | E | zc | u
  int tP, tLm, tSg, tZx, tZt, tF, tE, tp, tsg, tzc, tu, txin, ttho, tree_stride;  // (txin, ttho: state at the node's start, parameter at its end)
  // backward-sweep scratch per instance
  int wPh, wPt, wLp, wG, wHh, wgh, wCt, wW, wY, wSc, wV, wAcl, wvec, work_stride;
  int max_stage_ints, max_stage_doubles;
  int n_alpha;  // number of linesearch candidates evaluated per iteration
  int sc_cap;   // active-constraint Schur complements up to sc_cap x sc_cap are factorised in LDS, larger ones in HBM scratch
};

// misc slots of a knot record
enum { MISC_COST = 0, MISC_PEN = 1, MISC_PRIM = 2, MISC_NC = 3, MISC_M = 4, MISC_DUAL = 5, MISC_CRIT = 6, MISC_DMERIT = 7, MISC_COUNT = 8 };

// per-instance solver state kept on the device (the BCL outer loop runs there)
struct InstState {
  double mu, inner_tol, prim_tol;
  double phi0, dphi0, alpha, cost, prim, dual, crit;
  int32_t num_iters, al_iters, converged, done, skip_step, ls_step, ls_more, stalled;
  int32_t corrector, pad_;  // corrector: this run has been granted its one extra iteration (mpc_options.corrector_prim_tol)
};

#define MPC_MAX_LEGS 32  // riccati_legs is clamped to this (and to the horizon) ; measured up to 64: batch 1 is fastest with 32 (1.47 ms against 1.57 with 16), 40 - 64 no better

static inline int align2(int x) { return (x + 1) & ~1; }

static inline void make_layout(Layout& L) {
  const int n = L.n, m = L.m, c = L.c, nz = L.nz = n + m;
  int o = 0;
  auto take = [&](int cnt) { int r = o; o += align2(cnt); return r; };
  L.oH = take(nz * nz); L.oG = take(nz); L.oAB = take(n * nz); L.oF = take(n); L.oE6 = take(36); L.oT6k = take(36);
  // structure of the semi-implicit Euler rows: [A B]_q = D1 [I 0 0] + Dd [A B]_v with D1, Dd = identity, dt on the joints and 6x6 on the
  // base: D1_b (36) | Dd_b (36) | dt | valid flag — the Riccati sweep then multiplies with the v rows of [A B] only
  L.oD12 = take(74);
  L.oCV = take(c); L.oCD = take(c * nz); L.oLO = take(c); L.oHI = take(c); L.oDT = take(c); L.oACT = take(c); L.oCT = take(c);
  L.oMISC = take(MISC_COUNT); L.oXD = take(n); L.oWR = take(12); L.oXN = take(L.nx);
  L.knot_stride = o;
  o = 0;
  L.oP = take(n * n); L.op = take(n); L.oK = take(m * n); L.ok = take(m); L.oKnu = take(c * n); L.oknu = take(c);
  L.oMx = take(n * n); L.omx = take(n); L.oT6 = take(36);
  L.oPhi = take(n * n); L.ophi = take(n);  // closed-loop transition dx' = Phi dx + phi (closed_loop.h)
  L.mpad = (m + 15) & ~15;
  L.oMu = take(L.mpad * L.mpad); L.oZnu = take(c * L.mpad); L.oGam = take(n * n); L.oKu = take(m * n); L.oKnup = take(c * n); L.oLm = take(n * n);
  L.gain_stride = o;
  o = 0;
  L.lSg = take(n * n); L.lsg = take(n); L.lZx = take(n * n); L.lzc = take(n); L.lcP = take(n * n); L.lcp = take(n); L.lth = take(n); L.ldP = take(n * n);
  L.leg_stride = o;
  o = 0;
  L.tP = take(n * n); L.tLm = take(n * n); L.tSg = take(n * n); L.tZx = take(n * n); L.tZt = take(n * n); L.tF = take(n * n); L.tE = take(n * n);
  L.tp = take(n); L.tsg = take(n); L.tzc = take(n); L.tu = take(n); L.txin = take(n); L.ttho = take(n);
  L.tree_stride = o;
  o = 0;
  const int nr = n + 1;
  L.wPh = take(n * n); L.wPt = take(n * n); L.wLp = take(n * n); L.wG = take(n * nz); L.wHh = take(nz * nz); L.wgh = take(nz);
  L.wCt = take(c * nz); L.wW = take(m * nr); L.wY = take(m * c); L.wSc = take(c * c); L.wV = take(c * nr); L.wAcl = take(n * nr);
  L.wvec = take(8 * (nz + c));
  L.work_stride = o;
  L.sc_cap = c < 48 ? c : 48;
}
