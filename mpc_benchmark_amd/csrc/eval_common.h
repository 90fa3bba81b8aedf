// eval_common.h — pieces shared by the per-knot stage-evaluation kernels (one workgroup per knot x
// instance x linesearch candidate): term records, cost/constraint accumulation into the LQ-knot record,
// projection on the constraint sets and the per-knot merit partials.
#pragma once
#include "solver_args.h"

struct TermRec { int type, role, dim, i0, i1, poff, woff, flags; };

DEV TermRec load_term(const int32_t* desc, int t) {
  const int32_t* w = desc + MPC_STAGE_HEADER_WORDS + MPC_TERM_WORDS * t;
  TermRec r;
  r.type = w[0]; r.role = w[1]; r.dim = w[2]; r.i0 = w[3]; r.i1 = w[4]; r.poff = w[5]; r.woff = w[6]; r.flags = w[7];
  return r;
}

DEV TermRec lds_term(const int* w0, int t) {  // the same record from an LDS copy of the term table
  const int* w = w0 + MPC_TERM_WORDS * t;
  TermRec r;
  r.type = w[0]; r.role = w[1]; r.dim = w[2]; r.i0 = w[3]; r.i1 = w[4]; r.poff = w[5]; r.woff = w[6]; r.flags = w[7];
  return r;
}

// Add 1/2 r^T W r (and, when derivs, J^T W r and J^T W J) of one cost term.  r[dim], J[dim][ldj] in LDS,
// Wr/WJ are LDS scratch (dim and dim*ldj doubles).  `cost` is accumulated by thread 0 only.
DEV void accumulate_cost(const Layout& L, double* kn, const TermRec& t, const double* W, const double* r, const double* J, int ldj,
                         int nzk, double* Wr, double* WJ, bool derivs, double& cost, int tid, int nthr) {
  const int d = t.dim;
  const bool diag = t.flags & MPC_TERM_FLAG_DIAG_WEIGHT;
  for (int i = tid; i < d; i += nthr) {
    double s = 0;
    if (diag) s = W[i] * r[i];
    else for (int j = 0; j < d; ++j) s += W[i * d + j] * r[j];
    Wr[i] = s;
  }
  __syncthreads();
  if (tid == 0) { double c = 0; for (int i = 0; i < d; ++i) c += r[i] * Wr[i]; cost += 0.5 * c; }
  if (!derivs) { __syncthreads(); return; }
  for (int idx = tid; idx < d * nzk; idx += nthr) {
    const int i = idx / nzk, z = idx % nzk;
    double s = 0;
    if (diag) s = W[i] * J[i * ldj + z];
    else for (int j = 0; j < d; ++j) s += W[i * d + j] * J[j * ldj + z];
    WJ[i * ldj + z] = s;
  }
  __syncthreads();
  for (int z = tid; z < nzk; z += nthr) {
    double s = 0;
    for (int i = 0; i < d; ++i) s += J[i * ldj + z] * Wr[i];
    kn[L.oG + z] += s;
  }
  for (int idx = tid; idx < nzk * nzk; idx += nthr) {
    const int a = idx / nzk, b = idx % nzk;
    double s = 0;
    for (int i = 0; i < d; ++i) s += J[i * ldj + a] * WJ[i * ldj + b];
    kn[L.oH + a * L.nz + b] += s;
  }
  __syncthreads();
}

DEV void emit_constraint(const Layout& L, double* kn, const TermRec& t, const double* params, int row0, const double* r, const double* J,
                         int ldj, int nzk, bool derivs, int tid, int nthr) {
  for (int i = tid; i < t.dim; i += nthr) {
    kn[L.oCV + row0 + i] = r[i];
    kn[L.oCT + row0 + i] = (double)t.role;
    kn[L.oLO + row0 + i] = (t.role == MPC_ROLE_BOX) ? params[t.woff + i] : 0.0;
    kn[L.oHI + row0 + i] = (t.role == MPC_ROLE_BOX) ? params[t.woff + t.dim + i] : 0.0;
  }
  if (derivs) {
    const int lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;  // a row per wavefront: coalesced, no index divisions
    if (nw > 0) { for (int i = wv; i < t.dim; i += nw) for (int z = lane; z < nzk; z += 64) kn[L.oCD + (size_t)(row0 + i) * L.nz + z] = J[i * ldj + z]; }
    else for (int idx = tid; idx < t.dim * nzk; idx += nthr) kn[L.oCD + (row0 + idx / nzk) * L.nz + idx % nzk] = J[(idx / nzk) * ldj + idx % nzk];
  }
  __syncthreads();
}

// Zero the accumulated blocks of a knot record before the term loop.
DEV void clear_knot(const Layout& L, double* kn, int nzk, bool derivs, int tid, int nthr) {
  if (derivs) {
    for (int idx = tid; idx < nzk * nzk; idx += nthr) kn[L.oH + (idx / nzk) * L.nz + idx % nzk] = 0.0;
    for (int z = tid; z < nzk; z += nthr) kn[L.oG + z] = 0.0;
  }
  __syncthreads();
}

// Projection of the shifted constraint values, AL penalty and infeasibility of one knot; `f` is the
// dynamics gap (nullptr on the terminal knot).  Multipliers v/lam are those of the evaluated point
// (current iterate, or the trial multipliers of a linesearch candidate).  Result: pen, prim (thread 0).
DEV void knot_merit(const Layout& L, double* kn, int c, const double* f, const double* v, const double* dv, const double* ve,
                    const double* lam, const double* dlam, const double* lame, double alpha, double mu, double mud,
                    bool store_proj, double* red, double& pen_out, double& prim_out, int tid, int nthr) {
  double pen = 0, prim = 0;
  const int nt = nthr < 256 ? nthr : 256;  // `red` holds 2 x 256 partials
  if (tid < nt) {
  for (int i = tid; i < c; i += nt) {
    bool act;
    const double pn = proj_normal((int)kn[L.oCT + i], kn[L.oCV + i] + mu * ve[i], kn[L.oLO + i], kn[L.oHI + i], act);
    if (store_proj) { kn[L.oDT + i] = pn; kn[L.oACT + i] = act ? 1.0 : 0.0; }
    const double vp = pn / mu, vv = v[i] + (dv ? alpha * dv[i] : 0.0);
    pen += 0.5 * mu * vp * vp + 0.5 * mu * (vp - vv) * (vp - vv);
    prim = fmax(prim, fabs(pn - mu * ve[i]));
  }
  if (f) {
    for (int i = tid; i < L.n; i += nt) {
      const double lp = lame[i] + f[i] / mud, ll = lam[i] + (dlam ? alpha * dlam[i] : 0.0);
      pen += 0.5 * mud * lp * lp + 0.5 * mud * (lp - ll) * (lp - ll);
      prim = fmax(prim, fabs(f[i]));
    }
  }
  // deterministic reduction: DPP tree inside each wavefront, then the (at most 4) wavefront partials in order
  pen = wave_sum(pen);
  prim = wave_max_nonneg(prim);
  if ((tid & 63) == 0) { red[tid >> 6] = pen; red[8 + (tid >> 6)] = prim; }
  }
  __syncthreads();
  if (tid == 0) {
    double p = 0, q = 0;
    for (int i = 0; i < (nt >> 6); ++i) { p += red[i]; q = fmax(q, red[8 + i]); }
    pen_out = p; prim_out = q;
  }
  __syncthreads();
}

