// oracle/qp.hpp — TEST INFRASTRUCTURE: CPU restatement of the batched dense QP solver behind include/mpc_qp_abi.h
// ("next" row N3; replaces proxsuite.proxqp.dense.QP as used by QP_utils.py:437-575, 584-762).  Plain serial C++:
// proximal augmented Lagrangian (bound-constrained-Lagrangian outer loop), semismooth Newton on the active rows with
// an exact line search, Cholesky of the primal block and of the equality Schur complement.  Parity unpinned: the
// reference's solver (ProxSuite) is an absent third-party dependency and QP_utils.py has no tests; the tests pin this file
// by the KKT conditions of the convex QP (sufficient for optimality) on the reference's problem structure.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "../include/mpc_qp_abi.h"

namespace qp {

typedef std::vector<double> vec;

// in-place Cholesky of the n x n lower triangle (row-major, ld n); false if not positive definite
static inline bool chol(vec& M, int n) {
  for (int j = 0; j < n; ++j) {
    double d = M[j * n + j];
    for (int k = 0; k < j; ++k) d -= M[j * n + k] * M[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    M[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = M[i * n + j];
      for (int k = 0; k < j; ++k) s -= M[i * n + k] * M[j * n + k];
      M[i * n + j] = s / d;
    }
  }
  return true;
}
static inline void fwd(const vec& L, int n, double* v) {  // v <- L^-1 v
  for (int i = 0; i < n; ++i) { double s = v[i]; for (int k = 0; k < i; ++k) s -= L[i * n + k] * v[k]; v[i] = s / L[i * n + i]; }
}
static inline void bwd(const vec& L, int n, double* v) {  // v <- L^-T v
  for (int i = n - 1; i >= 0; --i) { double s = v[i]; for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * v[k]; v[i] = s / L[i * n + i]; }
}

struct Problem {
  int n, neq, nin, box;
  const double *H, *g, *A, *b, *C, *l, *u, *lb, *ub;
  int m() const { return nin + (box ? n : 0); }
  // row r of the stacked inequality matrix [C ; I] applied to x
  double row_dot(int r, const double* x) const {
    if (r >= nin) return x[r - nin];
    double s = 0; for (int j = 0; j < n; ++j) s += C[r * n + j] * x[j]; return s;
  }
  double lo(int r) const { return r < nin ? l[r] : lb[r - nin]; }
  double hi(int r) const { return r < nin ? u[r] : ub[r - nin]; }
};

// multiplier estimate of row r from the shifted constraint value (zk + (s - bound) / mu): > 0 upper active, < 0 lower active
static inline double zplus(double zk, double s, double lo, double hi, double mu) {
  const double tu = zk + (s - hi) / mu, tl = zk + (s - lo) / mu;
  return tu > 0.0 ? tu : (tl < 0.0 ? tl : 0.0);
}

static inline void residuals(const Problem& P, const double* x, const double* y, const double* z, double& rp, double& rd) {
  const int n = P.n, m = P.m();
  rp = 0; rd = 0;
  for (int i = 0; i < P.neq; ++i) { double s = -P.b[i]; for (int j = 0; j < n; ++j) s += P.A[i * n + j] * x[j]; rp = std::max(rp, std::fabs(s)); }
  for (int r = 0; r < m; ++r) { const double s = P.row_dot(r, x); rp = std::max(rp, std::max(s - P.hi(r), P.lo(r) - s)); }
  for (int j = 0; j < n; ++j) {
    double s = P.g[j];
    for (int k = 0; k < n; ++k) s += P.H[j * n + k] * x[k];
    for (int i = 0; i < P.neq; ++i) s += P.A[i * n + j] * y[i];
    for (int r = 0; r < P.nin; ++r) s += P.C[r * n + j] * z[r];
    if (P.box) s += z[P.nin + j];
    rd = std::max(rd, std::fabs(s));
  }
}

// x, y, z (z: nin rows then n box rows when box) hold the starting point and receive the solution
static inline void solve_one(const Problem& P, const mpc_qp_settings& S, double* x, double* y, double* z, mpc_qp_info& info) {
  const int n = P.n, neq = P.neq, m = P.m();
  double mu_eq = S.mu_eq, mu_in = S.mu_in;
  double prim_tol = std::pow(0.1, S.alpha_bcl), inner_tol = 1.0;
  vec xk(n), ye(neq), zp(m), s(m), ds(m), grad(n), r1(n), dx(n), w(n), Pm(n * n), Y(n * neq), Sm(neq * neq), yplus(neq), Ax(neq), Ad(neq), Hd(n);
  info.status = 1; info.iters = 0; info.iters_in = 0;
  auto eval = [&](const double* xx) {  // s, zp, ye, Ax at xx (multiplier estimates around the current y, z)
    for (int r = 0; r < m; ++r) { s[r] = P.row_dot(r, xx); zp[r] = zplus(z[r], s[r], P.lo(r), P.hi(r), mu_in); }
    for (int i = 0; i < neq; ++i) { double t = -P.b[i]; for (int j = 0; j < n; ++j) t += P.A[i * n + j] * xx[j]; Ax[i] = t; ye[i] = y[i] + t / mu_eq; }
  };
  for (int outer = 0; outer <= S.max_iter; ++outer) {
    double rp, rd;
    residuals(P, x, y, z, rp, rd);
    info.prim_res = rp; info.dual_res = rd;
    if (std::max(rp, rd) <= S.eps_abs) { info.status = 0; break; }
    if (outer == S.max_iter) break;
    info.iters = outer + 1;
    for (int j = 0; j < n; ++j) xk[j] = x[j];
    for (int it = 0; it < S.max_iter_in; ++it) {
      eval(x);
      double gnorm = 0;
      for (int j = 0; j < n; ++j) {
        double t = P.g[j] + S.rho * (x[j] - xk[j]);
        for (int k = 0; k < n; ++k) t += P.H[j * n + k] * x[k];
        for (int r = 0; r < P.nin; ++r) t += P.C[r * n + j] * zp[r];
        if (P.box) t += zp[P.nin + j];
        r1[j] = -t;
        for (int i = 0; i < neq; ++i) t += P.A[i * n + j] * ye[i];
        grad[j] = t;
        gnorm = std::max(gnorm, std::fabs(t));
      }
      if (gnorm <= inner_tol) break;
      info.iters_in += 1;
      // primal block
      for (int j = 0; j < n; ++j) for (int k = 0; k <= j; ++k) {
        double t = P.H[j * n + k] + (j == k ? S.rho : 0.0);
        for (int r = 0; r < P.nin; ++r) if (zp[r] != 0.0) t += P.C[r * n + j] * P.C[r * n + k] / mu_in;
        if (P.box && j == k && zp[P.nin + j] != 0.0) t += 1.0 / mu_in;
        Pm[j * n + k] = t;
      }
      if (!chol(Pm, n)) { info.status = 2; return; }
      for (int i = 0; i < neq; ++i) { for (int j = 0; j < n; ++j) w[j] = P.A[i * n + j]; fwd(Pm, n, w.data()); for (int j = 0; j < n; ++j) Y[j * neq + i] = w[j]; }
      for (int j = 0; j < n; ++j) w[j] = r1[j];
      fwd(Pm, n, w.data());
      for (int i = 0; i < neq; ++i) {
        for (int k = 0; k <= i; ++k) { double t = (i == k) ? mu_eq : 0.0; for (int j = 0; j < n; ++j) t += Y[j * neq + i] * Y[j * neq + k]; Sm[i * neq + k] = t; }
        double t = Ax[i] + mu_eq * y[i];  // -r2
        for (int j = 0; j < n; ++j) t += Y[j * neq + i] * w[j];
        yplus[i] = t;
      }
      if (!chol(Sm, neq)) { info.status = 2; return; }
      fwd(Sm, neq, yplus.data()); bwd(Sm, neq, yplus.data());
      for (int j = 0; j < n; ++j) { double t = w[j]; for (int i = 0; i < neq; ++i) t -= Y[j * neq + i] * yplus[i]; dx[j] = t; }
      bwd(Pm, n, dx.data());
      // exact line search on the (convex, piecewise quadratic) merit along dx: root of the increasing piecewise-linear phi'
      for (int r = 0; r < m; ++r) ds[r] = P.row_dot(r, dx.data());
      double a0 = 0, a1 = 0, e0 = 0, e1 = 0;
      for (int j = 0; j < n; ++j) {
        double hd = S.rho * dx[j], hx = P.g[j] + S.rho * (x[j] - xk[j]);
        for (int k = 0; k < n; ++k) { hd += P.H[j * n + k] * dx[k]; hx += P.H[j * n + k] * x[k]; }
        a0 += dx[j] * hx; a1 += dx[j] * hd;
      }
      for (int i = 0; i < neq; ++i) { double t = 0; for (int j = 0; j < n; ++j) t += P.A[i * n + j] * dx[j]; Ad[i] = t; e0 += t * ye[i]; e1 += t * t / mu_eq; }
      auto dphi = [&](double al, double& curv) {
        double f = a0 + e0 + al * (a1 + e1);
        curv = a1 + e1;
        for (int r = 0; r < m; ++r) {
          const double zr = zplus(z[r], s[r] + al * ds[r], P.lo(r), P.hi(r), mu_in);
          if (zr != 0.0) { f += ds[r] * zr; curv += ds[r] * ds[r] / mu_in; }
        }
        return f;
      };
      double alpha = 1.0, lo_a = 0.0, hi_a = -1.0, curv;
      for (int ls = 0; ls < 40; ++ls) {
        const double f = dphi(alpha, curv);
        if (std::fabs(f) <= 1e-13 * (std::fabs(a0 + e0) + 1.0)) break;
        if (f < 0) lo_a = alpha; else hi_a = alpha;
        double an = alpha - f / curv;
        if (an <= lo_a || (hi_a > 0 && an >= hi_a)) an = (hi_a > 0) ? 0.5 * (lo_a + hi_a) : 2.0 * alpha;
        if (std::fabs(an - alpha) <= 1e-15 * alpha) { alpha = an; break; }
        alpha = an;
      }
      double stepn = 0, xn = 1.0;
      for (int j = 0; j < n; ++j) { stepn = std::max(stepn, std::fabs(alpha * dx[j])); xn = std::max(xn, std::fabs(x[j])); x[j] += alpha * dx[j]; }
      if (stepn <= 1e-14 * xn) break;  // round-off floor (multiplier estimates carry an error of size eps / mu): no further progress
    }
    eval(x);
    double rp_eq = 0, rp_in = 0;
    for (int i = 0; i < neq; ++i) rp_eq = std::max(rp_eq, std::fabs(Ax[i]));
    for (int r = 0; r < m; ++r) rp_in = std::max(rp_in, std::max(s[r] - P.hi(r), P.lo(r) - s[r]));
    const double rpn = std::max(rp_eq, rp_in);
    if (rpn <= prim_tol) {  // BCL: good step — take the multipliers, tighten
      for (int i = 0; i < neq; ++i) y[i] = ye[i];
      for (int r = 0; r < m; ++r) z[r] = zp[r];
      prim_tol = std::max(prim_tol * std::pow(mu_in, S.beta_bcl), S.eps_abs);
      inner_tol = std::max(inner_tol * mu_in, S.eps_abs);
    } else {  // bad step — keep the multipliers, sharpen the penalties
      // only the penalty of the block that is infeasible: 1 / mu amplifies round-off in the multiplier estimates
      if (rp_eq > prim_tol) mu_eq = std::max(mu_eq * S.mu_update_factor, S.mu_min_eq);
      if (rp_in > prim_tol) mu_in = std::max(mu_in * S.mu_update_factor, S.mu_min_in);
      prim_tol = std::max(std::pow(mu_in, S.alpha_bcl) * std::pow(0.1, S.alpha_bcl), S.eps_abs);
      inner_tol = std::max(mu_in, S.eps_abs);
    }
  }
  info.mu_eq = mu_eq; info.mu_in = mu_in;
  int na = 0;
  for (int r = 0; r < m; ++r) if (z[r] != 0.0) ++na;
  info.n_active = na;
}

}  // namespace qp
