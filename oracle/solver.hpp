// oracle/solver.hpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
// CPU restatement of one SolverProxDDP.run (SURVEY.md §3.3 phases P1-P9, App. B.3-B.4):
// primal-dual augmented-Lagrangian (ProxDDP, Jallet et al.) with BCL outer loop, Gauss-Newton Hessians,
// proximal Riccati backward/forward sweep, ROLLOUT_LINEAR Armijo backtracking, forced initial condition
// — the configuration the reference sets at fulldynamic_talos.py:374-386.
// Upstream Aligator is not vendored (README.md:10) -> PARITY UNPINNED; the LQ step is cross-checked against
// a dense KKT solve and the iteration against its own optimality conditions in tests/.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
// The whole-body stage evaluation: the oracle's own (forward-mode AD, stage.hpp) — or, in the build of oracle/cpu_port/ only, the
// closed-form port that bench.py times as its CPU baseline (never part of the checker library).
#ifdef ORC_PROFILE
#include <cstdio>
namespace orc { struct ProfAcc { double t[16] = {}; double last = 0; ~ProfAcc() { for (int i = 0; i < 16; ++i) if (t[i] > 0) fprintf(stderr, "[orc prof] section %d: %.1f ms\n", i, t[i] * 1e3); } };
inline ProfAcc& prof_acc() { static ProfAcc a; return a; } }
#define ORC_PROF_START() do { if (omp_get_thread_num() == 0) orc::prof_acc().last = omp_get_wtime(); } while (0)
#define ORC_PROF(i) do { if (omp_get_thread_num() == 0) { const double t_ = omp_get_wtime(); orc::prof_acc().t[i] += t_ - orc::prof_acc().last; orc::prof_acc().last = t_; } } while (0)
#else
#define ORC_PROF_START() do {} while (0)
#define ORC_PROF(i) do {} while (0)
#endif
#ifndef ORC_EVAL_MULTIBODY
#define ORC_EVAL_MULTIBODY eval_multibody
#endif
#endif
#include "stage.hpp"

namespace orc {

// ---------------------------------------------------------------------------------------------
// small dense helpers (row-major)
// ---------------------------------------------------------------------------------------------
inline bool chol_lower(double* A, int n) {  // in place, lower triangle; returns false if not SPD
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  return true;
}
// solve L Y = B (forward), B is n x r
inline void trsm_lower(const double* L, int n, double* B, int r) {
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < i; ++k) { const double l = L[i * n + k]; if (l != 0.0) for (int j = 0; j < r; ++j) B[i * r + j] -= l * B[k * r + j]; }
    const double inv = 1.0 / L[i * n + i];
    for (int j = 0; j < r; ++j) B[i * r + j] *= inv;
  }
}
// solve L^T X = B (backward)
inline void trsm_lower_t(const double* L, int n, double* B, int r) {
  for (int i = n - 1; i >= 0; --i) {
    for (int k = i + 1; k < n; ++k) { const double l = L[k * n + i]; if (l != 0.0) for (int j = 0; j < r; ++j) B[i * r + j] -= l * B[k * r + j]; }
    const double inv = 1.0 / L[i * n + i];
    for (int j = 0; j < r; ++j) B[i * r + j] *= inv;
  }
}
inline void inv6(const double* A, double* Ainv) {
  std::vector<double> M(A, A + 36), I(36, 0.0);
  for (int i = 0; i < 6; ++i) I[i * 6 + i] = 1.0;
  solve_dense(M, 6, I, 6);
  std::memcpy(Ainv, I.data(), 36 * sizeof(double));
}

// projection of z on the normal cone of the constraint set of one row; returns value and active flag
inline double proj_normal(int role, double z, double lo, double hi, bool& active) {
  switch (role) {
    case MPC_ROLE_EQUALITY: active = true; return z;
    case MPC_ROLE_NEG_ORTHANT: active = z > 0.0; return active ? z : 0.0;
    case MPC_ROLE_BOX:
      if (z < lo) { active = true; return z - lo; }
      if (z > hi) { active = true; return z - hi; }
      active = false; return 0.0;
  }
  active = false;
  return 0.0;
}

struct Gains {
  std::vector<double> P, p, K, kff, Knu, knu, Mx, mx;
  double T6[36];  // Ebar^{-1} base block (Ebar = -E6)
  // parallel-in-time legs (riccati_legs > 1), knots of a leg whose end co-state theta is a parameter:
  // Lm = dp/dtheta (n x n), Kth = dk/dtheta (m x n), Knuth = dknu/dtheta (c x n), Mth = d dx'/dtheta (n x n),
  // Sg, sg: dx_cut = Lm^T dx + Sg theta + sg (condensed leg from this knot to its end)
  std::vector<double> Lm, Kth, Knuth, Mth, Sg, sg, Kexact;
  std::vector<double> mx0, p0, kff0;  // affine terms of a parametric knot before the co-state of its leg is applied
  std::vector<double> Pt, Mu, Znu;  // parametric knots only: Pt, and the (u,u) / (nu,u) blocks of the inverse stage KKT matrix (what csrc/legs.h builds Gamma, Ku, Knup from)
};

// node of the tree over the legs (legs lo..hi composed): its condensed form (P, p: value function at its first knot given the guess at its
// end; Lm, Sg, sg: p = p0 + Lm theta, x_end = Lm^T x + Sg theta + sg) and, for an inner node, what recovers the state and co-state
// parameter at the cut between its children: x_mid = Zx x_in + Zt theta_out + zc, theta_mid = D x_mid + Lm_right theta_out + p_right
struct TreeNode { int lo = 0, hi = 0, left = -1, right = -1; std::vector<double> P, p, Lm, Sg, sg, Zx, Zt, zc, D; };

struct Instance {
  std::vector<double> x0;
  std::vector<std::vector<double>> xs, us, vs, lams, vs_e, lams_e;
  std::vector<std::vector<double>> dxs, dus, dvs, dlams;
  std::vector<std::vector<double>> txs, tus, tvs, tlams;  // trial point
  std::vector<Knot> knots, tknots;
  std::vector<Gains> gains;
  double mu = 0, inner_tol = 0, prim_tol = 0;
  mpc_stats stats{};
  // parallel-in-time legs, per parametric leg (debug dumps): Zx | zc | calP | calp | theta
  std::vector<std::vector<double>> leg_Zx, leg_zc, leg_calP, leg_calp, leg_theta;
  // tree over the cuts (three legs or more): nodes of the last pass, and the value-function guesses at the cuts kept from pass to pass
  std::vector<struct TreeNode> tree;
  std::vector<std::vector<double>> tree_guess;
  bool tree_guess_valid = false;
};

struct Solver {
  mpc_dims dims{};
  mpc_options opt{};
  Model model;
  std::vector<StageDesc> stages;  // N + 1
  std::vector<Instance> inst;
  std::vector<std::vector<std::vector<double>>> inst_params;  // [B][N + 1]: per-instance parameter tables, empty = shared (mpc_enable_instance_params)
  std::string err;
  bool have_model = false;
  bool corrector_armed = true;  // mpc_options.corrector_window: set per run by the C-ABI layer (capi.cpp), which sees the mpc_cycle calls

  int N() const { return dims.horizon; }

  void set_default_options() {
    opt.tol = 1e-5; opt.mu_init = 1e-8; opt.dyn_al_scale = 1e-3; opt.reg_init = 1e-9;
    opt.ls_armijo_c1 = 1e-4; opt.ls_alpha_min = 1e-7;
    opt.bcl_prim_alpha = 0.1; opt.bcl_prim_beta = 0.9; opt.bcl_dual_alpha = 1.0; opt.bcl_dual_beta = 1.0;
    opt.bcl_mu_update_factor = 0.01; opt.bcl_mu_lower_bound = 1e-8;
    opt.inner_tol0 = 1.0; opt.prim_tol0 = 1.0;
    opt.max_iters = 100; opt.max_al_iters = 100; opt.force_initial_condition = 1; opt.rollout_linear = 1;
    opt.ls_max_steps = 8; opt.num_threads = 1; opt.riccati_legs = 1; opt.refine_appended_knot = 0;
    opt.corrector_prim_tol = 0.0; opt.corrector_window = 0;
  }

  void init(const mpc_dims& d) {
    dims = d;
    set_default_options();
    stages.assign(d.horizon + 1, StageDesc());
    inst.assign(d.batch, Instance());
    for (auto& in : inst) {
      in.x0.assign(d.nx, 0.0);
      alloc(in);
    }
  }
  void alloc(Instance& in) {
    const int N = dims.horizon;
    auto mk = [&](std::vector<std::vector<double>>& v, int cnt, int len) { v.assign(cnt, std::vector<double>(len, 0.0)); };
    mk(in.xs, N + 1, dims.nx); mk(in.us, N, dims.nu); mk(in.vs, N + 1, dims.nc_max); mk(in.lams, N + 1, dims.ndx);
    mk(in.vs_e, N + 1, dims.nc_max); mk(in.lams_e, N + 1, dims.ndx);
    mk(in.dxs, N + 1, dims.ndx); mk(in.dus, N, dims.nu); mk(in.dvs, N + 1, dims.nc_max); mk(in.dlams, N + 1, dims.ndx);
    mk(in.txs, N + 1, dims.nx); mk(in.tus, N, dims.nu); mk(in.tvs, N + 1, dims.nc_max); mk(in.tlams, N + 1, dims.ndx);
    in.knots.assign(N + 1, Knot()); in.tknots.assign(N + 1, Knot()); in.gains.assign(N + 1, Gains());
  }

  // ---- manifold ops -------------------------------------------------------------------------
  void integrate(const double* x, const double* dx, double* out) const {
    if (dims.space == MPC_SPACE_VECTOR) for (int i = 0; i < dims.nx; ++i) out[i] = x[i] + dx[i];
    else mb_integrate(model, x, dx, out);
  }

  void eval_knot(int b, int k, const double* x, const double* u, const double* xnext, Knot& kn, bool derivs) const {
    StageDesc own;  // per-instance parameters (mpc_enable_instance_params): the shared descriptor with the instance's own parameter table
    if (!inst_params.empty()) { own = stages[k]; own.params = inst_params[b][k]; }
    const StageDesc& sd = inst_params.empty() ? stages[k] : own;
    if (sd.nc > dims.nc_max) throw std::runtime_error("stage has more constraint rows than nc_max");
    if (dims.space == MPC_SPACE_VECTOR) eval_centroidal(sd, dims.nx, dims.nu, x, u, xnext, kn, derivs);
    else ORC_EVAL_MULTIBODY(model, sd, dims.nu, x, u, xnext, kn, derivs);
    if (derivs) {
      const int nz = kn.n + kn.m;
      for (int i = 0; i < nz; ++i) kn.H[i * nz + i] += opt.reg_init;
    }
  }

  // P1 + P3: evaluate every knot of one instance (values, and derivatives when requested)
  void evaluate(Instance& in, const std::vector<std::vector<double>>& xs, const std::vector<std::vector<double>>& us,
                std::vector<Knot>& knots, bool derivs) const {
    const int N = dims.horizon;
    const int b_inst = (int)(&in - inst.data());
    std::string omp_err;
#pragma omp parallel for schedule(dynamic) num_threads(opt.num_threads > 0 ? opt.num_threads : 1)
    for (int k = 0; k <= N; ++k) {
      try {
        eval_knot(b_inst, k, xs[k].data(), k < N ? us[k].data() : nullptr, k < N ? xs[k + 1].data() : nullptr, knots[k], derivs);
      } catch (const std::exception& e) {
#pragma omp critical
        omp_err = e.what();
      }
    }
    if (!omp_err.empty()) throw std::runtime_error(omp_err);
  }

  double mu_dyn(const Instance& in) const { return in.mu * opt.dyn_al_scale; }

  // P2 + merit: first-order multiplier estimates and primal-dual AL merit at (knots, vs, lams)
  //   M = sum l + sum_k mu/2 |v+|^2 + mu/2 |v+ - v|^2 + sum_k mud/2 |l+|^2 + mud/2 |l+ - l|^2
  double merit(const Instance& in, const std::vector<Knot>& knots, const std::vector<std::vector<double>>& vs,
               const std::vector<std::vector<double>>& lams, double* cost_out = nullptr, double* prim_out = nullptr) const {
    const int N = dims.horizon;
    const double mu = in.mu, mud = mu_dyn(in);
    double cost = 0, pen = 0, prim = 0;
    for (int k = 0; k <= N; ++k) {
      const Knot& kn = knots[k];
      cost += kn.cost;
      for (int i = 0; i < kn.c; ++i) {
        bool act;
        const double z = kn.cval[i] + mu * in.vs_e[k][i];
        const double pn = proj_normal(kn.ctype[i], z, kn.lo[i], kn.hi[i], act);
        const double vp = pn / mu;
        pen += 0.5 * mu * vp * vp + 0.5 * mu * (vp - vs[k][i]) * (vp - vs[k][i]);
        prim = std::max(prim, std::fabs(pn - mu * in.vs_e[k][i]));
      }
      if (k < N) {
        for (int i = 0; i < kn.n; ++i) {
          const double lp = in.lams_e[k + 1][i] + kn.f[i] / mud;
          pen += 0.5 * mud * lp * lp + 0.5 * mud * (lp - lams[k + 1][i]) * (lp - lams[k + 1][i]);
          prim = std::max(prim, std::fabs(kn.f[i]));
        }
      }
    }
    if (cost_out) *cost_out = cost;
    if (prim_out) *prim_out = prim;
    return cost + pen;
  }

  // P4: dual infeasibility (gradient of the Lagrangian) and inner criterion at the current iterate
  void lagrangian_residuals(const Instance& in, double& dual_infeas, double& inner_crit) const {
    const int N = dims.horizon, n = dims.ndx;
    const double mu = in.mu, mud = mu_dyn(in);
    dual_infeas = 0; inner_crit = 0;
    for (int k = 0; k <= N; ++k) {
      const Knot& kn = in.knots[k];
      const int nz = kn.n + kn.m;
      std::vector<double> L(kn.grad);
      for (int i = 0; i < kn.c; ++i) { const double v = in.vs[k][i]; if (v != 0.0) for (int a = 0; a < nz; ++a) L[a] += kn.CD[i * nz + a] * v; }
      if (k < N) for (int i = 0; i < n; ++i) { const double l = in.lams[k + 1][i]; if (l != 0.0) for (int a = 0; a < nz; ++a) L[a] += kn.AB[i * nz + a] * l; }
      if (k > 0) {
        // E_{k-1}^T lambda_k : E = blockdiag(E6, -I)
        const Knot& kp = in.knots[k - 1];
        std::vector<double> El(n);
        for (int i = 0; i < n; ++i) El[i] = -in.lams[k][i];
        if (dims.space == MPC_SPACE_MULTIBODY && model.has_freeflyer())
          for (int j = 0; j < 6; ++j) { double s = 0; for (int i = 0; i < 6; ++i) s += kp.E6[i * 6 + j] * in.lams[k][i]; El[j] = s; }
        for (int i = 0; i < n; ++i) L[i] += El[i];
      }
      const int a0 = (k == 0 && opt.force_initial_condition) ? kn.n : 0;  // x0 is fixed
      for (int a = a0; a < nz; ++a) dual_infeas = std::max(dual_infeas, std::fabs(L[a]));
      for (int i = 0; i < kn.c; ++i) {
        bool act;
        const double pn = proj_normal(kn.ctype[i], kn.cval[i] + mu * in.vs_e[k][i], kn.lo[i], kn.hi[i], act);
        inner_crit = std::max(inner_crit, std::fabs(pn - mu * in.vs[k][i]));
      }
      if (k < N) for (int i = 0; i < n; ++i) inner_crit = std::max(inner_crit, std::fabs(kn.f[i] + mud * (in.lams_e[k + 1][i] - in.lams[k + 1][i])));
    }
    inner_crit = std::max(inner_crit, dual_infeas);
  }

  // P6: proximal Riccati backward sweep, unpivoted quasi-definite elimination
  // (controls first, then constraint multipliers).
  void backward_terminal(Instance& in) const {
    const int N = dims.horizon, n = dims.ndx;
    const double mu = in.mu;
    const Knot& kn = in.knots[N];
    Gains& g = in.gains[N];
    g.P.assign(kn.H.begin(), kn.H.end()); g.p.assign(kn.grad.begin(), kn.grad.end());
    g.Knu.assign(kn.c * n, 0.0); g.knu.assign(kn.c, 0.0);
    g.Lm.clear(); g.Kth.clear(); g.Knuth.clear(); g.Mth.clear(); g.Sg.clear(); g.sg.clear();
    for (int i = 0; i < kn.c; ++i) {
      bool act;
      const double pn = proj_normal(kn.ctype[i], kn.cval[i] + mu * in.vs_e[N][i], kn.lo[i], kn.hi[i], act);
      g.knu[i] = pn / mu;
      if (!act) continue;
      for (int a = 0; a < n; ++a) g.Knu[i * n + a] = kn.CD[i * n + a] / mu;
      for (int a = 0; a < n; ++a) {
        g.p[a] += kn.CD[i * n + a] * g.knu[i];
        for (int b = 0; b < n; ++b) g.P[a * n + b] += kn.CD[i * n + a] * g.Knu[i * n + b];
      }
    }
  }

  // One backward step at knot k < N from the value function of knot k + 1,
  //   V'(x', theta) = 1/2 x'^T Pn x' + x'^T (pn + Lmn theta) + 1/2 theta^T Sgn theta + sgn^T theta .
  // Lmn == nullptr: the plain recursion (no parameter).  With a parameter (the co-state at the end of a leg of the
  // parallel-in-time solver) every affine quantity gains n columns: Kth, Knuth, Lm = dp/dtheta, Mth = d dx'/dtheta, and the
  // envelope theorem gives Sg = Sgn + Lmn^T Mth, sg = sgn + Lmn^T mx.
  void knot_backward(const Instance& in, int k, const std::vector<double>& Pn, const std::vector<double>& pn,
                     const std::vector<double>* Lmn, const std::vector<double>* Sgn, const std::vector<double>* sgn, Gains& g) const {
    const int n = dims.ndx;
    const double mu = in.mu, mud = mu_dyn(in);
    const bool ff = dims.space == MPC_SPACE_MULTIBODY && model.has_freeflyer();
    const Knot& kn = in.knots[k];
    const int m = kn.m, c = kn.c, nz = n + m;
    const int np = Lmn ? n : 0;
    ORC_PROF_START();
    // 1. change of variable y = Ebar x'
    for (int i = 0; i < 36; ++i) g.T6[i] = (i % 7 == 0) ? 1.0 : 0.0;
    std::vector<double> Ph(Pn), ph(pn), phM;
    if (np) phM = *Lmn;
    if (ff) {
      double Eb[36];
      for (int i = 0; i < 36; ++i) Eb[i] = -kn.E6[i];
      inv6(Eb, g.T6);
      // Ph = T^T P T, ph = T^T p with T = blockdiag(T6, I)
      std::vector<double> tmp(n * n);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          if (j < 6) { double s = 0; for (int l = 0; l < 6; ++l) s += Pn[i * n + l] * g.T6[l * 6 + j]; tmp[i * n + j] = s; }
          else tmp[i * n + j] = Pn[i * n + j];
        }
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          if (i < 6) { double s = 0; for (int l = 0; l < 6; ++l) s += g.T6[l * 6 + i] * tmp[l * n + j]; Ph[i * n + j] = s; }
          else Ph[i * n + j] = tmp[i * n + j];
        }
      for (int i = 0; i < 6; ++i) { double s = 0; for (int l = 0; l < 6; ++l) s += g.T6[l * 6 + i] * pn[l]; ph[i] = s; }
      for (int j = 0; j < np; ++j) for (int i = 0; i < 6; ++i) { double s = 0; for (int l = 0; l < 6; ++l) s += g.T6[l * 6 + i] * (*Lmn)[l * np + j]; phM[i * np + j] = s; }
    }
    ORC_PROF(1);
    // 2. Lam = (I + mud Ph)^-1 ; Pt = Lam Ph ; pt = Lam (Ph ft + ph)
    std::vector<double> Lp(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Lp[i * n + j] = mud * 0.5 * (Ph[i * n + j] + Ph[j * n + i]) + (i == j ? 1.0 : 0.0);
    if (!chol_lower(Lp.data(), n)) throw std::runtime_error("Riccati: I + mu_dyn P not positive definite");
    std::vector<double> Pt(Ph), ft(n), w(n), wM(phM);
    for (int i = 0; i < n; ++i) ft[i] = kn.f[i] + mud * in.lams_e[k + 1][i];
    for (int i = 0; i < n; ++i) { double s = ph[i]; for (int j = 0; j < n; ++j) s += Ph[i * n + j] * ft[j]; w[i] = s; }
    trsm_lower(Lp.data(), n, Pt.data(), n); trsm_lower_t(Lp.data(), n, Pt.data(), n);
    trsm_lower(Lp.data(), n, w.data(), 1); trsm_lower_t(Lp.data(), n, w.data(), 1);
    if (np) { trsm_lower(Lp.data(), n, wM.data(), np); trsm_lower_t(Lp.data(), n, wM.data(), np); }
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { const double s = 0.5 * (Pt[i * n + j] + Pt[j * n + i]); Pt[i * n + j] = Pt[j * n + i] = s; }
    ORC_PROF(2);
    // 3. Hh = H + AB^T Pt AB ; gh = grad + AB^T pt
    std::vector<double> G(n * nz, 0.0), Hh(kn.H), gh(kn.grad), ghM((size_t)nz * np, 0.0);
    for (int i = 0; i < n; ++i) for (int l = 0; l < n; ++l) { const double pv = Pt[i * n + l]; if (pv != 0.0) for (int a = 0; a < nz; ++a) G[i * nz + a] += pv * kn.AB[l * nz + a]; }
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < nz; ++a) {
        const double ab = kn.AB[i * nz + a];
        if (ab == 0.0) continue;
        gh[a] += ab * w[i];
        for (int b = a; b < nz; ++b) Hh[a * nz + b] += ab * G[i * nz + b];  // upper triangle (AB^T Pt AB is symmetric: Pt was symmetrised), mirrored below
        for (int j = 0; j < np; ++j) ghM[a * np + j] += ab * wM[i * np + j];
      }
    for (int a = 0; a < nz; ++a) for (int b = a + 1; b < nz; ++b) Hh[b * nz + a] = Hh[a * nz + b] - kn.H[a * nz + b] + kn.H[b * nz + a];
    ORC_PROF(3);
    // 4. stage KKT  [[R, D^T],[D, -mu I]] [U; V] = -[[S^T r],[C d]]
    std::vector<double> Lr(m * m), Ct(c * nz, 0.0), dt_(c, 0.0);
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Lr[i * m + j] = 0.5 * (Hh[(n + i) * nz + n + j] + Hh[(n + j) * nz + n + i]);
    for (int attempt = 1; !chol_lower(Lr.data(), m); ++attempt) {
      // inertia correction (same rule as csrc/riccati_mfma.h): Ruu + rho I, rho = max(1e-8, 1e-6 max|diag|) x 10^t
      if (attempt > 10) throw std::runtime_error("Riccati: reduced control Hessian not positive definite");
      double dmax = 0.0;
      for (int i = 0; i < m; ++i) dmax = std::max(dmax, std::fabs(Hh[(n + i) * nz + n + i]));
      double rho = std::max(1e-8, 1e-6 * dmax);
      for (int t = 1; t < attempt; ++t) rho *= 10.0;
      for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j)
        Lr[i * m + j] = 0.5 * (Hh[(n + i) * nz + n + j] + Hh[(n + j) * nz + n + i]) + (i == j ? rho : 0.0);
    }
    // Only the ACTIVE rows enter the elimination: an inactive row (zero Jacobian row, Pi_N(z) = 0) decouples — its Schur diagonal is
    // mu, its gains are zero — so the compacted system gives the same numbers as the full one (same operations in the same order
    // on the active rows) at a fraction of the work: c = 98 rows on a double-support knot, about a dozen active.
    std::vector<int> act;
    act.reserve(c);
    for (int i = 0; i < c; ++i) {
      bool a_;
      dt_[i] = proj_normal(kn.ctype[i], kn.cval[i] + mu * in.vs_e[k][i], kn.lo[i], kn.hi[i], a_);
      if (a_) { act.push_back(i); for (int a = 0; a < nz; ++a) Ct[i * nz + a] = kn.CD[i * nz + a]; }
    }
    const int ca = (int)act.size();
    const int nr = n + 1 + np;  // columns: feedback on dx | feed-forward | feedback on theta
    std::vector<double> W((size_t)m * nr), Yr((size_t)(ca > 0 ? ca : 1) * m, 0.0);  // Yr: row r = L^-1 d_r of active row r (contiguous)
    for (int i = 0; i < m; ++i) {
      for (int a = 0; a < n; ++a) W[i * nr + a] = -Hh[(n + i) * nz + a];
      W[i * nr + n] = -gh[n + i];
      for (int j = 0; j < np; ++j) W[i * nr + n + 1 + j] = -ghM[(n + i) * np + j];
    }
    trsm_lower(Lr.data(), m, W.data(), nr);  // W = L^-1 T
    std::vector<double> Vc((size_t)(ca > 0 ? ca : 1) * nr, 0.0), Sc((size_t)(ca > 0 ? ca : 1) * ca, 0.0), tmp(nr > m ? nr : m);
    if (ca > 0) {
      for (int r = 0; r < ca; ++r) {  // Y = L^-1 D^T, one active row at a time (forward substitution, dot-product form)
        double* y = Yr.data() + (size_t)r * m;
        const double* d = Ct.data() + (size_t)act[r] * nz + n;
        for (int i = 0; i < m; ++i) { double sacc = d[i]; const double* Li = Lr.data() + (size_t)i * m; for (int l = 0; l < i; ++l) sacc -= Li[l] * y[l]; y[i] = sacc / Li[i]; }
      }
      for (int r = 0; r < ca; ++r) for (int q = 0; q < ca; ++q) {
        double sacc = (r == q) ? mu : 0.0;
        const double *yr = Yr.data() + (size_t)r * m, *yq = Yr.data() + (size_t)q * m;
        for (int l = 0; l < m; ++l) sacc += yr[l] * yq[l];
        Sc[(size_t)r * ca + q] = sacc;
      }
      if (!chol_lower(Sc.data(), ca)) throw std::runtime_error("Riccati: constraint Schur complement not positive definite");
      // V = Sc^-1 (Y^T W - Bt),  Bt = -[C d 0]
      for (int r = 0; r < ca; ++r) {
        double* vr = Vc.data() + (size_t)r * nr;
        const double* cr = Ct.data() + (size_t)act[r] * nz;
        for (int a = 0; a < nr; ++a) vr[a] = (a < n) ? cr[a] : (a == n ? dt_[act[r]] : 0.0);
        const double* yr = Yr.data() + (size_t)r * m;
        for (int l = 0; l < m; ++l) { const double yl = yr[l]; const double* wl = W.data() + (size_t)l * nr; for (int a = 0; a < nr; ++a) vr[a] += yl * wl[a]; }
      }
      trsm_lower(Sc.data(), ca, Vc.data(), nr); trsm_lower_t(Sc.data(), ca, Vc.data(), nr);
      for (int l = 0; l < m; ++l) {
        for (int a = 0; a < nr; ++a) tmp[a] = 0.0;
        for (int r = 0; r < ca; ++r) { const double yl = Yr[(size_t)r * m + l]; const double* vr = Vc.data() + (size_t)r * nr; for (int a = 0; a < nr; ++a) tmp[a] += yl * vr[a]; }
        double* wl = W.data() + (size_t)l * nr;
        for (int a = 0; a < nr; ++a) wl[a] -= tmp[a];
      }
    }
    trsm_lower_t(Lr.data(), m, W.data(), nr);  // U = L^-T (W - Y V)
    g.Pt.clear(); g.Mu.clear(); g.Znu.clear();
    if (np) {  // inverse stage KKT matrix applied to [I; 0]: Mu = (Ruu + Da^T Da / mu)^-1, Znu = Sc^-1 Y^T L^-1
      g.Pt = Pt;
      std::vector<double> W2((size_t)m * m, 0.0), V2((size_t)(ca > 0 ? ca : 1) * m, 0.0);
      for (int i = 0; i < m; ++i) W2[i * m + i] = 1.0;
      trsm_lower(Lr.data(), m, W2.data(), m);
      if (ca > 0) {
        for (int r = 0; r < ca; ++r) {
          double* vr = V2.data() + (size_t)r * m;
          const double* yr = Yr.data() + (size_t)r * m;
          for (int l = 0; l < m; ++l) { const double yl = yr[l]; const double* wl = W2.data() + (size_t)l * m; for (int a = 0; a < m; ++a) vr[a] += yl * wl[a]; }
        }
        trsm_lower(Sc.data(), ca, V2.data(), m); trsm_lower_t(Sc.data(), ca, V2.data(), m);
        for (int l = 0; l < m; ++l) {
          for (int a = 0; a < m; ++a) tmp[a] = 0.0;
          for (int r = 0; r < ca; ++r) { const double yl = Yr[(size_t)r * m + l]; const double* vr = V2.data() + (size_t)r * m; for (int a = 0; a < m; ++a) tmp[a] += yl * vr[a]; }
          for (int a = 0; a < m; ++a) W2[(size_t)l * m + a] -= tmp[a];
        }
      }
      trsm_lower_t(Lr.data(), m, W2.data(), m);
      g.Mu = W2;
      g.Znu.assign((size_t)c * m, 0.0);
      for (int r = 0; r < ca; ++r) std::memcpy(g.Znu.data() + (size_t)act[r] * m, V2.data() + (size_t)r * m, m * sizeof(double));
    }
    // scatter the active rows back to the row numbering of the stage (inactive rows: zero gains ; their multiplier step is Pi_N(z) / mu = 0)
    std::vector<double> V((size_t)(c > 0 ? c : 1) * nr, 0.0);
    for (int i = 0; i < c; ++i) V[(size_t)i * nr + n] = dt_[i] / mu;
    for (int r = 0; r < ca; ++r) std::memcpy(V.data() + (size_t)act[r] * nr, Vc.data() + (size_t)r * nr, nr * sizeof(double));
    g.K.assign(m * n, 0.0); g.kff.assign(m, 0.0); g.Knu.assign(c * n, 0.0); g.knu.assign(c, 0.0);
    g.Kth.assign((size_t)m * np, 0.0); g.Knuth.assign((size_t)c * np, 0.0);
    for (int i = 0; i < m; ++i) { for (int a = 0; a < n; ++a) g.K[i * n + a] = W[i * nr + a]; g.kff[i] = W[i * nr + n]; for (int j = 0; j < np; ++j) g.Kth[i * np + j] = W[i * nr + n + 1 + j]; }
    for (int i = 0; i < c; ++i) { for (int a = 0; a < n; ++a) g.Knu[i * n + a] = V[i * nr + a]; g.knu[i] = V[i * nr + n]; for (int j = 0; j < np; ++j) g.Knuth[i * np + j] = V[i * nr + n + 1 + j]; }
    ORC_PROF(4);
    // 5. value function  P = Qh + Sh K + C^T Knu ,  p = qh + Sh k + C^T knu
    g.P.assign(n * n, 0.0); g.p.assign(n, 0.0); g.Lm.assign((size_t)n * np, 0.0);
    for (int a = 0; a < n; ++a) {
      double s = gh[a];
      for (int i = 0; i < m; ++i) s += Hh[a * nz + n + i] * g.kff[i];
      for (int r = 0; r < ca; ++r) s += Ct[act[r] * nz + a] * g.knu[act[r]];
      g.p[a] = s;
      for (int b = 0; b < n; ++b) {
        double t = Hh[a * nz + b];
        for (int i = 0; i < m; ++i) t += Hh[a * nz + n + i] * g.K[i * n + b];
        for (int r = 0; r < ca; ++r) t += Ct[act[r] * nz + a] * g.Knu[act[r] * n + b];
        g.P[a * n + b] = t;
      }
      for (int j = 0; j < np; ++j) {
        double t = ghM[a * np + j];
        for (int i = 0; i < m; ++i) t += Hh[a * nz + n + i] * g.Kth[i * np + j];
        for (int r = 0; r < ca; ++r) t += Ct[act[r] * nz + a] * g.Knuth[act[r] * np + j];
        g.Lm[a * np + j] = t;
      }
    }
    for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) { const double s = 0.5 * (g.P[a * n + b] + g.P[b * n + a]); g.P[a * n + b] = g.P[b * n + a] = s; }
    ORC_PROF(5);
    // 6. closed-loop next-state map  x' = T Lam (A x + B u + ft - mud ph)  =  Mx x + mx + Mth theta
    std::vector<double> Acl((size_t)n * nr, 0.0);
    for (int i = 0; i < n; ++i) {
      for (int a = 0; a < n; ++a) { double s = kn.AB[i * nz + a]; for (int l = 0; l < m; ++l) s += kn.AB[i * nz + n + l] * g.K[l * n + a]; Acl[i * nr + a] = s; }
      double s = ft[i] - mud * ph[i];
      for (int l = 0; l < m; ++l) s += kn.AB[i * nz + n + l] * g.kff[l];
      Acl[i * nr + n] = s;
      for (int j = 0; j < np; ++j) {
        double t = -mud * phM[i * np + j];
        for (int l = 0; l < m; ++l) t += kn.AB[i * nz + n + l] * g.Kth[l * np + j];
        Acl[i * nr + n + 1 + j] = t;
      }
    }
    trsm_lower(Lp.data(), n, Acl.data(), nr); trsm_lower_t(Lp.data(), n, Acl.data(), nr);
    if (ff) {
      std::vector<double> top(6 * nr);
      for (int i = 0; i < 6; ++i) for (int a = 0; a < nr; ++a) { double s2 = 0; for (int l = 0; l < 6; ++l) s2 += g.T6[i * 6 + l] * Acl[l * nr + a]; top[i * nr + a] = s2; }
      std::memcpy(Acl.data(), top.data(), 6 * nr * sizeof(double));
    }
    g.Mx.assign(n * n, 0.0); g.mx.assign(n, 0.0); g.Mth.assign((size_t)n * np, 0.0);
    for (int i = 0; i < n; ++i) { for (int a = 0; a < n; ++a) g.Mx[i * n + a] = Acl[i * nr + a]; g.mx[i] = Acl[i * nr + n]; for (int j = 0; j < np; ++j) g.Mth[i * np + j] = Acl[i * nr + n + 1 + j]; }
    ORC_PROF(6);
    // 7. condensed leg: dx_cut = Lm^T dx + Sg theta + sg
    g.Sg.clear(); g.sg.clear(); g.mx0.clear(); g.p0.clear(); g.kff0.clear();
    if (np) {
      g.mx0 = g.mx; g.p0 = g.p; g.kff0 = g.kff;
      g.Sg = *Sgn; g.sg = *sgn;
      for (int l = 0; l < n; ++l)
        for (int a = 0; a < np; ++a) {
          const double lv = (*Lmn)[l * np + a];
          if (lv == 0.0) continue;
          g.sg[a] += lv * g.mx[l];
          for (int b = 0; b < np; ++b) g.Sg[a * np + b] += lv * g.Mth[l * np + b];
        }
      for (int a = 0; a < np; ++a) for (int b = a + 1; b < np; ++b) { const double s = 0.5 * (g.Sg[a * np + b] + g.Sg[b * np + a]); g.Sg[a * np + b] = g.Sg[b * np + a] = s; }
    }
    ORC_PROF(7);
  }

  void backward(Instance& in) const {
    const int N = dims.horizon;
    backward_terminal(in);
    for (int k = N - 1; k >= 0; --k) knot_backward(in, k, in.gains[k + 1].P, in.gains[k + 1].p, nullptr, nullptr, nullptr, in.gains[k]);
  }

  // ---- parallel-in-time Riccati: linear_solver_choice = LQ_SOLVER_PARALLEL + setNumThreads (fulldynamic_talos.py:383-385) ----
  // The horizon is cut into `legs` legs.  Leg j < legs - 1 runs the recursion from a zero value function at its end with the
  // co-state theta_{j+1} of the cut state as a parameter (Jallet et al., "Parallel and proximal constrained LQ", 2024);
  // all legs are independent of one another.  A serial pass over the cuts (consensus) then fixes the cut states and co-states,
  // and the forward sweeps of the legs are independent again.  Same KKT system as the serial sweep: identical results up to
  // round-off.
  int nlegs() const { int L = opt.riccati_legs; if (L > 32) L = 32; if (L > dims.horizon) L = dims.horizon; if (L < 1) L = 1; return L; }  // 32 = MPC_MAX_LEGS of csrc/layout.h
  int leg_start(int j) const { return (int)((long long)j * dims.horizon / nlegs()); }  // first knot of leg j (leg nlegs()-1 ends with the terminal knot)

  // consensus data of leg j < legs-1: cut state dx_{j+1} = Zx dx_j + zc, co-state theta_{j+1} = calP dx_{j+1} + calp
  // ((calP, calp) = exact value function at the start of leg j+1)
  struct LegLink { std::vector<double> Zx, zc, calP, calp; };

  // a parametric knot and the exact value function (calP, calp) at the end of its leg:
  // x_cut = Lm^T x + Sg theta + sg, theta = calP x_cut + calp  ->  x_cut = (I - Sg calP)^-1 (Lm^T x + Sg calp + sg) = Zx x + zc
  void leg_link(const Gains& g, const std::vector<double>& calP, const std::vector<double>& calp, std::vector<double>& Zx, std::vector<double>* zc) const {
    const int n = dims.ndx;
    std::vector<double> Mt((size_t)n * n), R((size_t)n * (n + 1));
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int l = 0; l < n; ++l) s -= g.Sg[i * n + l] * calP[l * n + j]; Mt[i * n + j] = s; }
      for (int j = 0; j < n; ++j) R[i * (n + 1) + j] = g.Lm[j * n + i];
      double s = g.sg[i]; for (int l = 0; l < n; ++l) s += g.Sg[i * n + l] * calp[l];
      R[i * (n + 1) + n] = s;
    }
    solve_dense(Mt, n, R, n + 1);
    Zx.assign((size_t)n * n, 0.0);
    if (zc) zc->assign(n, 0.0);
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) Zx[i * n + j] = R[i * (n + 1) + j];
      if (zc) (*zc)[i] = R[i * (n + 1) + n];
    }
  }

  // ptil[j] (n x n, or empty = zero): guess of the value-function Hessian at the START of leg j (j >= 1), used as the quadratic
  // terminal cost of leg j - 1; the consensus then only solves for the correction calP - ptil.  calP_out[j] = exact Hessian found.
  void backward_legs(Instance& in, std::vector<LegLink>& links, const std::vector<std::vector<double>>& ptil, std::vector<std::vector<double>>& calP_out) const {
    const int N = dims.horizon, n = dims.ndx, J = nlegs();
    calP_out.assign(J, {});
    backward_terminal(in);
    std::vector<double> zeroP((size_t)n * n, 0.0), zerop(n, 0.0), eye((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) eye[i * n + i] = 1.0;
    // the legs are independent of one another
    std::vector<std::string> leg_err(J);
#pragma omp parallel for schedule(dynamic) num_threads(opt.num_threads > 0 ? opt.num_threads : 1)
    for (int j = 0; j < J; ++j) {
      // an exception must not leave the parallel region (it would end the process): per-leg message, rethrown after the region
      try {
        const int s = leg_start(j), e = (j + 1 < J) ? leg_start(j + 1) - 1 : N - 1;
        for (int k = e; k >= s; --k) {
          if (j + 1 == J) knot_backward(in, k, in.gains[k + 1].P, in.gains[k + 1].p, nullptr, nullptr, nullptr, in.gains[k]);
          else if (k == e) knot_backward(in, k, (j + 1 < (int)ptil.size() && !ptil[j + 1].empty()) ? ptil[j + 1] : zeroP, zerop, &eye, &zeroP, &zerop, in.gains[k]);
          else knot_backward(in, k, in.gains[k + 1].P, in.gains[k + 1].p, &in.gains[k + 1].Lm, &in.gains[k + 1].Sg, &in.gains[k + 1].sg, in.gains[k]);
        }
      } catch (const std::exception& ex) {
        leg_err[j] = ex.what();
      }
    }
    for (const std::string& m : leg_err) if (!m.empty()) throw std::runtime_error(m);
    // consensus over the cuts, last to first: true value function (calP, calp) at the start of each leg
    links.assign(J, LegLink());
    std::vector<double> calP = in.gains[leg_start(J - 1)].P, calp = in.gains[leg_start(J - 1)].p;
    for (int j = J - 2; j >= 0; --j) {
      const int s = leg_start(j), e = leg_start(j + 1) - 1;
      calP_out[j + 1] = calP;
      // the leg carries ptil[j + 1] itself: the consensus works on the difference (theta = (calP - ptil) x_cut + calp)
      if (j + 1 < (int)ptil.size() && !ptil[j + 1].empty()) for (size_t i = 0; i < calP.size(); ++i) calP[i] -= ptil[j + 1][i];
      // exact feedback gains of every knot of the leg (checker only: the product corrects knot 0): K + Kth dtheta/dx, dtheta/dx = calP Zx
      for (int k = s; k <= e; ++k) {
        Gains& g = in.gains[k];
        std::vector<double> Zk, Dk((size_t)n * n, 0.0);
        leg_link(g, calP, calp, Zk, nullptr);
        for (int i = 0; i < n; ++i) for (int a = 0; a < n; ++a) { double t = 0; for (int l = 0; l < n; ++l) t += calP[i * n + l] * Zk[l * n + a]; Dk[i * n + a] = t; }
        const int m = in.knots[k].m;
        g.Kexact = g.K;
        for (int i = 0; i < m; ++i) for (int a = 0; a < n; ++a) { double t = 0; for (int l = 0; l < n; ++l) t += g.Kth[i * n + l] * Dk[l * n + a]; g.Kexact[i * n + a] += t; }
      }
      const Gains& g = in.gains[s];
      LegLink& lk = links[j];
      leg_link(g, calP, calp, lk.Zx, &lk.zc);
      lk.calP = calP; lk.calp = calp;
      // value function at the start of leg j: theta = D x + e with D = calP Zx, e = calP zc + calp
      std::vector<double> D((size_t)n * n), ev(n), nP(g.P), np_(g.p);
      for (int i = 0; i < n; ++i) {
        for (int a = 0; a < n; ++a) { double t = 0; for (int l = 0; l < n; ++l) t += calP[i * n + l] * lk.Zx[l * n + a]; D[i * n + a] = t; }
        double t = calp[i]; for (int l = 0; l < n; ++l) t += calP[i * n + l] * lk.zc[l];
        ev[i] = t;
      }
      for (int a = 0; a < n; ++a) {
        for (int b = 0; b < n; ++b) { double t = 0; for (int l = 0; l < n; ++l) t += g.Lm[a * n + l] * D[l * n + b]; nP[a * n + b] += t; }
        double t = 0; for (int l = 0; l < n; ++l) t += g.Lm[a * n + l] * ev[l];
        np_[a] += t;
      }
      for (int a = 0; a < n; ++a) for (int b = a + 1; b < n; ++b) { const double t = 0.5 * (nP[a * n + b] + nP[b * n + a]); nP[a * n + b] = nP[b * n + a] = t; }
      calP.swap(nP); calp.swap(np_);
    }
    for (int k = leg_start(J - 1); k < N; ++k) in.gains[k].Kexact = in.gains[k].K;
    // controlFeedbacks()[0] is what the scripts read (fulldynamic_talos.py:522): knot 0 carries the exact gain
    if (J > 1) in.gains[0].K = in.gains[0].Kexact;
  }

  // ---- tree over the cuts (csrc/legs_tree.h, three legs or more unless MPC_LEGS_CHAIN=1): the chain of backward_legs costs one solve per cut, one after the other;
  // composing the condensed forms of adjacent (groups of) legs pairwise takes ceil(log2 J) rounds of independent solves.  Same KKT
  // system: identical steps up to round-off.  The guess of the value-function Hessian at a cut is the Hessian of the node that
  // starts there, given ITS end guess (exact for the nodes that hold the last leg; the others catch up one tree level per pass).
  bool use_tree() const { const char* e = std::getenv("MPC_LEGS_CHAIN"); return nlegs() >= 3 && !(e && std::atoi(e) > 0); }  // (the rule of csrc/mpc_hip.hip)
  static void mm(const std::vector<double>& A, const std::vector<double>& B, std::vector<double>& C, int n, bool ta = false) {  // C = A B (or A^T B)
    C.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) for (int l = 0; l < n; ++l) { const double a = ta ? A[l * n + i] : A[i * n + l]; if (a == 0.0) continue; for (int j = 0; j < n; ++j) C[i * n + j] += a * B[l * n + j]; }
  }
  static void mv(const std::vector<double>& A, const std::vector<double>& x, std::vector<double>& y, int n, bool ta = false) {
    y.assign(n, 0.0);
    for (int i = 0; i < n; ++i) for (int l = 0; l < n; ++l) y[i] += (ta ? A[l * n + i] : A[i * n + l]) * x[l];
  }
  // a followed by b; pg = the Hessian the last leg of a carried as its terminal cost (or null = zero)
  void tree_compose(const TreeNode& a, const TreeNode& b, const std::vector<double>* pg, TreeNode& ab) const {
    const int n = dims.ndx;
    ab.lo = a.lo; ab.hi = b.hi;
    ab.D = b.P;
    if (pg && !pg->empty()) for (size_t i = 0; i < ab.D.size(); ++i) ab.D[i] -= (*pg)[i];
    // W [Lm_a^T | Sg_a | Sg_a p_b + sg_a],  W = (I - Sg_a D)^-1
    std::vector<double> Mt, R((size_t)n * (2 * n + 1)), t;
    mm(a.Sg, ab.D, Mt, n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Mt[i * n + j] = ((i == j) ? 1.0 : 0.0) - Mt[i * n + j];
    mv(a.Sg, b.p, t, n);
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < n; ++j) { R[i * (2 * n + 1) + j] = a.Lm[j * n + i]; R[i * (2 * n + 1) + n + j] = a.Sg[i * n + j]; }
      R[i * (2 * n + 1) + 2 * n] = t[i] + a.sg[i];
    }
    solve_dense(Mt, n, R, 2 * n + 1);
    std::vector<double> T1((size_t)n * n), T2((size_t)n * n), t3(n);
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) { T1[i * n + j] = R[i * (2 * n + 1) + j]; T2[i * n + j] = R[i * (2 * n + 1) + n + j]; } t3[i] = R[i * (2 * n + 1) + 2 * n]; }
    ab.Zx = T1; ab.zc = t3;
    mm(T2, b.Lm, ab.Zt, n);                                   // Zt = W Sg_a Lm_b
    std::vector<double> X, Y, u, w;
    mm(b.Lm, ab.Zt, X, n, true);                              // Sg_ab = Sg_b + Lm_b^T Zt
    ab.Sg = b.Sg; for (size_t i = 0; i < X.size(); ++i) ab.Sg[i] += X[i];
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { const double v = 0.5 * (ab.Sg[i * n + j] + ab.Sg[j * n + i]); ab.Sg[i * n + j] = ab.Sg[j * n + i] = v; }
    mv(b.Lm, t3, u, n, true);                                 // sg_ab = sg_b + Lm_b^T t3
    ab.sg = b.sg; for (int i = 0; i < n; ++i) ab.sg[i] += u[i];
    mm(T1, b.Lm, ab.Lm, n, true);                             // Lm_ab = T1^T Lm_b  (= Lm_a (I - D Sg_a)^-1 Lm_b)
    mm(ab.D, T1, X, n); mm(a.Lm, X, Y, n);                    // P_ab = P_a + Lm_a D T1
    ab.P = a.P; for (size_t i = 0; i < Y.size(); ++i) ab.P[i] += Y[i];
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { const double v = 0.5 * (ab.P[i * n + j] + ab.P[j * n + i]); ab.P[i * n + j] = ab.P[j * n + i] = v; }
    mv(ab.D, t3, u, n); for (int i = 0; i < n; ++i) u[i] += b.p[i];  // p_ab = p_a + Lm_a (D t3 + p_b)
    mv(a.Lm, u, w, n);
    ab.p = a.p; for (int i = 0; i < n; ++i) ab.p[i] += w[i];
  }
  // the sweeps of the legs (as backward_legs), then the tree; the new guesses go to in.tree_guess
  void backward_legs_tree(Instance& in) const {
    const int N = dims.horizon, n = dims.ndx, J = nlegs();
    backward_terminal(in);
    std::vector<double> zeroP((size_t)n * n, 0.0), zerop(n, 0.0), eye((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) eye[i * n + i] = 1.0;
    const std::vector<std::vector<double>>& ptil = in.tree_guess;
    std::vector<std::string> leg_err(J);
#pragma omp parallel for schedule(dynamic) num_threads(opt.num_threads > 0 ? opt.num_threads : 1)
    for (int j = 0; j < J; ++j) {
      // an exception must not leave the parallel region (it would end the process): per-leg message, rethrown after the region
      try {
        const int s = leg_start(j), e = (j + 1 < J) ? leg_start(j + 1) - 1 : N - 1;
        for (int k = e; k >= s; --k) {
          if (j + 1 == J) knot_backward(in, k, in.gains[k + 1].P, in.gains[k + 1].p, nullptr, nullptr, nullptr, in.gains[k]);
          else if (k == e) knot_backward(in, k, (j + 1 < (int)ptil.size() && !ptil[j + 1].empty()) ? ptil[j + 1] : zeroP, zerop, &eye, &zeroP, &zerop, in.gains[k]);
          else knot_backward(in, k, in.gains[k + 1].P, in.gains[k + 1].p, &in.gains[k + 1].Lm, &in.gains[k + 1].Sg, &in.gains[k + 1].sg, in.gains[k]);
        }
      } catch (const std::exception& ex) {
        leg_err[j] = ex.what();
      }
    }
    for (const std::string& m : leg_err) if (!m.empty()) throw std::runtime_error(m);
    std::vector<TreeNode>& T = in.tree;
    T.clear();
    std::vector<int> level;
    for (int j = 0; j < J; ++j) {
      const Gains& g = in.gains[leg_start(j)];
      TreeNode nd; nd.lo = nd.hi = j; nd.P = g.P; nd.p = g.p;
      if (j + 1 < J) { nd.Lm = g.Lm; nd.Sg = g.Sg; nd.sg = g.sg; } else { nd.Lm = zeroP; nd.Sg = zeroP; nd.sg = zerop; }
      T.push_back(nd); level.push_back(j);
    }
    while (level.size() > 1) {
      std::vector<int> next;
      for (size_t i = 0; i + 1 < level.size(); i += 2) {
        TreeNode ab;
        const int cut = T[level[i + 1]].lo;
        tree_compose(T[level[i]], T[level[i + 1]], cut < (int)ptil.size() ? &ptil[cut] : nullptr, ab);
        ab.left = level[i]; ab.right = level[i + 1];
        T.push_back(ab); next.push_back((int)T.size() - 1);
      }
      if (level.size() % 2) next.push_back(level.back());
      level.swap(next);
    }
    // guesses of the next pass: the Hessian of the node that starts at the cut
    std::vector<std::vector<double>> ng(J);
    for (const TreeNode& nd : T) if (nd.right >= 0) ng[T[nd.right].lo] = T[nd.right].P;
    in.tree_guess.swap(ng);
    for (int k = 0; k < N; ++k) in.gains[k].Kexact = in.gains[k].K;
  }
  // cut states and co-state parameters by the down-sweep, the exact gain of knot 0 along the leftmost path; then as forward_legs
  void forward_legs_tree(Instance& in) const {
    const int n = dims.ndx, J = nlegs();
    std::fill(in.dxs[0].begin(), in.dxs[0].end(), 0.0);
    std::fill(in.dlams[0].begin(), in.dlams[0].end(), 0.0);
    std::vector<std::vector<double>> ths(J, std::vector<double>(n, 0.0));
    const std::vector<TreeNode>& T = in.tree;
    struct Item { int node; std::vector<double> xin, thout; };
    std::vector<Item> stack;
    stack.push_back({(int)T.size() - 1, in.dxs[0], std::vector<double>(n, 0.0)});
    while (!stack.empty()) {
      Item it = stack.back(); stack.pop_back();
      const TreeNode& nd = T[it.node];
      if (nd.right < 0) continue;
      const TreeNode& b = T[nd.right];
      std::vector<double> xm(nd.zc), thm(b.p), t;
      mv(nd.Zx, it.xin, t, n); for (int i = 0; i < n; ++i) xm[i] += t[i];
      mv(nd.Zt, it.thout, t, n); for (int i = 0; i < n; ++i) xm[i] += t[i];
      mv(nd.D, xm, t, n); for (int i = 0; i < n; ++i) thm[i] += t[i];
      mv(b.Lm, it.thout, t, n); for (int i = 0; i < n; ++i) thm[i] += t[i];
      in.dxs[leg_start(b.lo)] = xm; ths[b.lo - 1] = thm;
      stack.push_back({nd.left, it.xin, thm});
      stack.push_back({nd.right, xm, it.thout});
    }
    // exact K_0 = K_0 + Kth_0 d theta_1 / d x_0: sensitivities along the leftmost path (theta_out of the root is absent)
    {
      std::vector<double> S((size_t)n * n, 0.0), X, Y, Z;  // S = d theta_out / d x_0 of the current node
      int node = (int)T.size() - 1;
      while (T[node].right >= 0) {
        const TreeNode& nd = T[node];
        const TreeNode& b = T[nd.right];
        mm(nd.Zt, S, X, n); for (size_t i = 0; i < X.size(); ++i) X[i] += nd.Zx[i];     // d x_mid / d x_0
        mm(nd.D, X, Y, n); mm(b.Lm, S, Z, n); for (size_t i = 0; i < Y.size(); ++i) Y[i] += Z[i];  // d theta_mid / d x_0
        S.swap(Y);
        node = nd.left;
      }
      Gains& g0 = in.gains[0];
      const int m = in.knots[0].m;
      g0.Kexact = g0.K;
      for (int i = 0; i < m; ++i) for (int a = 0; a < n; ++a) { double t = 0; for (int l = 0; l < n; ++l) t += g0.Kth[i * n + l] * S[l * n + a]; g0.Kexact[i * n + a] += t; }
      g0.K = g0.Kexact;
    }
    in.leg_Zx.assign(J, {}); in.leg_zc.assign(J, {}); in.leg_calP.assign(J, {}); in.leg_calp.assign(J, {}); in.leg_theta.assign(J, {});
    for (int j = 0; j + 1 < J; ++j) in.leg_theta[j] = ths[j];
    forward_legs_apply(in, ths);
  }

  void forward_legs(Instance& in, const std::vector<LegLink>& links) const {
    const int N = dims.horizon, n = dims.ndx, J = nlegs();
    const bool ff = dims.space == MPC_SPACE_MULTIBODY && model.has_freeflyer();
    std::fill(in.dxs[0].begin(), in.dxs[0].end(), 0.0);  // force_initial_condition
    std::fill(in.dlams[0].begin(), in.dlams[0].end(), 0.0);
    // consensus, first cut to last (csrc/legs.h k_leg_consensus): the cut state and its co-state come from the SAME solve
    // (theta = calP x_cut + calp holds to round-off of that product), so the stationarity condition of the cut state is met as
    // accurately as that of any other knot; what is left of the round-off of the leg's own forward sweep is a dynamics gap of
    // the order of 1e-12 at the cut
    std::vector<std::vector<double>> ths(J, std::vector<double>(n, 0.0));
    in.leg_Zx.assign(J, {}); in.leg_zc.assign(J, {}); in.leg_calP.assign(J, {}); in.leg_calp.assign(J, {}); in.leg_theta.assign(J, {});
    for (int j = 0; j + 1 < J; ++j) {
      const LegLink& lk = links[j];
      const int s = leg_start(j), c = leg_start(j + 1);
      for (int i = 0; i < n; ++i) { double t = lk.zc[i]; for (int a = 0; a < n; ++a) t += lk.Zx[i * n + a] * in.dxs[s][a]; in.dxs[c][i] = t; }
      for (int i = 0; i < n; ++i) { double t = lk.calp[i]; for (int a = 0; a < n; ++a) t += lk.calP[i * n + a] * in.dxs[c][a]; ths[j][i] = t; }
      in.leg_Zx[j] = lk.Zx; in.leg_zc[j] = lk.zc; in.leg_calP[j] = lk.calP; in.leg_calp[j] = lk.calp; in.leg_theta[j] = ths[j];
    }
    forward_legs_apply(in, ths);
  }

  void forward_legs_apply(Instance& in, const std::vector<std::vector<double>>& ths) const {
    const int N = dims.horizon, n = dims.ndx, J = nlegs();
    const bool ff = dims.space == MPC_SPACE_MULTIBODY && model.has_freeflyer();
    // apply (csrc/legs.h k_leg_apply): with theta known, the affine terms of the leg's knots take their final values
    //   p += Lm theta, k += Kth theta, knu += Knuth theta, mx += Mth theta ; the sweeps below are then the plain ones
    for (int j = 0; j + 1 < J; ++j) {
      const std::vector<double>& th = ths[j];
      for (int k = leg_start(j); k < leg_start(j + 1); ++k) {
        Gains& g = in.gains[k];
        const Knot& kn = in.knots[k];
        for (int i = 0; i < n; ++i) { double t = 0, t2 = 0; for (int a = 0; a < n; ++a) { t += g.Lm[i * n + a] * th[a]; t2 += g.Mth[i * n + a] * th[a]; } g.p[i] += t; g.mx[i] += t2; }
        for (int i = 0; i < kn.m; ++i) { double t = 0; for (int a = 0; a < n; ++a) t += g.Kth[i * n + a] * th[a]; g.kff[i] += t; }
        for (int i = 0; i < kn.c; ++i) { double t = 0; for (int a = 0; a < n; ++a) t += g.Knuth[i * n + a] * th[a]; g.knu[i] += t; }
      }
    }
    for (int j = 0; j < J; ++j) {
      const bool par = j + 1 < J;
      const int s = leg_start(j), e = par ? leg_start(j + 1) - 1 : N;
      for (int k = s; k <= e; ++k) {
        const Gains& g = in.gains[k];
        const Knot& kn = in.knots[k];
        const double* dx = in.dxs[k].data();
        for (int i = 0; i < kn.c; ++i) {
          double t = g.knu[i];
          for (int a = 0; a < n; ++a) t += g.Knu[i * n + a] * dx[a];
          in.dvs[k][i] = t - in.vs[k][i];
        }
        for (int i = kn.c; i < dims.nc_max; ++i) in.dvs[k][i] = 0.0;
        if (k == N) break;
        for (int i = 0; i < kn.m; ++i) {
          double t = g.kff[i];
          for (int a = 0; a < n; ++a) t += g.K[i * n + a] * dx[a];  // knot 0 holds the exact gain; dx_0 = 0 (force_initial_condition)
          in.dus[k][i] = t;
        }
        double* dxn = in.dxs[k + 1].data();
        if (!(par && k == e))  // the cut state is the consensus value
          for (int i = 0; i < n; ++i) {
            double t = g.mx[i];
            for (int a = 0; a < n; ++a) t += g.Mx[i * n + a] * dx[a];
            dxn[i] = t;
          }
        // co-state of knot k + 1: T^T (P' dx' + p') with the final p' — at a cut this is theta itself up to round-off (consensus)
        std::vector<double> l(n);
        {
          const Gains& gn = in.gains[k + 1];
          for (int i = 0; i < n; ++i) {
            double t = gn.p[i];
            for (int a = 0; a < n; ++a) t += gn.P[i * n + a] * dxn[a];
            l[i] = t;
          }
        }
        if (ff) { double t[6]; for (int i = 0; i < 6; ++i) { double s2 = 0; for (int a = 0; a < 6; ++a) s2 += g.T6[a * 6 + i] * l[a]; t[i] = s2; } for (int i = 0; i < 6; ++i) l[i] = t[i]; }
        for (int i = 0; i < n; ++i) in.dlams[k + 1][i] = l[i] - in.lams[k + 1][i];
      }
    }
  }

  // P7: forward sweep -> steps (dx, du) and NEW multipliers, converted to increments
  void forward(Instance& in) const {
    const int N = dims.horizon, n = dims.ndx;
    const bool ff = dims.space == MPC_SPACE_MULTIBODY && model.has_freeflyer();
    std::fill(in.dxs[0].begin(), in.dxs[0].end(), 0.0);  // force_initial_condition
    std::fill(in.dlams[0].begin(), in.dlams[0].end(), 0.0);
    for (int k = 0; k <= N; ++k) {
      const Gains& g = in.gains[k];
      const Knot& kn = in.knots[k];
      const double* dx = in.dxs[k].data();
      for (int i = 0; i < kn.c; ++i) { double s = g.knu[i]; for (int a = 0; a < n; ++a) s += g.Knu[i * n + a] * dx[a]; in.dvs[k][i] = s - in.vs[k][i]; }
      for (int i = kn.c; i < dims.nc_max; ++i) in.dvs[k][i] = 0.0;
      if (k == N) break;
      for (int i = 0; i < kn.m; ++i) { double s = g.kff[i]; for (int a = 0; a < n; ++a) s += g.K[i * n + a] * dx[a]; in.dus[k][i] = s; }
      double* dxn = in.dxs[k + 1].data();
      for (int i = 0; i < n; ++i) { double s = g.mx[i]; for (int a = 0; a < n; ++a) s += g.Mx[i * n + a] * dx[a]; dxn[i] = s; }
      const Gains& gn = in.gains[k + 1];
      std::vector<double> l(n);
      for (int i = 0; i < n; ++i) { double s = gn.p[i]; for (int a = 0; a < n; ++a) s += gn.P[i * n + a] * dxn[a]; l[i] = s; }
      if (ff) { double t[6]; for (int i = 0; i < 6; ++i) { double s = 0; for (int a = 0; a < 6; ++a) s += g.T6[a * 6 + i] * l[a]; t[i] = s; } for (int i = 0; i < 6; ++i) l[i] = t[i]; }
      for (int i = 0; i < n; ++i) in.dlams[k + 1][i] = l[i] - in.lams[k + 1][i];
    }
  }

  // directional derivative of the merit along the step
  double dmerit(const Instance& in) const {
    const int N = dims.horizon, n = dims.ndx;
    const double mu = in.mu, mud = mu_dyn(in);
    const bool ff = dims.space == MPC_SPACE_MULTIBODY && model.has_freeflyer();
    double d = 0;
    for (int k = 0; k <= N; ++k) {
      const Knot& kn = in.knots[k];
      const int nz = kn.n + kn.m;
      std::vector<double> dz(nz);
      for (int a = 0; a < kn.n; ++a) dz[a] = in.dxs[k][a];
      for (int a = 0; a < kn.m; ++a) dz[kn.n + a] = in.dus[k][a];
      for (int a = 0; a < nz; ++a) d += kn.grad[a] * dz[a];
      for (int i = 0; i < kn.c; ++i) {
        bool act;
        const double pn = proj_normal(kn.ctype[i], kn.cval[i] + mu * in.vs_e[k][i], kn.lo[i], kn.hi[i], act);
        const double vp = pn / mu, v = in.vs[k][i];
        double jd = 0;
        for (int a = 0; a < nz; ++a) jd += kn.CD[i * nz + a] * dz[a];
        d += (vp + (act ? (vp - v) : 0.0)) * jd - mu * (vp - v) * in.dvs[k][i];
      }
      if (k < N) {
        for (int i = 0; i < n; ++i) {
          const double lp = in.lams_e[k + 1][i] + kn.f[i] / mud, l = in.lams[k + 1][i];
          double jd = 0;
          for (int a = 0; a < nz; ++a) jd += kn.AB[i * nz + a] * dz[a];
          if (ff && i < 6) { for (int a = 0; a < 6; ++a) jd += kn.E6[i * 6 + a] * in.dxs[k + 1][a]; }
          else jd -= in.dxs[k + 1][i];
          d += (2 * lp - l) * jd - mud * (lp - l) * in.dlams[k + 1][i];
        }
      }
    }
    return d;
  }

  void make_trial(Instance& in, double alpha) const {
    const int N = dims.horizon;
    std::vector<double> sd(dims.ndx);
    for (int k = 0; k <= N; ++k) {
      for (int i = 0; i < dims.ndx; ++i) sd[i] = alpha * in.dxs[k][i];
      integrate(in.xs[k].data(), sd.data(), in.txs[k].data());
      for (int i = 0; i < dims.nc_max; ++i) in.tvs[k][i] = in.vs[k][i] + alpha * in.dvs[k][i];
      for (int i = 0; i < dims.ndx; ++i) in.tlams[k][i] = in.lams[k][i] + alpha * in.dlams[k][i];
      if (k < N) for (int i = 0; i < dims.nu; ++i) in.tus[k][i] = in.us[k][i] + alpha * in.dus[k][i];
    }
  }

  void update_tols_on_failure(Instance& in) const {
    in.prim_tol = opt.prim_tol0 * std::pow(in.mu, opt.bcl_prim_alpha);
    in.inner_tol = opt.inner_tol0 * std::pow(in.mu, opt.bcl_dual_alpha);
  }
  void update_tols_on_success(Instance& in) const {
    in.prim_tol *= std::pow(in.mu, opt.bcl_prim_beta);
    in.inner_tol *= std::pow(in.mu, opt.bcl_dual_beta);
  }

  void setup() {
    for (auto& in : inst) {
      in.mu = opt.mu_init;
      for (auto& v : in.vs) std::fill(v.begin(), v.end(), 0.0);
      for (auto& v : in.lams) std::fill(v.begin(), v.end(), 0.0);
      for (auto& v : in.vs_e) std::fill(v.begin(), v.end(), 0.0);
      for (auto& v : in.lams_e) std::fill(v.begin(), v.end(), 0.0);
      in.stats = mpc_stats{};
    }
  }

  // one inner iteration; returns 0 after a step, 1 if the inner criterion was already met, 2 if no descent is left (no step)
  int iterate(Instance& in) {
    evaluate(in, in.xs, in.us, in.knots, true);
    double cost, prim, dual, crit;
    const double phi0 = merit(in, in.knots, in.vs, in.lams, &cost, &prim);
    lagrangian_residuals(in, dual, crit);
    in.stats.traj_cost = cost; in.stats.merit = phi0; in.stats.prim_infeas = prim; in.stats.dual_infeas = dual; in.stats.mu = in.mu;
    if (std::getenv("MPC_ORACLE_DEBUG")) fprintf(stderr, "it %d: cost %.10e merit %.10e prim %.3e dual %.3e crit %.3e (inner_tol %.3e) mu %.1e\n", in.stats.num_iters, cost, phi0, prim, dual, crit, in.inner_tol, in.mu);
    if (crit <= in.inner_tol) return 1;
    if (nlegs() > 1) {
      std::vector<LegLink> links;
      std::vector<std::vector<double>> ptil, calP;
      // Two sweeps: the first from a zero value function at the leg ends, the second with the Hessians the consensus of the first
      // found at the cuts as terminal costs of the legs.  The consensus then only solves for a small correction: with a zero
      // guess I - Sg calP mixes the stiffest directions of calP (constraint penalties, 1/mu) with the most controllable ones of
      // the leg and the cut states lose up to ten digits (complete Talos model, far from the solution).  The HIP library takes
      // the guess from its previous pass / MPC tick instead (csrc/legs.h).  MPC_LEGS_PLAIN=1: one sweep from zero (both
      // libraries: the intermediates of the leg kernels are then comparable one to one).
      const char* ep = std::getenv("MPC_LEGS_PLAIN");
      const int passes = (ep && std::atoi(ep) > 0) ? 1 : 2;
      if (use_tree()) {
        // the guesses live from pass to pass (as in the HIP library); a pass without any sweeps once per level of the tree so that
        // every cut has seen a guess derived from the true terminal cost
        int depth = 0; for (int w = 1; w < nlegs(); w *= 2) ++depth;
        const int sweeps = (ep && std::atoi(ep) > 0) ? 1 : (in.tree_guess_valid ? 1 : depth + 1);
        if (ep && std::atoi(ep) > 0) in.tree_guess.clear();
        for (int pass = 0; pass < sweeps; ++pass) backward_legs_tree(in);
        in.tree_guess_valid = true;
        forward_legs_tree(in);
      } else {
      for (int pass = 0; pass < passes; ++pass) { backward_legs(in, links, ptil, calP); ptil = calP; }
      forward_legs(in, links);
      }
    } else {
      backward(in);
      forward(in);
    }
    const double dphi0 = dmerit(in);
    // no descent left in the inner problem (round-off floor of the 1/mu-conditioned system): counts as solved, no step
    // (MPC_STALL_TOL of csrc/solver_kernels.h)
    if (std::fabs(dphi0) <= 1e-13 * (1.0 + std::fabs(phi0))) return 2;
    double alpha = 1.0, phi = 0.0;
    int step = 0;
    for (;; ++step) {
      make_trial(in, alpha);
      evaluate(in, in.txs, in.tus, in.tknots, false);
      phi = merit(in, in.tknots, in.tvs, in.tlams);
      if (std::getenv("MPC_ORACLE_DEBUG")) fprintf(stderr, "  ls: alpha %.4g phi %.10e phi0 %.10e dphi0 %.4e\n", alpha, phi, phi0, dphi0);
      if (phi <= phi0 + opt.ls_armijo_c1 * alpha * dphi0) break;
      if (step + 1 >= opt.ls_max_steps || 0.5 * alpha < opt.ls_alpha_min) break;
      alpha *= 0.5;
    }
    in.xs.swap(in.txs); in.us.swap(in.tus); in.vs.swap(in.tvs); in.lams.swap(in.tlams);
    in.stats.alpha = alpha; in.stats.ls_steps = step; in.stats.num_iters += 1;
    return 0;
  }

  // N2: closed-loop simulation stand-in — knot 0's contact dynamics integrated `substeps` times with step
  // `dt` under the feedback law of the low-level loop u = us[0] - K0 difference(x, xs[0]) (fulldynamic_talos.py:512-530);
  // the result becomes the measured state x0 of the next tick.
  void simulate(Instance& in, int substeps, double dt, const double* push = nullptr) const {  // push: world-frame force at the base origin (3) or null
    if (dims.space != MPC_SPACE_MULTIBODY || stages[0].dyn != MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER)
      throw std::runtime_error("simulate: only contact-constrained whole-body dynamics are supported");
    const int n = dims.ndx, nu = dims.nu, nx = dims.nx;
    StageDesc sd = stages[0];
    sd.params[sd.dyn_poff] = dt;  // the stage's time step is the first dynamics parameter
    sd.terms.clear(); sd.nc = 0;  // dynamics only
    std::vector<double> x = in.xs[0], d(n), u(nu);
    Knot kn;
    kn.resize(n, nu, 0, nx);
    for (int sstep = 0; sstep < substeps; ++sstep) {
      mb_difference(model, x.data(), in.xs[0].data(), d.data());
      for (int i = 0; i < nu; ++i) {
        double su = in.us[0][i];
        for (int j = 0; j < n; ++j) su -= in.gains[0].K[i * n + j] * d[j];
        u[i] = su;
      }
      std::vector<double> tau;
      if (push) {  // generalized force of a world-frame force f at the base origin: R^T f on the linear base dofs
        tau.assign(model.nv, 0.0);
        const State<double> sx = state_from_x(model, x.data());
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tau[i] += sx.base.R(j, i) * push[j];
        ext_tau() = tau.data();
      }
      ORC_EVAL_MULTIBODY(model, sd, nu, x.data(), u.data(), x.data(), kn, false);
      ext_tau() = nullptr;
      x = kn.xnext;
    }
    in.x0 = x;
  }

  // mpc_options.refine_appended_knot (include/mpc_abi.h): the knot mpc_cycle appended starts from a control consistent with ITS stage —
  // R Newton steps on u_{N-1} alone (x_{N-1} fixed) on the knot's own penalty problem, then x_N = phi(x_{N-1}, u_{N-1}).  Each step is
  // the stage KKT system of the knot with the controls eliminated first (as in knot_backward):
  //     Huu = L L^T ;  Y = L^-1 D_a^T ;  S = Y^T Y + rho I ;  nu = S^-1 (Pi_N(z)_a - Y^T L^-1 g_u) ;  du = -L^-T (L^-1 g_u + Y nu)
  // with rho = max(mu, 1e-8 max diag(Y^T Y)): the active rows of a wrench cone are linearly dependent (more than six rows on a 6-D
  // wrench), and with rho = mu = 1e-8 the step is only determined to ~1e-4 (cond(S) ~ 1e12: two libraries, two answers); the relative
  // floor makes it reproducible to ~1e-8 at the price of an inexact restoration along directions that D_a^T annihilates anyway.
  void refine_appended_knot(Instance& in) {
    const int N = dims.horizon, n = dims.ndx;
    const int b_inst = (int)(&in - inst.data());
    if (N < 1 || in.mu <= 0.0) return;
    Knot kn;
    const int R = opt.refine_appended_knot < 0 ? -opt.refine_appended_knot : opt.refine_appended_knot;
    for (int it = 0; it < R; ++it) {
      eval_knot(b_inst, N - 1, in.xs[N - 1].data(), in.us[N - 1].data(), in.xs[N].data(), kn, true);
      const int m = kn.m, nz = n + m;
      // an abandoned step leaves the control as it is; the remaining steps and the final x_N = phi(x_{N-1}, u_{N-1}) still run (as the launches of
      // csrc/mpc_hip.hip launch_refine do: k_refine_knot returns from ITS step only)
      if (m <= 0 || m > 48) continue;
      std::vector<double> Lr((size_t)m * m), w(m);
      for (int i = 0; i < m; ++i) {
        w[i] = kn.grad[n + i];
        for (int j = 0; j < m; ++j) Lr[(size_t)i * m + j] = 0.5 * (kn.H[(size_t)(n + i) * nz + n + j] + kn.H[(size_t)(n + j) * nz + n + i]);
      }
      bool bad = false;
      std::vector<int> act;
      std::vector<double> v;
      for (int r = 0; r < kn.c; ++r) {
        bool a_;
        const double pn = proj_normal(kn.ctype[r], kn.cval[r] + in.mu * in.vs_e[N - 1][r], kn.lo[r], kn.hi[r], a_);
        if (a_) { act.push_back(r); v.push_back(pn); }
      }
      const int ca = (int)act.size();
      if (ca > 48) continue;  // (the HIP kernel's LDS carve-out)
      if (!chol_lower(Lr.data(), m)) continue;  // (an indefinite knot Hessian: this step leaves the control as it is)
      trsm_lower(Lr.data(), m, w.data(), 1);
      std::vector<double> Y((size_t)m * std::max(ca, 1)), nu(std::max(ca, 1), 0.0);
      if (ca > 0) {
        for (int i = 0; i < m; ++i) for (int q = 0; q < ca; ++q) Y[(size_t)i * ca + q] = kn.CD[(size_t)act[q] * nz + n + i];
        trsm_lower(Lr.data(), m, Y.data(), ca);
        std::vector<double> S((size_t)ca * ca);
        double dmax = 0.0;
        for (int p = 0; p < ca; ++p) for (int q = 0; q < ca; ++q) { double t = 0; for (int i = 0; i < m; ++i) t += Y[(size_t)i * ca + p] * Y[(size_t)i * ca + q]; S[(size_t)p * ca + q] = t; if (p == q) dmax = std::max(dmax, t); }
        const double rho = std::max(in.mu, 1e-8 * dmax);
        for (int q = 0; q < ca; ++q) {
          S[(size_t)q * ca + q] += rho;
          double t = v[q];
          for (int i = 0; i < m; ++i) t -= Y[(size_t)i * ca + q] * w[i];
          nu[q] = t;
        }
        if (!chol_lower(S.data(), ca)) { bad = true; }
        if (!bad) {
        trsm_lower(S.data(), ca, nu.data(), 1); trsm_lower_t(S.data(), ca, nu.data(), 1);
        for (int i = 0; i < m; ++i) { double t = w[i]; for (int q = 0; q < ca; ++q) t += Y[(size_t)i * ca + q] * nu[q]; w[i] = t; }
        }
      }
      if (bad) continue;
      trsm_lower_t(Lr.data(), m, w.data(), 1);
      for (int i = 0; i < m; ++i) if (!std::isfinite(w[i])) bad = true;
      if (bad) continue;
      for (int i = 0; i < m; ++i) in.us[N - 1][i] -= w[i];
    }
    eval_knot(b_inst, N - 1, in.xs[N - 1].data(), in.us[N - 1].data(), in.xs[N].data(), kn, false);
    in.xs[N] = kn.xnext;
  }

  // Torque-driven form of the stand-in (bullet_robot.py:138-145 execute + stepSimulation): knot 0's contact dynamics under the given joint
  // torques, from `x` (null: from the measured state); wrench (12, may be null) = contact wrenches of the last sub-step.
  void simulate_torque(Instance& in, const double* x_start, const double* tau, int substeps, double dt, double* wrench) const {
    if (dims.space != MPC_SPACE_MULTIBODY || stages[0].dyn != MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER)
      throw std::runtime_error("simulate: only contact-constrained whole-body dynamics are supported");
    const int n = dims.ndx, nu = dims.nu, nx = dims.nx;
    StageDesc sd = stages[0];
    sd.params[sd.dyn_poff] = dt;
    sd.terms.clear(); sd.nc = 0;
    std::vector<double> x = x_start ? std::vector<double>(x_start, x_start + nx) : in.x0;
    Knot kn;
    kn.resize(n, nu, 0, nx);
    for (int sstep = 0; sstep < substeps; ++sstep) {
      ORC_EVAL_MULTIBODY(model, sd, nu, x.data(), tau, x.data(), kn, false);
      x = kn.xnext;
    }
    if (wrench) std::memcpy(wrench, kn.wrench, 12 * sizeof(double));
    in.x0 = x;
  }

  // SolverProxDDP::run for one instance (xs/us already installed)
  void run_instance(Instance& in) {
    if (opt.force_initial_condition) in.xs[0] = in.x0;
    in.stats.num_iters = 0; in.stats.converged = 0; in.stats.al_iters = 0;
    update_tols_on_failure(in);
    in.inner_tol = std::max(in.inner_tol, opt.tol); in.prim_tol = std::max(in.prim_tol, opt.tol);
    int stalls = 0;
    // mpc_options.corrector_prim_tol: when the iteration budget ends with an iteration that STARTED from an iterate infeasible by more than the
    // tolerance (stats.prim_infeas is measured before the step) or whose step had to be shortened (alpha < 1), the instance takes one more
    // iteration — once per run (k_after_step of
    // csrc/solver_kernels.h: the same rule)
    int max_it = opt.max_iters;
    bool corrected = false;
    while (in.stats.al_iters < opt.max_al_iters && in.stats.num_iters < max_it) {
      bool inner_conv = false, via_stall = false;
      while (in.stats.num_iters < max_it) {
        const int r = iterate(in);
        if (r == 0 && !corrected && corrector_armed && opt.corrector_prim_tol > 0.0 && in.stats.num_iters >= opt.max_iters && (in.stats.prim_infeas > opt.corrector_prim_tol || in.stats.alpha < 1.0)) { corrected = true; ++max_it; }
        if (r == 0) { stalls = 0; continue; }
        inner_conv = true;
        if (r == 2) { via_stall = true; ++stalls; }
        break;
      }
      if (!inner_conv) break;
      if (in.stats.prim_infeas <= in.prim_tol) {
        update_tols_on_success(in);
        in.vs_e = in.vs; in.lams_e = in.lams;
        if (std::max(in.stats.prim_infeas, in.stats.dual_infeas) <= opt.tol) { in.stats.converged = 1; break; }
      } else {
        in.mu = std::max(in.mu * opt.bcl_mu_update_factor, opt.bcl_mu_lower_bound);
        update_tols_on_failure(in);
      }
      in.inner_tol = std::max(in.inner_tol, opt.tol); in.prim_tol = std::max(in.prim_tol, opt.tol);
      in.stats.al_iters += 1;
      if (via_stall && stalls >= 4) break;  // four stalls with no step in between (two full BCL cycles): nothing left to gain
    }
    in.stats.mu = in.mu;
  }
};

}  // namespace orc
