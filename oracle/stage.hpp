// oracle/stage.hpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
// Per-knot evaluation of one StageModel (dynamics + cost stack + constraints) and its first-order /
// Gauss-Newton derivatives, written into the LQ-knot layout.
// Restates, per stage kind:
//   * aligator CentroidalFwdDynamics + IntegratorEuler and the centroidal residuals   (centroidal_talos.py:202-247)
//   * aligator MultibodyConstraintFwdDynamics (pin.constraintDynamics, KKT of SURVEY.md App. B.1)
//     + IntegratorSemiImplEuler and the whole-body residuals                          (fulldynamic_talos.py:100-232)
// The upstream C++ (Aligator >= 0.10, Pinocchio >= 2.9.1, README.md:10-16) is not vendored in the reference:
// PARITY UNPINNED — derivatives here come from forward-mode AD of the primal functions and are
// cross-checked by finite differences in tests/.
#pragma once
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include "model.hpp"

namespace orc {

struct Term {
  int type, role, dim, i0, i1, poff, woff, flags;
};

struct StageDesc {
  int dyn = MPC_DYN_NONE, ncontact = 0, cid[2] = {0, 0}, dyn_poff = 0, nc = 0;
  std::vector<Term> terms;
  std::vector<double> params;
  void parse(const int32_t* d, int n_d, const double* p, int n_p) {
    if (n_d < MPC_STAGE_HEADER_WORDS) throw std::runtime_error("stage descriptor too short");
    dyn = d[0]; ncontact = d[1]; cid[0] = d[2]; cid[1] = d[3]; dyn_poff = d[4];
    const int nt = d[5];
    nc = d[6];
    if (n_d < MPC_STAGE_HEADER_WORDS + MPC_TERM_WORDS * nt) throw std::runtime_error("stage descriptor truncated");
    terms.resize(nt);
    int ncsum = 0;
    for (int t = 0; t < nt; ++t) {
      const int32_t* w = d + MPC_STAGE_HEADER_WORDS + MPC_TERM_WORDS * t;
      terms[t] = Term{w[0], w[1], w[2], w[3], w[4], w[5], w[6], w[7]};
      if (terms[t].role != MPC_ROLE_COST) ncsum += terms[t].dim;
    }
    if (ncsum != nc) throw std::runtime_error("stage descriptor: constraint row count mismatch");
    params.assign(p, p + n_p);
  }
};

// LQ knot: z = [dx; du], row-major dense blocks
struct Knot {
  int n = 0, m = 0, c = 0;
  std::vector<double> H, grad, AB, f, cval, CD, lo, hi;
  std::vector<int> ctype;  // MPC_ROLE_* per row
  double E6[36];           // d f_base / d x'_base  (= -I for vector spaces / zero gap)
  double cost = 0.0;
  std::vector<double> xnext, xdot;
  double wrench[12];
  void resize(int n_, int m_, int c_, int nx) {
    n = n_; m = m_; c = c_;
    const int nz = n + m;
    H.assign(nz * nz, 0.0); grad.assign(nz, 0.0); AB.assign(n * nz, 0.0); f.assign(n, 0.0);
    cval.assign(c, 0.0); CD.assign(c * nz, 0.0); lo.assign(c, 0.0); hi.assign(c, 0.0); ctype.assign(c, 0);
    xnext.assign(nx, 0.0); xdot.assign(n, 0.0);
    std::memset(wrench, 0, sizeof(wrench));
    for (int i = 0; i < 36; ++i) E6[i] = (i % 7 == 0) ? -1.0 : 0.0;
    cost = 0.0;
  }
};

// ---- accumulation helpers ---------------------------------------------------------------------
// cost term: r (dim), J (dim x nz), weight -> cost, grad, GN Hessian
inline void add_cost(Knot& kn, const Term& t, const double* W, const double* r, const double* J, bool with_derivs) {
  const int nz = kn.n + kn.m, d = t.dim;
  std::vector<double> Wr(d, 0.0);
  const bool diag = t.flags & MPC_TERM_FLAG_DIAG_WEIGHT;
  for (int i = 0; i < d; ++i) {
    if (diag) Wr[i] = W[i] * r[i];
    else for (int j = 0; j < d; ++j) Wr[i] += W[i * d + j] * r[j];
  }
  double c = 0;
  for (int i = 0; i < d; ++i) c += r[i] * Wr[i];
  kn.cost += 0.5 * c;
  if (!with_derivs) return;
  std::vector<double> WJ(d * nz, 0.0);
  for (int i = 0; i < d; ++i) {
    if (diag) for (int k = 0; k < nz; ++k) WJ[i * nz + k] = W[i] * J[i * nz + k];
    else for (int j = 0; j < d; ++j) { const double w = W[i * d + j]; if (w != 0.0) for (int k = 0; k < nz; ++k) WJ[i * nz + k] += w * J[j * nz + k]; }
  }
  for (int i = 0; i < d; ++i)
    for (int a = 0; a < nz; ++a) {
      const double ja = J[i * nz + a];
      if (ja == 0.0) continue;
      kn.grad[a] += ja * Wr[i];
      for (int b = 0; b < nz; ++b) kn.H[a * nz + b] += ja * WJ[i * nz + b];
    }
}

inline void add_constraint(Knot& kn, const Term& t, const double* params, int& row, const double* r, const double* J, bool with_derivs) {
  const int nz = kn.n + kn.m;
  for (int i = 0; i < t.dim; ++i, ++row) {
    kn.cval[row] = r[i];
    kn.ctype[row] = t.role;
    if (t.role == MPC_ROLE_BOX) { kn.lo[row] = params[t.woff + i]; kn.hi[row] = params[t.woff + t.dim + i]; }
    if (with_derivs) for (int k = 0; k < nz; ++k) kn.CD[row * nz + k] = J[i * nz + k];
  }
}

// ================================================================================================
// Vector-space (centroidal) stage: x = [c; h_lin; L], u = [f0 tau0 f1 tau1]   (centroidal_talos.py:40-48)
// ================================================================================================
inline void eval_centroidal(const StageDesc& sd, int nx, int nu, const double* x, const double* u, const double* xnext,
                            Knot& kn, bool with_derivs) {
  const int n = nx, m = (sd.dyn == MPC_DYN_NONE) ? 0 : nu, nz = n + m;
  kn.resize(n, m, sd.nc, nx);
  const double* P = sd.params.data();
  if (sd.dyn == MPC_DYN_CENTROIDAL_EULER) {
    const double* dp = P + sd.dyn_poff;
    const double mass = dp[0], *g = dp + 1, dt = dp[4];
    const int nk = sd.ncontact;
    double xd[9];
    for (int i = 0; i < 3; ++i) { xd[i] = x[3 + i] / mass; xd[3 + i] = mass * g[i]; xd[6 + i] = 0.0; }
    for (int i = 0; i < n * nz; ++i) kn.AB[i] = 0.0;
    for (int i = 0; i < n; ++i) kn.AB[i * nz + i] = 1.0;
    for (int i = 0; i < 3; ++i) kn.AB[i * nz + 3 + i] += dt / mass;
    for (int k = 0; k < nk; ++k) {
      if (!sd.cid[k]) continue;
      const double* p = dp + 5 + 3 * k;
      const double* fk = u + 6 * k;
      const double r[3] = {p[0] - x[0], p[1] - x[1], p[2] - x[2]};
      for (int i = 0; i < 3; ++i) xd[3 + i] += fk[i];
      xd[6] += r[1] * fk[2] - r[2] * fk[1] + fk[3];
      xd[7] += r[2] * fk[0] - r[0] * fk[2] + fk[4];
      xd[8] += r[0] * fk[1] - r[1] * fk[0] + fk[5];
      // d(hdot)/df = I ; d(Ldot)/dc = [f]x ; d(Ldot)/df = [r]x ; d(Ldot)/dtau = I
      const double F[3][3] = {{0, -fk[2], fk[1]}, {fk[2], 0, -fk[0]}, {-fk[1], fk[0], 0}};
      const double Rx[3][3] = {{0, -r[2], r[1]}, {r[2], 0, -r[0]}, {-r[1], r[0], 0}};
      for (int i = 0; i < 3; ++i) {
        kn.AB[(3 + i) * nz + n + 6 * k + i] += dt;
        kn.AB[(6 + i) * nz + n + 6 * k + 3 + i] += dt;
        for (int j = 0; j < 3; ++j) { kn.AB[(6 + i) * nz + j] += dt * F[i][j]; kn.AB[(6 + i) * nz + n + 6 * k + j] += dt * Rx[i][j]; }
      }
    }
    for (int i = 0; i < n; ++i) { kn.xdot[i] = xd[i]; kn.xnext[i] = x[i] + dt * xd[i]; kn.f[i] = kn.xnext[i] - xnext[i]; }
  }
  // terms
  int row = 0;
  std::vector<double> r(64), J;
  for (const Term& t : sd.terms) {
    const int d = t.dim;
    r.assign(d, 0.0); J.assign(d * nz, 0.0);
    const double* tp = P + t.poff;
    switch (t.type) {
      case MPC_TERM_STATE_ERROR:
        for (int i = 0; i < d; ++i) { r[i] = tp[t.i0 + i] - x[t.i0 + i]; J[i * nz + t.i0 + i] = -1.0; }  // x_ref (-) x, see DESIGN.md
        break;
      case MPC_TERM_CONTROL_ERROR:
        for (int i = 0; i < d; ++i) { r[i] = u[t.i0 + i] - tp[t.i0 + i]; J[i * nz + n + t.i0 + i] = 1.0; }
        break;
      case MPC_TERM_CENTROIDAL_WRENCH_CONE:
        for (int i = 0; i < d; ++i) for (int j = 0; j < 6; ++j) { r[i] += tp[i * 6 + j] * u[6 * t.i0 + j]; J[i * nz + n + 6 * t.i0 + j] = tp[i * 6 + j]; }
        break;
      case MPC_TERM_CENTROIDAL_LIN_ACC: {
        const double mass = tp[0], *g = tp + 1;
        const int nk = t.i0;
        for (int i = 0; i < 3; ++i) r[i] = g[i];
        for (int k = 0; k < nk; ++k) {
          if (tp[4 + 4 * k] == 0.0) continue;
          for (int i = 0; i < 3; ++i) { r[i] += u[6 * k + i] / mass; J[i * nz + n + 6 * k + i] = 1.0 / mass; }
        }
      } break;
      case MPC_TERM_CENTROIDAL_ANG_ACC: {
        const int nk = t.i0;
        for (int k = 0; k < nk; ++k) {
          if (tp[4 + 4 * k] == 0.0) continue;
          const double* p = tp + 4 + 4 * k + 1;
          const double* fk = u + 6 * k;
          const double rr[3] = {p[0] - x[0], p[1] - x[1], p[2] - x[2]};
          r[0] += rr[1] * fk[2] - rr[2] * fk[1] + fk[3];
          r[1] += rr[2] * fk[0] - rr[0] * fk[2] + fk[4];
          r[2] += rr[0] * fk[1] - rr[1] * fk[0] + fk[5];
          const double F[3][3] = {{0, -fk[2], fk[1]}, {fk[2], 0, -fk[0]}, {-fk[1], fk[0], 0}};
          const double Rx[3][3] = {{0, -rr[2], rr[1]}, {rr[2], 0, -rr[0]}, {-rr[1], rr[0], 0}};
          for (int i = 0; i < 3; ++i) {
            J[i * nz + n + 6 * k + 3 + i] += 1.0;
            for (int j = 0; j < 3; ++j) { J[i * nz + j] += F[i][j]; J[i * nz + n + 6 * k + j] += Rx[i][j]; }
          }
        }
      } break;
      default:
        throw std::runtime_error("term type " + std::to_string(t.type) + " not valid on a vector-space stage");
    }
    if (t.role == MPC_ROLE_COST) add_cost(kn, t, P + t.woff, r.data(), J.data(), with_derivs);
    else add_constraint(kn, t, P, row, r.data(), J.data(), with_derivs);
  }
}

// ================================================================================================
// Multibody stages
// ================================================================================================
inline void solve_dense(std::vector<double>& A, int n, std::vector<double>& B, int nrhs) {
  // Gaussian elimination with partial pivoting, A (n x n) and B (n x nrhs) row-major, in place
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double best = std::fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) if (std::fabs(A[i * n + k]) > best) { best = std::fabs(A[i * n + k]); piv = i; }
    if (best == 0.0) throw std::runtime_error("singular matrix in solve_dense");
    if (piv != k) {
      for (int j = 0; j < n; ++j) std::swap(A[k * n + j], A[piv * n + j]);
      for (int j = 0; j < nrhs; ++j) std::swap(B[k * nrhs + j], B[piv * nrhs + j]);
    }
    const double inv = 1.0 / A[k * n + k];
    for (int i = k + 1; i < n; ++i) {
      const double l = A[i * n + k] * inv;
      if (l == 0.0) continue;
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= l * A[k * n + j];
      for (int j = 0; j < nrhs; ++j) B[i * nrhs + j] -= l * B[k * nrhs + j];
    }
  }
  for (int k = n - 1; k >= 0; --k) {
    const double inv = 1.0 / A[k * n + k];
    for (int j = 0; j < nrhs; ++j) {
      double s = B[k * nrhs + j];
      for (int i = k + 1; i < n; ++i) s -= A[k * n + i] * B[i * nrhs + j];
      B[k * nrhs + j] = s * inv;
    }
  }
}

template <class T> Mot<T> elemwise(const double* k, const Mot<T>& m) {
  return Mot<T>{V3<T>(T(k[0]) * m.lin[0], T(k[1]) * m.lin[1], T(k[2]) * m.lin[2]), V3<T>(T(k[3]) * m.ang[0], T(k[4]) * m.ang[1], T(k[5]) * m.ang[2])};
}

// ---- kinodynamics (kinodynamic_talos.py:107-112): centroidal momentum balance closes the base acceleration ----
// rate of the centroidal momentum d/dt hg (about the CoM, world axes) for joint accelerations `a`, no gravity
template <class T>
Frc<T> momentum_rate(const Model& m, const State<T>& s, const double* a, V3<T>& com) {
  std::vector<T> aT(m.nv);
  for (int i = 0; i < m.nv; ++i) aT[i] = T(a ? a[i] : 0.0);
  Kin<T> k;
  forward_pass(m, s, aT.data(), k, false);
  Frc<T> f0;
  V3<T> mc;
  for (int i = 0; i < m.nj; ++i) {
    const Inertia<T> Y = convert<T>(m.inertia[i]);
    f0 = f0 + act(k.oMi[i], Y * k.a[i] + fcross(k.v[i], Y * k.v[i]));
    mc = mc + Y.mass * (k.oMi[i].R * Y.c + k.oMi[i].p);
  }
  com = T(1.0 / m.total_mass) * mc;
  return Frc<T>{f0.lin, f0.ang - cross(com, f0.lin)};
}
// momentum rate produced by the contact wrenches u = [f0 tau0 f1 tau1 ...] (world axes, applied at the sole
// frame origins) and gravity:  [sum f + m g ; sum (p_i - c) x f_i + tau_i]
template <class T>
Frc<T> wrench_rate(const Model& m, const State<T>& s, const double* u, const double* grav, const int* states, const int* frames, int nk) {
  Kin<T> k;
  forward_pass<T>(m, s, nullptr, k, false);
  V3<T> mc;
  for (int i = 0; i < m.nj; ++i) { const Inertia<T> Y = convert<T>(m.inertia[i]); mc = mc + Y.mass * (k.oMi[i].R * Y.c + k.oMi[i].p); }
  const V3<T> com = T(1.0 / m.total_mass) * mc;
  Frc<T> h;
  h.lin = V3<T>(T(m.total_mass * grav[0]), T(m.total_mass * grav[1]), T(m.total_mass * grav[2]));
  for (int c = 0; c < nk; ++c) {
    if (!states[c]) continue;
    const SE3<T> oMf = k.oMi[m.frame_joint[frames[c]]] * convert<T>(m.frame_pl[frames[c]]);
    const V3<T> f(T(u[6 * c]), T(u[6 * c + 1]), T(u[6 * c + 2])), tau(T(u[6 * c + 3]), T(u[6 * c + 4]), T(u[6 * c + 5]));
    h.lin = h.lin + f;
    h.ang = h.ang + cross(oMf.p - com, f) + tau;
  }
  return h;
}

// Everything that is a function of (q, v) only, given the solved (a, lambda) as constants:
//   r1 = RNEA(q, v, a; fext = contact forces fixed in the contact frames)              (nv)
//   r2 = contact-frame spatial acceleration + Kd*vel_err - Kp*log6(c1Mc2) per contact  (6 each)
//   rt = stacked residuals of the (q,v)-only terms of the stage
template <class T>
void mb_functions(const Model& m, const StageDesc& sd, const State<T>& s, const double* a, const double* lam,
                  std::vector<T>& r1, std::vector<T>& r2, std::vector<T>& rt, const double* u = nullptr) {
  const bool dyn = sd.dyn == MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER;
  const int nk = dyn ? sd.ncontact : 0;
  std::vector<T> aT(m.nv);
  for (int i = 0; i < m.nv; ++i) aT[i] = T(a ? a[i] : 0.0);
  Kin<T> k;
  forward_pass(m, s, aT.data(), k, true);
  r1.assign(m.nv, T(0.0));
  r2.assign(6 * nk, T(0.0));
  if (dyn) {
    std::vector<Frc<T>> fext(m.nj);
    for (int c = 0; c < nk; ++c) {
      const ContactModel& cm = m.contacts[sd.cid[c]];
      Frc<T> l;
      for (int i = 0; i < 3; ++i) { l.lin[i] = T(lam[6 * c + i]); l.ang[i] = T(lam[6 * c + 3 + i]); }
      fext[cm.joint] = fext[cm.joint] + act(convert<T>(cm.pl1), l);
    }
    rnea_backward(m, k, &fext, r1.data());
    for (int c = 0; c < nk; ++c) {
      const ContactModel& cm = m.contacts[sd.cid[c]];
      const SE3<T> iMc = convert<T>(cm.pl1);
      const int j = cm.joint;
      // true spatial acceleration = gravity-field acceleration minus the field itself
      Mot<T> a0;
      a0.lin = V3<T>(T(-m.gravity[0]), T(-m.gravity[1]), T(-m.gravity[2]));
      const Mot<T> atrue = k.a[j] - actInv(k.oMi[j], a0);
      const Mot<T> ac = actInv(iMc, atrue);
      const Mot<T> vc = actInv(iMc, k.v[j]);
      const SE3<T> c1Mc2 = inverse(k.oMi[j] * iMc) * convert<T>(cm.pl2);
      const Mot<T> e = log6(c1Mc2);
      const Mot<T> res = ac + elemwise(cm.Kd, vc) - elemwise(cm.Kp, e);
      for (int i = 0; i < 3; ++i) { r2[6 * c + i] = res.lin[i]; r2[6 * c + 3 + i] = res.ang[i]; }
    }
  }
  // (q,v)-only terms
  rt.clear();
  V3<T> com;
  Frc<T> hg;
  bool have_cent = false;
  for (const Term& t : sd.terms) {
    const double* tp = sd.params.data() + t.poff;
    switch (t.type) {
      case MPC_TERM_STATE_ERROR: {
        // r = x_ref (-) x = difference(x, x_ref) = [log6(M^-1 Mref); qa_ref - qa; v_ref - v], sliced
        // (sign convention inferred from the flipped joint-limit bounds at fulldynamic_talos.py:209)
        std::vector<T> full(2 * m.nv);
        const State<double> ref = state_from_x(m, tp);
        for (int i = 0; i < m.nv; ++i) { full[i] = T(ref.qa[i]) - s.qa[i]; full[m.nv + i] = T(ref.v[i]) - s.v[i]; }
        if (m.has_freeflyer()) {
          const Mot<T> e = log6(inverse(s.base) * convert<T>(ref.base));
          for (int i = 0; i < 3; ++i) { full[i] = e.lin[i]; full[3 + i] = e.ang[i]; }
        }
        for (int i = 0; i < t.dim; ++i) rt.push_back(full[t.i0 + i]);
      } break;
      case MPC_TERM_FRAME_PLACEMENT: {
        const SE3<T> oMf = k.oMi[m.frame_joint[t.i0]] * convert<T>(m.frame_pl[t.i0]);
        const Mot<T> e = log6(inverse(convert<T>(Model::read_se3(tp))) * oMf);
        for (int i = 0; i < 3; ++i) rt.push_back(e.lin[i]);
        for (int i = 0; i < 3; ++i) rt.push_back(e.ang[i]);
      } break;
      case MPC_TERM_FRAME_TRANSLATION: {
        const SE3<T> oMf = k.oMi[m.frame_joint[t.i0]] * convert<T>(m.frame_pl[t.i0]);
        for (int i = 0; i < t.dim; ++i) rt.push_back(oMf.p[t.i1 + i] - T(tp[t.i1 + i]));
      } break;
      case MPC_TERM_FRAME_VELOCITY: {
        const Mot<T> vf = actInv(convert<T>(m.frame_pl[t.i0]), k.v[m.frame_joint[t.i0]]);
        for (int i = 0; i < 3; ++i) rt.push_back(vf.lin[i] - T(tp[i]));
        for (int i = 0; i < 3; ++i) rt.push_back(vf.ang[i] - T(tp[3 + i]));
      } break;
      case MPC_TERM_COM_TRANSLATION:
      case MPC_TERM_CENTROIDAL_MOMENTUM: {
        if (!have_cent) { centroidal(m, k, com, hg); have_cent = true; }
        if (t.type == MPC_TERM_COM_TRANSLATION) {
          for (int i = 0; i < t.dim; ++i) rt.push_back(com[t.i1 + i] - T(tp[t.i1 + i]));
        } else {
          for (int i = 0; i < 3; ++i) rt.push_back(hg.lin[i] - T(tp[i]));
          for (int i = 0; i < 3; ++i) rt.push_back(hg.ang[i] - T(tp[3 + i]));
        }
      } break;
      case MPC_TERM_CENTROIDAL_MOMENTUM_DER: {
        // params: g[3], states[nk], frames[nk]
        const int nkk = t.i0;
        int states[4], frames[4];
        for (int c = 0; c < nkk; ++c) { states[c] = tp[3 + c] != 0.0; frames[c] = (int)tp[3 + nkk + c]; }
        const Frc<T> h = wrench_rate<T>(m, s, u, tp, states, frames, nkk);
        for (int i = 0; i < 3; ++i) rt.push_back(h.lin[i]);
        for (int i = 0; i < 3; ++i) rt.push_back(h.ang[i]);
      } break;
      default: break;  // control / contact-force terms handled by the caller
    }
  }
}

inline bool is_qv_term(int type) {
  return type == MPC_TERM_STATE_ERROR || type == MPC_TERM_FRAME_PLACEMENT || type == MPC_TERM_FRAME_TRANSLATION ||
         type == MPC_TERM_FRAME_VELOCITY || type == MPC_TERM_COM_TRANSLATION || type == MPC_TERM_CENTROIDAL_MOMENTUM ||
         type == MPC_TERM_CENTROIDAL_MOMENTUM_DER;
}

// x' = x (+) dx on the multibody phase space (Pinocchio integrate on q, plain sum on v)
inline void mb_integrate(const Model& m, const double* x, const double* dx, double* out) {
  State<double> s = state_from_x(m, x);
  if (m.has_freeflyer()) {
    Mot<double> nu;
    for (int i = 0; i < 3; ++i) { nu.lin[i] = dx[i]; nu.ang[i] = dx[3 + i]; }
    s.base = s.base * exp6(nu);
  }
  for (int i = 0; i < m.nj; ++i) if (m.kind[i] != MPC_JOINT_FREEFLYER) s.qa[m.idx_v[i]] += dx[m.idx_v[i]];
  for (int i = 0; i < m.nv; ++i) s.v[i] += dx[m.nv + i];
  x_from_state(m, s, out);
}
// d = x1 (-) x0
inline void mb_difference(const Model& m, const double* x0, const double* x1, double* d) {
  const State<double> a = state_from_x(m, x0), b = state_from_x(m, x1);
  for (int i = 0; i < m.nv; ++i) { d[i] = b.qa[i] - a.qa[i]; d[m.nv + i] = b.v[i] - a.v[i]; }
  if (m.has_freeflyer()) {
    const Mot<double> e = log6(inverse(a.base) * b.base);
    for (int i = 0; i < 3; ++i) { d[i] = e.lin[i]; d[3 + i] = e.ang[i]; }
  }
}

// external generalized force (size nv, Pinocchio's joint-frame convention) added to the right-hand side of the forward dynamics: set
// only by Solver::simulate around its value-only evaluations (the disturbance of fulldynamic_talos.py:433-435, 524-526)
inline const double*& ext_tau() { static thread_local const double* p = nullptr; return p; }

inline void eval_multibody(const Model& m, const StageDesc& sd, int nu, const double* x, const double* u, const double* xnext,
                           Knot& kn, bool with_derivs) {
  const int nv = m.nv, n = 2 * nv, nx = m.nq + nv;
  const bool dyn = sd.dyn == MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER;
  const bool kino = sd.dyn == MPC_DYN_KINODYNAMICS_SEMIEULER;
  if (sd.dyn != MPC_DYN_NONE && !dyn && !kino) throw std::runtime_error("oracle: dynamics kind not implemented for multibody stages");
  const int mm = (dyn || kino) ? nu : 0, nz = n + mm;
  const int nk = dyn ? sd.ncontact : 0, nl = 6 * nk, nK = nv + nl;
  kn.resize(n, mm, sd.nc, nx);
  const double* P = sd.params.data();
  const State<double> s0 = state_from_x(m, x);

  // ---------------- K1: constrained forward dynamics (primal) ----------------
  std::vector<double> a(nv, 0.0), lam(nl > 0 ? nl : 1, 0.0), Kinv;
  double dt = 0.0;
  if (dyn) {
    dt = P[sd.dyn_poff];
    dual_nd() = 0;
    // bias terms: b = RNEA(q, v, 0) and contact drift at a = 0, lambda = 0
    std::vector<double> b, gam, rt;
    mb_functions<double>(m, sd, s0, nullptr, lam.data(), b, gam, rt);
    // mass matrix and contact Jacobian column by column (unit accelerations, no velocity, no gravity)
    std::vector<double> Kmat(nK * nK, 0.0);
    {
      State<double> sz = s0;
      std::fill(sz.v.begin(), sz.v.end(), 0.0);
      Kin<double> k;
      std::vector<double> e(nv, 0.0), col(nv);
      for (int j = 0; j < nv; ++j) {
        e[j] = 1.0;
        forward_pass(m, sz, e.data(), k, false);
        rnea_backward<double>(m, k, nullptr, col.data());
        for (int i = 0; i < nv; ++i) Kmat[i * nK + j] = col[i];
        for (int c = 0; c < nk; ++c) {
          const ContactModel& cm = m.contacts[sd.cid[c]];
          const Mot<double> ac = actInv(cm.pl1, k.a[cm.joint]);
          for (int i = 0; i < 3; ++i) {
            Kmat[(nv + 6 * c + i) * nK + j] = ac.lin[i]; Kmat[(nv + 6 * c + 3 + i) * nK + j] = ac.ang[i];
            Kmat[j * nK + nv + 6 * c + i] = ac.lin[i]; Kmat[j * nK + nv + 6 * c + 3 + i] = ac.ang[i];
          }
        }
        e[j] = 0.0;
      }
      for (int i = 0; i < nl; ++i) Kmat[(nv + i) * nK + nv + i] = -m.prox_mu;
    }
    // K [a; -lambda] = [S u - b ; -gamma]   (one proximal step from lambda = 0: pin.ProximalSettings(1e-9, 1e-10, 1))
    Kinv.assign(nK * nK, 0.0);
    for (int i = 0; i < nK; ++i) Kinv[i * nK + i] = 1.0;
    solve_dense(Kmat, nK, Kinv, nK);
    std::vector<double> rhs(nK, 0.0);
    for (int i = 0; i < nv; ++i) rhs[i] = -b[i] + (i >= nv - nu ? u[i - (nv - nu)] : 0.0) + (ext_tau() ? ext_tau()[i] : 0.0);
    for (int i = 0; i < nl; ++i) rhs[nv + i] = -gam[i];
    for (int i = 0; i < nv; ++i) { double sacc = 0; for (int j = 0; j < nK; ++j) sacc += Kinv[i * nK + j] * rhs[j]; a[i] = sacc; }
    for (int i = 0; i < nl; ++i) { double sacc = 0; for (int j = 0; j < nK; ++j) sacc += Kinv[(nv + i) * nK + j] * rhs[j]; lam[i] = -sacc; }
    for (int c = 0; c < nk; ++c) {
      // wrench slots are indexed by the model's contact id so that callers see [left; right]
      for (int i = 0; i < 6; ++i) kn.wrench[6 * sd.cid[c] + i] = lam[6 * c + i];
    }
    for (int i = 0; i < nv; ++i) { kn.xdot[i] = s0.v[i]; kn.xdot[nv + i] = a[i]; }
  }

  // ---------------- kinodynamics: a_joint = u[12:], base acceleration from the centroidal momentum balance ----------------
  std::vector<double> da_kino;  // nv x nz
  if (kino) {
    const double* dp = P + sd.dyn_poff;
    dt = dp[0];
    const int nkk = sd.ncontact, nf = 6 * nkk;
    int states[4], frames[4];
    for (int c = 0; c < nkk; ++c) { states[c] = sd.cid[c]; frames[c] = (int)dp[4 + c]; }
    dual_nd() = 0;
    for (int i = 6; i < nv; ++i) a[i] = u[nf + i - 6];
    // Ag columns: hg(q, e_k)
    std::vector<double> Ag(6 * nv);
    {
      State<double> se = s0;
      Kin<double> k;
      for (int j = 0; j < nv; ++j) {
        std::fill(se.v.begin(), se.v.end(), 0.0);
        se.v[j] = 1.0;
        forward_pass<double>(m, se, nullptr, k, false);
        V3<double> c; Frc<double> hg;
        centroidal(m, k, c, hg);
        for (int i = 0; i < 3; ++i) { Ag[i * nv + j] = hg.lin[i]; Ag[(3 + i) * nv + j] = hg.ang[i]; }
      }
    }
    V3<double> com;
    std::vector<double> aj(a);
    for (int i = 0; i < 6; ++i) aj[i] = 0.0;
    const Frc<double> hd0 = momentum_rate<double>(m, s0, aj.data(), com);  // dAg v + Ag[:,6:] a_j
    const Frc<double> hdes = wrench_rate<double>(m, s0, u, dp + 1, states, frames, nkk);
    std::vector<double> Agb(36), rhs(6);
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Agb[i * 6 + j] = Ag[i * nv + j];
    for (int i = 0; i < 3; ++i) { rhs[i] = hdes.lin[i] - hd0.lin[i]; rhs[3 + i] = hdes.ang[i] - hd0.ang[i]; }
    std::vector<double> Agb_inv(36, 0.0);
    for (int i = 0; i < 6; ++i) Agb_inv[i * 6 + i] = 1.0;
    { std::vector<double> tmp(Agb); solve_dense(tmp, 6, Agb_inv, 6); }
    for (int i = 0; i < 6; ++i) { double sacc = 0; for (int j = 0; j < 6; ++j) sacc += Agb_inv[i * 6 + j] * rhs[j]; a[i] = sacc; }
    for (int i = 0; i < nv; ++i) { kn.xdot[i] = s0.v[i]; kn.xdot[nv + i] = a[i]; }
    if (with_derivs) {
      // r(q, v) = momentum_rate(q, v, a) - wrench_rate(q, u)   (6 rows), a and u held fixed
      dual_nd() = n;
      const State<Dual> sT = lift<Dual>(m, s0, 0);
      V3<Dual> cT;
      const Frc<Dual> hrT = momentum_rate<Dual>(m, sT, a.data(), cT);
      const Frc<Dual> hwT = wrench_rate<Dual>(m, sT, u, dp + 1, states, frames, nkk);
      std::vector<double> dr(6 * nz, 0.0);
      for (int i = 0; i < 3; ++i) for (int k = 0; k < n; ++k) { dr[i * nz + k] = hrT.lin[i].d[k] - hwT.lin[i].d[k]; dr[(3 + i) * nz + k] = hrT.ang[i].d[k] - hwT.ang[i].d[k]; }
      dual_nd() = 0;
      // d r / d u: wrenches (- [I ; (p - c) x], - [0 ; I]) and joint accelerations (Ag columns)
      Kin<double> k0;
      forward_pass<double>(m, s0, nullptr, k0, false);
      for (int c = 0; c < nkk; ++c) {
        if (!states[c]) continue;
        const SE3<double> oMf = k0.oMi[m.frame_joint[frames[c]]] * m.frame_pl[frames[c]];
        const V3<double> r = oMf.p - com;
        const M3<double> Rx = skew(r);
        for (int i = 0; i < 3; ++i) {
          dr[i * nz + n + 6 * c + i] -= 1.0;
          dr[(3 + i) * nz + n + 6 * c + 3 + i] -= 1.0;
          for (int j = 0; j < 3; ++j) dr[(3 + i) * nz + n + 6 * c + j] -= Rx(i, j);
        }
      }
      for (int j = 6; j < nv; ++j) for (int i = 0; i < 6; ++i) dr[i * nz + n + nf + j - 6] += Ag[i * nv + j];
      da_kino.assign(nv * nz, 0.0);
      for (int i = 0; i < 6; ++i) for (int k = 0; k < nz; ++k) { double sacc = 0; for (int j = 0; j < 6; ++j) sacc += Agb_inv[i * 6 + j] * dr[j * nz + k]; da_kino[i * nz + k] = -sacc; }
      for (int j = 6; j < nv; ++j) da_kino[j * nz + n + nf + j - 6] = 1.0;
    }
  }

  // ---------------- K2/K4: residual values and (q,v)-Jacobians by forward AD ----------------
  std::vector<double> r1v, r2v, rtv;
  std::vector<double> Jr1, Jr2, Jrt;  // (rows x 2nv)
  {
    dual_nd() = 0;
    mb_functions<double>(m, sd, s0, a.data(), lam.data(), r1v, r2v, rtv, u);
  }
  const int nrt = (int)rtv.size();
  if (with_derivs) {
    dual_nd() = n;
    std::vector<Dual> r1, r2, rt;
    const State<Dual> sT = lift<Dual>(m, s0, 0);
    mb_functions<Dual>(m, sd, sT, a.data(), lam.data(), r1, r2, rt, u);
    Jr1.assign(nv * n, 0.0); Jr2.assign(nl * n, 0.0); Jrt.assign(nrt * n, 0.0);
    for (int i = 0; i < nv && dyn; ++i) for (int k = 0; k < n; ++k) Jr1[i * n + k] = r1[i].d[k];
    for (int i = 0; i < nl; ++i) for (int k = 0; k < n; ++k) Jr2[i * n + k] = r2[i].d[k];
    for (int i = 0; i < nrt; ++i) for (int k = 0; k < n; ++k) Jrt[i * n + k] = rt[i].d[k];
    dual_nd() = 0;
  }

  // ---------------- implicit differentiation of the KKT system ----------------
  // d[a; -lam]/d(q,v) = -Kinv [dr1; dr2] ;  d[a; -lam]/du = Kinv[:, actuated columns]
  std::vector<double> da(nv * nz, 0.0), dlam(nl * nz > 0 ? nl * nz : 1, 0.0);
  if (dyn && with_derivs) {
    for (int i = 0; i < nK; ++i) {
      for (int k = 0; k < n; ++k) {
        double sacc = 0;
        for (int j = 0; j < nv; ++j) sacc += Kinv[i * nK + j] * Jr1[j * n + k];
        for (int j = 0; j < nl; ++j) sacc += Kinv[i * nK + nv + j] * Jr2[j * n + k];
        if (i < nv) da[i * nz + k] = -sacc; else dlam[(i - nv) * nz + k] = sacc;
      }
      for (int k = 0; k < nu; ++k) {
        const double val = Kinv[i * nK + (nv - nu) + k];
        if (i < nv) da[i * nz + n + k] = val; else dlam[(i - nv) * nz + n + k] = -val;
      }
    }
  }

  if (kino && with_derivs) da = da_kino;
  // ---------------- K3: semi-implicit Euler, dynamics gap and its Jacobians ----------------
  if (dyn || kino) {
    // primal: v+ = v + dt a ; q+ = q (+) dt v+
    std::vector<double> dx(n);
    for (int i = 0; i < nv; ++i) { const double vp = s0.v[i] + dt * a[i]; dx[i] = dt * vp; dx[nv + i] = dt * a[i]; }
    mb_integrate(m, x, dx.data(), kn.xnext.data());
    const State<double> sn = state_from_x(m, xnext);
    if (with_derivs) {
      if (nz > MAXD) throw std::runtime_error("oracle: n + m exceeds the AD tangent capacity");
      dual_nd() = nz;
      State<Dual> sT = lift<Dual>(m, s0, 0);
      std::vector<Dual> vp(nv);
      for (int i = 0; i < nv; ++i) {
        Dual aT(a[i]);
        for (int k = 0; k < nz; ++k) aT.d[k] = da[i * nz + k];
        vp[i] = sT.v[i] + Dual(dt) * aT;
      }
      std::vector<Dual> fT(n);
      for (int i = 0; i < nv; ++i) { fT[i] = sT.qa[i] + Dual(dt) * vp[i] - Dual(sn.qa[i]); fT[nv + i] = vp[i] - Dual(sn.v[i]); }
      if (m.has_freeflyer()) {
        Mot<Dual> nuT;
        for (int i = 0; i < 3; ++i) { nuT.lin[i] = Dual(dt) * vp[i]; nuT.ang[i] = Dual(dt) * vp[3 + i]; }
        const SE3<Dual> Mp = sT.base * exp6(nuT);
        const Mot<Dual> e = log6(inverse(convert<Dual>(sn.base)) * Mp);
        for (int i = 0; i < 3; ++i) { fT[i] = e.lin[i]; fT[3 + i] = e.ang[i]; }
      }
      for (int i = 0; i < n; ++i) { kn.f[i] = fT[i].v; for (int k = 0; k < nz; ++k) kn.AB[i * nz + k] = fT[i].d[k]; }
      // E6 = d f_base / d x'_base
      if (m.has_freeflyer()) {
        dual_nd() = 6;
        State<double> sp = state_from_x(m, kn.xnext.data());
        // seed the *next* state's base
        State<Dual> snT;
        {
          const State<double>& tmp = sn;  // seed only the base of the *next* state
          snT.base = convert<Dual>(tmp.base);
          for (int k = 0; k < 3; ++k) {
            for (int i = 0; i < 3; ++i) snT.base.p[i].d[k] = tmp.base.R(i, k);
            V3<double> ek; ek[k] = 1.0;
            const M3<double> dR = tmp.base.R * skew(ek);
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) snT.base.R(i, j).d[3 + k] = dR(i, j);
          }
        }
        const Mot<Dual> e = log6(inverse(snT.base) * convert<Dual>(sp.base));
        for (int i = 0; i < 3; ++i) for (int k = 0; k < 6; ++k) { kn.E6[i * 6 + k] = e.lin[i].d[k]; kn.E6[(3 + i) * 6 + k] = e.ang[i].d[k]; }
      }
      dual_nd() = 0;
    } else {
      mb_difference(m, xnext, kn.xnext.data(), kn.f.data());
    }
  }

  // ---------------- K4/K5: cost stack and constraints ----------------
  int row = 0, rtoff = 0;
  std::vector<double> r, J;
  for (const Term& t : sd.terms) {
    const int d = t.dim;
    r.assign(d, 0.0); J.assign(d * nz, 0.0);
    const double* tp = P + t.poff;
    if (is_qv_term(t.type)) {
      for (int i = 0; i < d; ++i) {
        r[i] = rtv[rtoff + i];
        if (with_derivs) for (int k = 0; k < n; ++k) J[i * nz + k] = Jrt[(rtoff + i) * n + k];
      }
      rtoff += d;
      if (t.type == MPC_TERM_CENTROIDAL_MOMENTUM_DER && with_derivs && mm > 0) {
        // u-part: d/df = [I ; (p - c) x], d/dtau = [0 ; I] per active contact
        const int nkk = t.i0;
        Kin<double> k0;
        dual_nd() = 0;
        forward_pass<double>(m, s0, nullptr, k0, false);
        V3<double> c0; Frc<double> hg0;
        centroidal(m, k0, c0, hg0);
        for (int c = 0; c < nkk; ++c) {
          if (tp[3 + c] == 0.0) continue;
          const int fr = (int)tp[3 + nkk + c];
          const SE3<double> oMf = k0.oMi[m.frame_joint[fr]] * m.frame_pl[fr];
          const M3<double> Rx = skew(oMf.p - c0);
          for (int i = 0; i < 3; ++i) {
            J[i * nz + n + 6 * c + i] = 1.0;
            J[(3 + i) * nz + n + 6 * c + 3 + i] = 1.0;
            for (int j = 0; j < 3; ++j) J[(3 + i) * nz + n + 6 * c + j] = Rx(i, j);
          }
        }
      }
    } else if (t.type == MPC_TERM_CENTROIDAL_WRENCH_CONE) {
      for (int i = 0; i < d; ++i) for (int j = 0; j < 6; ++j) { r[i] += tp[i * 6 + j] * u[6 * t.i0 + j]; J[i * nz + n + 6 * t.i0 + j] = tp[i * 6 + j]; }
    } else if (t.type == MPC_TERM_CONTROL_ERROR) {
      for (int i = 0; i < d; ++i) { r[i] = u[t.i0 + i] - tp[t.i0 + i]; J[i * nz + n + t.i0 + i] = 1.0; }
    } else if (t.type == MPC_TERM_CONTACT_FORCE) {
      for (int i = 0; i < 6; ++i) { r[i] = lam[6 * t.i0 + i] - tp[i]; for (int k = 0; k < nz; ++k) J[i * nz + k] = dlam[(6 * t.i0 + i) * nz + k]; }
    } else if (t.type == MPC_TERM_MB_WRENCH_CONE) {
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < 6; ++j) {
          const double aij = tp[i * 6 + j];
          r[i] += aij * lam[6 * t.i0 + j];
          if (aij != 0.0) for (int k = 0; k < nz; ++k) J[i * nz + k] += aij * dlam[(6 * t.i0 + j) * nz + k];
        }
    } else {
      throw std::runtime_error("term type " + std::to_string(t.type) + " not implemented on multibody stages");
    }
    if (t.role == MPC_ROLE_COST) add_cost(kn, t, P + t.woff, r.data(), J.data(), with_derivs);
    else add_constraint(kn, t, P, row, r.data(), J.data(), with_derivs);
  }
}

}  // namespace orc
