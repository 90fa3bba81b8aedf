// oracle/qp_capi.cpp — TEST INFRASTRUCTURE: the mpc_qp_* entry points of include/mpc_qp_abi.h on the CPU (oracle/qp.hpp).
#include <cstring>
#include <stdexcept>
#include <string>
#include "qp.hpp"
#include "model.hpp"

struct mpc_qp_solver {
  mpc_qp_dims d;
  std::vector<double> x, y, z;  // previous solution (warm start)
  orc::Model model;             // mpc_qp_set_model
  bool has_model = false;
  std::string err;
};

extern "C" {

int mpc_qp_create(const mpc_qp_dims* dims, mpc_qp_solver** out) {
  if (!dims || !out) return -2;
  if (dims->batch <= 0 || dims->n <= 0 || dims->neq < 0 || dims->nin < 0) return -2;
  mpc_qp_solver* s = new mpc_qp_solver();
  s->d = *dims;
  const size_t B = dims->batch, m = dims->nin + (dims->box ? dims->n : 0);
  s->x.assign(B * dims->n, 0.0); s->y.assign(B * dims->neq, 0.0); s->z.assign(B * m, 0.0);
  *out = s;
  return 0;
}
void mpc_qp_destroy(mpc_qp_solver* s) { delete s; }
const char* mpc_qp_last_error(mpc_qp_solver* s) { return s ? s->err.c_str() : "null handle"; }
void orc_qp_set_error(mpc_qp_solver* s, const char* what) { if (s) s->err = what; }  // (for mpc_qp_low_level_steps in capi.cpp)
void mpc_qp_default_settings(mpc_qp_settings* o) {
  o->eps_abs = 1e-5; o->rho = 1e-6; o->mu_eq = 1e-3; o->mu_in = 1e-1; o->mu_min_eq = 1e-9; o->mu_min_in = 1e-8;
  o->mu_update_factor = 0.1; o->alpha_bcl = 0.1; o->beta_bcl = 0.9; o->max_iter = 10000; o->max_iter_in = 1500; o->warm_start = 0; o->reserved = 0;
}

int mpc_qp_solve(mpc_qp_solver* s, const mpc_qp_settings* S, const double* H, const double* g, const double* A, const double* b,
                 const double* C, const double* l, const double* u, const double* l_box, const double* u_box,
                 double* x, double* y, double* z, double* z_box, mpc_qp_info* info) {
  if (!s) return -2;
  try {
    if (!S || !H || !g || !x || !info) throw std::runtime_error("qp_solve: null argument");
    const mpc_qp_dims& d = s->d;
    if (d.box && (!l_box || !u_box)) throw std::runtime_error("qp_solve: box bounds missing");
    const size_t n = d.n, neq = d.neq, nin = d.nin, m = nin + (d.box ? n : 0);
    for (int bi = 0; bi < d.batch; ++bi) {
      qp::Problem P;
      P.n = d.n; P.neq = d.neq; P.nin = d.nin; P.box = d.box;
      P.H = H + bi * n * n; P.g = g + bi * n; P.A = A ? A + bi * neq * n : nullptr; P.b = b ? b + bi * neq : nullptr;
      P.C = C ? C + bi * nin * n : nullptr; P.l = l ? l + bi * nin : nullptr; P.u = u ? u + bi * nin : nullptr;
      P.lb = d.box ? l_box + bi * n : nullptr; P.ub = d.box ? u_box + bi * n : nullptr;
      double* xs = s->x.data() + bi * n; double* ys = s->y.data() + bi * neq; double* zs = s->z.data() + bi * m;
      if (!S->warm_start) { std::fill(xs, xs + n, 0.0); std::fill(ys, ys + neq, 0.0); std::fill(zs, zs + m, 0.0); }
      qp::solve_one(P, *S, xs, ys, zs, info[bi]);
      std::memcpy(x + bi * n, xs, n * sizeof(double));
      if (y) std::memcpy(y + bi * neq, ys, neq * sizeof(double));
      if (z) std::memcpy(z + bi * nin, zs, nin * sizeof(double));
      if (z_box && d.box) std::memcpy(z_box + bi * n, zs + nin, n * sizeof(double));
    }
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

int mpc_qp_set_model(mpc_qp_solver* s, const int32_t* itab, int32_t n_i, const double* dtab, int32_t n_d) {
  if (!s) return -2;
  try {
    s->model = orc::Model();
    s->model.parse(itab, n_i, dtab, n_d);
    s->has_model = true;
    return 0;
  } catch (const std::exception& e) { s->err = e.what(); return -1; }
}

// The inverse-dynamics QP of QP_utils.py:120-158 built from recursive Newton-Euler evaluations only: nle = RNEA(q, v, 0),
// M e_k = RNEA(q, 0, e_k) - RNEA(q, 0, 0), the LOCAL frame Jacobian column k = frame velocity under v = e_k, the drift =
// frame spatial acceleration under (v, a = 0) without gravity.  (The HIP kernel uses composite inertias and world-frame
// Jacobian columns: a different route to the same matrices.)
int mpc_qp_solve_id(mpc_qp_solver* s, const mpc_qp_settings* S, int32_t nk, const int32_t* frames, const double* weights, const double* cone, double kd,
                    const double* xrob, const double* acc, const double* forces, const int32_t* contact_states,
                    double* x, double* y, double* z, mpc_qp_info* info, double* A_out, double* b_out, double* C_out, double* l_out) {
  if (!s) return -2;
  try {
    using namespace orc;
    if (!S || !frames || !weights || !cone || !xrob || !acc || !forces || !contact_states || !x || !info) throw std::runtime_error("qp_solve_id: null argument");
    if (!s->has_model) throw std::runtime_error("qp_solve_id: mpc_qp_set_model first");
    const Model& m = s->model;
    const mpc_qp_dims& d = s->d;
    const int nv = m.nv, nq = m.nq;
    if (nk <= 0 || d.n != 2 * nv - 6 + 6 * nk || d.neq != nv + 6 * nk || d.nin != 9 * nk || d.box)
      throw std::runtime_error("qp_solve_id: the handle's dimensions are not those of the inverse-dynamics QP");
    for (int c = 0; c < nk; ++c) if (frames[c] < 0 || frames[c] >= (int)m.frame_joint.size()) throw std::runtime_error("qp_solve_id: contact frame index out of range");
    const size_t B = d.batch, n = d.n, neq = d.neq, nin = d.nin;
    std::vector<double> H(B * n * n, 0.0), g(B * n, 0.0), A(B * neq * n, 0.0), b(B * neq, 0.0), C(B * nin * n, 0.0), l(B * nin, 0.0), u(B * nin, 1e5);
    for (size_t bi = 0; bi < B; ++bi) {
      double* Hb = H.data() + bi * n * n; double* Ab = A.data() + bi * neq * n; double* bb = b.data() + bi * neq;
      double* Cb = C.data() + bi * nin * n; double* lb = l.data() + bi * nin;
      for (int i = 0; i < nv; ++i) Hb[(size_t)i * n + i] = weights[0];
      for (int i = 0; i < 6 * nk; ++i) Hb[(size_t)(nv + i) * n + nv + i] = weights[1];
      const double* a0 = acc + bi * nv; const double* f0 = forces + bi * 6 * nk; const int32_t* cs = contact_states + bi * nk;
      State<double> st = state_from_x(m, xrob + bi * (nq + nv));
      Kin<double> k;
      std::vector<double> nle(nv), g0(nv), col(nv), e(nv, 0.0);
      forward_pass<double>(m, st, nullptr, k, true);
      rnea_backward<double>(m, k, nullptr, nle.data());
      State<double> s0 = st; std::fill(s0.v.begin(), s0.v.end(), 0.0);
      forward_pass<double>(m, s0, nullptr, k, true);
      rnea_backward<double>(m, k, nullptr, g0.data());
      std::vector<double> Mm((size_t)nv * nv);
      for (int c = 0; c < nv; ++c) {
        e[c] = 1.0;
        forward_pass<double>(m, s0, e.data(), k, true);
        rnea_backward<double>(m, k, nullptr, col.data());
        for (int r = 0; r < nv; ++r) Mm[(size_t)r * nv + c] = col[r] - g0[r];
        e[c] = 0.0;
      }
      std::vector<double> Jc((size_t)6 * nk * nv, 0.0), gamma(6 * nk, 0.0);
      for (int c = 0; c < nk; ++c) {
        if (!cs[c]) continue;
        const int fj = m.frame_joint[frames[c]];
        const SE3<double>& pl = m.frame_pl[frames[c]];
        for (int kk = 0; kk < nv; ++kk) {
          State<double> se = s0; se.v[kk] = 1.0;
          forward_pass<double>(m, se, nullptr, k, false);
          const Mot<double> vf = actInv(pl, k.v[fj]);
          for (int r = 0; r < 3; ++r) { Jc[(size_t)(6 * c + r) * nv + kk] = vf.lin[r]; Jc[(size_t)(6 * c + 3 + r) * nv + kk] = vf.ang[r]; }
        }
        forward_pass<double>(m, st, nullptr, k, false);
        const Mot<double> af = actInv(pl, k.a[fj]), vf = actInv(pl, k.v[fj]);
        for (int r = 0; r < 3; ++r) { gamma[6 * c + r] = af.lin[r] + kd * (vf.lin[r] + vf.ang[r]); gamma[6 * c + 3 + r] = af.ang[r]; }
      }
      for (int r = 0; r < nv; ++r) {
        double acc_r = -nle[r];
        for (int c = 0; c < nv; ++c) { Ab[(size_t)r * n + c] = Mm[(size_t)r * nv + c]; acc_r -= Mm[(size_t)r * nv + c] * a0[c]; }
        for (int c = 0; c < 6 * nk; ++c) { Ab[(size_t)r * n + nv + c] = -Jc[(size_t)c * nv + r]; acc_r += Jc[(size_t)c * nv + r] * f0[c]; }
        if (r >= 6) Ab[(size_t)r * n + nv + 6 * nk + r - 6] = -1.0;
        bb[r] = acc_r;
      }
      for (int c = 0; c < 6 * nk; ++c) {
        double v = -gamma[c];
        for (int kk = 0; kk < nv; ++kk) { Ab[(size_t)(nv + c) * n + kk] = Jc[(size_t)c * nv + kk]; v -= Jc[(size_t)c * nv + kk] * a0[kk]; }
        bb[nv + c] = v;
      }
      for (int c = 0; c < nk; ++c) {
        if (!cs[c]) continue;
        for (int r = 0; r < 9; ++r) {
          double v = 0;
          for (int j = 0; j < 6; ++j) { Cb[(size_t)(9 * c + r) * n + nv + 6 * c + j] = cone[6 * r + j]; v += cone[54 + 6 * r + j] * f0[6 * c + j]; }
          lb[9 * c + r] = -v;
        }
      }
    }
    if (A_out) std::memcpy(A_out, A.data(), A.size() * sizeof(double));
    if (b_out) std::memcpy(b_out, b.data(), b.size() * sizeof(double));
    if (C_out) std::memcpy(C_out, C.data(), C.size() * sizeof(double));
    if (l_out) std::memcpy(l_out, l.data(), l.size() * sizeof(double));
    return mpc_qp_solve(s, S, H.data(), g.data(), A.data(), b.data(), C.data(), l.data(), u.data(), nullptr, nullptr, x, y, z, nullptr, info);
  } catch (const std::exception& e) { s->err = e.what(); return -1; }
}

// IK + ID QP (QP_utils.py:584-762) built from recursive Newton-Euler evaluations, as mpc_qp_solve_id above: frame Jacobians as frame
// velocities under unit joint velocities, drifts as frame accelerations of the zero-acceleration motion without gravity, the centroidal
// momentum matrix as the momentum under unit joint velocities, its drift from the root force of that motion moved to the CoM.
int mpc_qp_solve_ikid(mpc_qp_solver* s, const mpc_qp_settings* S, int32_t nk, const int32_t* frames, int32_t base_frame, int32_t torso_frame,
                      const double* weights, const double* gains, const double* cone, const double* l_box, const double* u_box,
                      const double* xrob, const double* ik, const double* forces, const int32_t* contact_states,
                      double* x, double* y, double* z, double* z_box, mpc_qp_info* info,
                      double* H_out, double* g_out, double* A_out, double* b_out, double* C_out, double* l_out) {
  if (!s) return -2;
  try {
    using namespace orc;
    if (!S || !frames || !weights || !gains || !cone || !l_box || !u_box || !xrob || !ik || !forces || !contact_states || !x || !info) throw std::runtime_error("qp_solve_ikid: null argument");
    if (!s->has_model) throw std::runtime_error("qp_solve_ikid: mpc_qp_set_model first");
    const Model& m = s->model;
    const mpc_qp_dims& d = s->d;
    const int nv = m.nv, nq = m.nq, nkf = nk + 2;
    if (nk != 2 || d.n != 2 * nv - 6 + 6 * nk || d.neq != nv + 6 * nk || d.nin != 9 * nk || !d.box)
      throw std::runtime_error("qp_solve_ikid: two contacts and the handle's dimensions n = 2 nv - 6 + 6 nk, neq = nv + 6 nk, nin = 9 nk, box = 1 expected");
    std::vector<int> fr(nkf);
    for (int c = 0; c < nkf; ++c) {
      fr[c] = c < nk ? frames[c] : (c == nk ? base_frame : torso_frame);
      if (fr[c] < 0 || fr[c] >= (int)m.frame_joint.size()) throw std::runtime_error("qp_solve_ikid: frame index out of range");
    }
    const size_t B = d.batch, n = d.n, neq = d.neq, nin = d.nin, nik = 2 * nv + 42;
    const double *Kp0 = gains, *Kd0 = Kp0 + nv * nv, *Kp1 = Kd0 + nv * nv, *Kd1 = Kp1 + 36, *Kp3 = Kd1 + 36, *Kd3 = Kp3 + 9;
    std::vector<double> H(B * n * n, 0.0), g(B * n, 0.0), A(B * neq * n, 0.0), b(B * neq, 0.0), C(B * nin * n, 0.0), l(B * nin, 0.0), u(B * nin, 1e5), lb(B * n), ub(B * n);
    for (size_t bi = 0; bi < B; ++bi) {
      std::copy(l_box, l_box + n, lb.begin() + bi * n); std::copy(u_box, u_box + n, ub.begin() + bi * n);
      double* Hb = H.data() + bi * n * n; double* gb = g.data() + bi * n; double* Ab = A.data() + bi * neq * n; double* bb = b.data() + bi * neq;
      double* Cb = C.data() + bi * nin * n; double* lbv = l.data() + bi * nin;
      const double* f0 = forces + bi * 6 * nk; const int32_t* cs = contact_states + bi * nk; const double* e = ik + bi * nik;
      State<double> st = state_from_x(m, xrob + bi * (nq + nv));
      State<double> s0 = st; std::fill(s0.v.begin(), s0.v.end(), 0.0);
      Kin<double> k;
      std::vector<double> nle(nv), g0(nv), col(nv), unit(nv, 0.0), Mm((size_t)nv * nv);
      forward_pass<double>(m, st, nullptr, k, true); rnea_backward<double>(m, k, nullptr, nle.data());
      forward_pass<double>(m, s0, nullptr, k, true); rnea_backward<double>(m, k, nullptr, g0.data());
      for (int c = 0; c < nv; ++c) {
        unit[c] = 1.0;
        forward_pass<double>(m, s0, unit.data(), k, true); rnea_backward<double>(m, k, nullptr, col.data());
        for (int r = 0; r < nv; ++r) Mm[(size_t)r * nv + c] = col[r] - g0[r];
        unit[c] = 0.0;
      }
      // frame Jacobians (LOCAL), centroidal momentum matrix
      std::vector<double> Jf((size_t)6 * nkf * nv, 0.0), Ag((size_t)6 * nv, 0.0), drift(6 * nkf), dAgv(6);
      for (int kk = 0; kk < nv; ++kk) {
        State<double> se = s0; se.v[kk] = 1.0;
        forward_pass<double>(m, se, nullptr, k, false);
        for (int c = 0; c < nkf; ++c) {
          const Mot<double> vf = actInv(m.frame_pl[fr[c]], k.v[m.frame_joint[fr[c]]]);
          for (int r = 0; r < 3; ++r) { Jf[(size_t)(6 * c + r) * nv + kk] = vf.lin[r]; Jf[(size_t)(6 * c + 3 + r) * nv + kk] = vf.ang[r]; }
        }
        V3<double> com; Frc<double> hg;
        centroidal<double>(m, k, com, hg);
        for (int r = 0; r < 3; ++r) { Ag[(size_t)r * nv + kk] = hg.lin[r]; Ag[(size_t)(3 + r) * nv + kk] = hg.ang[r]; }
      }
      forward_pass<double>(m, st, nullptr, k, false);
      for (int c = 0; c < nkf; ++c) {
        const Mot<double> af = actInv(m.frame_pl[fr[c]], k.a[m.frame_joint[fr[c]]]);
        for (int r = 0; r < 3; ++r) { drift[6 * c + r] = af.lin[r]; drift[6 * c + 3 + r] = af.ang[r]; }
      }
      {
        std::vector<double> tau(nv);
        rnea_backward<double>(m, k, nullptr, tau.data());  // zero joint accelerations, no gravity: the base rows are the net force, base frame
        V3<double> com; Frc<double> hg;
        centroidal<double>(m, k, com, hg);
        Frc<double> fl; for (int r = 0; r < 3; ++r) { fl.lin[r] = tau[r]; fl.ang[r] = tau[3 + r]; }
        const Frc<double> fw = act(k.oMi[0], fl);
        const V3<double> ang = fw.ang - cross(com, fw.lin);
        for (int r = 0; r < 3; ++r) { dAgv[r] = fw.lin[r]; dAgv[3 + r] = ang[r]; }
      }
      // task targets
      std::vector<double> tt(6 * nkf + 6, 0.0), gp(nv, 0.0);
      for (int c = 0; c < nk; ++c) for (int r = 0; r < 6; ++r) {
        double v = drift[6 * c + r];
        for (int j = 0; j < 6; ++j) v -= Kp1[6 * r + j] * e[2 * nv + 12 * c + j] + Kd1[6 * r + j] * e[2 * nv + 12 * c + 6 + j];
        tt[6 * c + r] = v;
      }
      for (int c = nk; c < nkf; ++c) for (int r = 0; r < 3; ++r) {
        double v = drift[6 * c + 3 + r];
        const double* ee = e + 2 * nv + 12 * nk + 6 * (c - nk);
        for (int j = 0; j < 3; ++j) v -= Kp3[3 * r + j] * ee[j] + Kd3[3 * r + j] * ee[3 + j];
        tt[6 * c + 3 + r] = v;
      }
      for (int r = 0; r < 6; ++r) tt[6 * nkf + r] = -(e[2 * nv + 12 * nk + 12 + r] - dAgv[r]);
      for (int r = 0; r < nv; ++r) { double v = 0; for (int j = 0; j < nv; ++j) v -= Kp0[r * nv + j] * e[j] + Kd0[r * nv + j] * e[nv + j]; gp[r] = v; }
      // H, g
      for (int r = 0; r < nv; ++r) for (int c = 0; c < nv; ++c) {
        double sf = 0, sa = 0, so = 0;
        for (int rc = 0; rc < 6 * nk; ++rc) sf += Jf[(size_t)rc * nv + r] * Jf[(size_t)rc * nv + c];
        for (int q = 0; q < 6; ++q) sa += Ag[(size_t)q * nv + r] * Ag[(size_t)q * nv + c];
        for (int f = nk; f < nkf; ++f) for (int q = 3; q < 6; ++q) so += Jf[(size_t)(6 * f + q) * nv + r] * Jf[(size_t)(6 * f + q) * nv + c];
        Hb[(size_t)r * n + c] = (r == c ? weights[0] : 0.0) + weights[1] * sf + weights[2] * sa + weights[3] * so;
      }
      for (int i = 0; i < 6 * nk; ++i) Hb[(size_t)(nv + i) * n + nv + i] = weights[4];
      for (int c = 0; c < nv; ++c) {
        double sf = 0, sa = 0, so = 0;
        for (int rc = 0; rc < 6 * nk; ++rc) sf += tt[rc] * Jf[(size_t)rc * nv + c];
        for (int q = 0; q < 6; ++q) sa += tt[6 * nkf + q] * Ag[(size_t)q * nv + c];
        for (int f = nk; f < nkf; ++f) for (int q = 3; q < 6; ++q) so += tt[6 * f + q] * Jf[(size_t)(6 * f + q) * nv + c];
        gb[c] = weights[0] * gp[c] + weights[1] * sf + weights[2] * sa + weights[3] * so;
      }
      // A, b, C, l
      for (int r = 0; r < nv; ++r) {
        double v = -nle[r];
        for (int c = 0; c < nv; ++c) Ab[(size_t)r * n + c] = Mm[(size_t)r * nv + c];
        for (int rc = 0; rc < 6 * nk; ++rc) if (cs[rc / 6]) { Ab[(size_t)r * n + nv + rc] = -Jf[(size_t)rc * nv + r]; v += Jf[(size_t)rc * nv + r] * f0[rc]; }
        if (r >= 6) Ab[(size_t)r * n + nv + 6 * nk + r - 6] = -1.0;
        bb[r] = v;
      }
      for (int rc = 0; rc < 6 * nk; ++rc) if (cs[rc / 6]) {
        for (int kk = 0; kk < nv; ++kk) Ab[(size_t)(nv + rc) * n + kk] = Jf[(size_t)rc * nv + kk];
        bb[nv + rc] = -drift[rc];
      }
      for (int c = 0; c < nk; ++c) {
        if (!cs[c]) continue;
        for (int r = 0; r < 9; ++r) {
          double v = 0;
          for (int j = 0; j < 6; ++j) { Cb[(size_t)(9 * c + r) * n + nv + 6 * c + j] = cone[6 * r + j]; v += cone[54 + 6 * r + j] * f0[6 * c + j]; }
          lbv[9 * c + r] = -v;
        }
      }
    }
    auto out = [](double* dst, const std::vector<double>& v) { if (dst) std::memcpy(dst, v.data(), v.size() * sizeof(double)); };
    out(H_out, H); out(g_out, g); out(A_out, A); out(b_out, b); out(C_out, C); out(l_out, l);
    return mpc_qp_solve(s, S, H.data(), g.data(), A.data(), b.data(), C.data(), l.data(), u.data(), lb.data(), ub.data(), x, y, z, z_box, info);
  } catch (const std::exception& e) { s->err = e.what(); return -1; }
}

}  // extern "C"
