// oracle/model.hpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
// Kinematic-tree model parsed from the flat tables of include/mpc_abi.h, plus the primal rigid-body
// algorithms (recursive Newton-Euler in joint frames, Featherstone RBDA ch.5) templated on the
// scalar.  Restates what Pinocchio computes for the reference's models (pin.forwardKinematics,
// pin.rnea, pin.crba, pin.centerOfMass, pin.computeCentroidalMomentum — call sites in SURVEY.md §8b-2).
#pragma once
#include <stdexcept>
#include <vector>
#include "../include/mpc_abi.h"
#include "spatial.hpp"

namespace orc {

struct ContactModel {
  int joint;
  SE3<double> pl1, pl2;  // joint1_placement, joint2_placement (world side), fulldynamic_talos.py:82-92
  double Kp[6], Kd[6];   // Baumgarte corrector, fulldynamic_talos.py:93-94
};

struct Model {
  int nj = 0, nq = 0, nv = 0;
  std::vector<int> parent, kind, idx_q, idx_v;
  std::vector<SE3<double>> placement;
  std::vector<Inertia<double>> inertia;
  std::vector<int> frame_joint;
  std::vector<SE3<double>> frame_pl;
  std::vector<ContactModel> contacts;
  double gravity[3] = {0, 0, -9.81};
  double prox_mu = 0.0;
  double total_mass = 0.0;

  static SE3<double> read_se3(const double* d) {
    SE3<double> M;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M.R(i, j) = d[3 * i + j];
    for (int i = 0; i < 3; ++i) M.p[i] = d[9 + i];
    return M;
  }

  void parse(const int32_t* it, int n_i, const double* dt, int n_d) {
    if (n_i < MPC_MODEL_HEADER_WORDS) throw std::runtime_error("model table too short");
    nj = it[0]; nq = it[1]; nv = it[2];
    const int nf = it[3], ncn = it[4];
    const int need_i = MPC_MODEL_HEADER_WORDS + MPC_MODEL_JOINT_WORDS * nj + nf + ncn;
    const int need_d = MPC_MODEL_HEADER_DOUBLES + MPC_MODEL_JOINT_DOUBLES * nj + MPC_MODEL_FRAME_DOUBLES * nf + MPC_MODEL_CONTACT_DOUBLES * ncn;
    if (n_i < need_i || n_d < need_d) throw std::runtime_error("model table size mismatch");
    const int32_t* ip = it + MPC_MODEL_HEADER_WORDS;
    const double* dp = dt;
    for (int i = 0; i < 3; ++i) gravity[i] = dp[i];
    prox_mu = dp[3];
    dp += MPC_MODEL_HEADER_DOUBLES;
    parent.resize(nj); kind.resize(nj); idx_q.resize(nj); idx_v.resize(nj); placement.resize(nj); inertia.resize(nj);
    total_mass = 0;
    for (int i = 0; i < nj; ++i) {
      parent[i] = ip[0]; kind[i] = ip[1]; idx_q[i] = ip[2]; idx_v[i] = ip[3]; ip += MPC_MODEL_JOINT_WORDS;
      if (parent[i] >= i) throw std::runtime_error("model joints must be topologically ordered");
      if (kind[i] == MPC_JOINT_FREEFLYER && i != 0) throw std::runtime_error("free-flyer only supported as the root joint");
      placement[i] = read_se3(dp);
      inertia[i].mass = dp[12];
      for (int k = 0; k < 3; ++k) inertia[i].c[k] = dp[13 + k];
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) inertia[i].I(r, c) = dp[16 + 3 * r + c];
      total_mass += inertia[i].mass;
      dp += MPC_MODEL_JOINT_DOUBLES;
    }
    frame_joint.resize(nf); frame_pl.resize(nf);
    for (int f = 0; f < nf; ++f) { frame_joint[f] = *ip++; frame_pl[f] = read_se3(dp); dp += MPC_MODEL_FRAME_DOUBLES; }
    contacts.resize(ncn);
    for (int c = 0; c < ncn; ++c) {
      contacts[c].joint = *ip++;
      contacts[c].pl1 = read_se3(dp); contacts[c].pl2 = read_se3(dp + 12);
      for (int k = 0; k < 6; ++k) { contacts[c].Kp[k] = dp[24 + k]; contacts[c].Kd[k] = dp[30 + k]; }
      dp += MPC_MODEL_CONTACT_DOUBLES;
    }
  }
  bool has_freeflyer() const { return nj > 0 && kind[0] == MPC_JOINT_FREEFLYER; }
};

// configuration + velocity in "tangent-indexed" form: base placement separately, revolute angles at idx_v
template <class T>
struct State {
  SE3<T> base;
  std::vector<T> qa;  // size nv; entries of a free-flyer are unused
  std::vector<T> v;   // size nv
};

template <class T> SE3<T> convert(const SE3<double>& M) {
  SE3<T> r;
  for (int i = 0; i < 3; ++i) { r.p[i] = T(M.p[i]); for (int j = 0; j < 3; ++j) r.R(i, j) = T(M.R(i, j)); }
  return r;
}
template <class T> Inertia<T> convert(const Inertia<double>& Y) {
  Inertia<T> r;
  r.mass = T(Y.mass);
  for (int i = 0; i < 3; ++i) { r.c[i] = T(Y.c[i]); for (int j = 0; j < 3; ++j) r.I(i, j) = T(Y.I(i, j)); }
  return r;
}

// x = [q; v] (Pinocchio layout, free-flyer quaternion xyzw) -> State<double>
inline State<double> state_from_x(const Model& m, const double* x) {
  State<double> s;
  s.qa.assign(m.nv, 0.0);
  s.v.assign(x + m.nq, x + m.nq + m.nv);
  for (int i = 0; i < m.nj; ++i) {
    if (m.kind[i] == MPC_JOINT_FREEFLYER) {
      const double* q = x + m.idx_q[i];
      const double n = std::sqrt(q[3] * q[3] + q[4] * q[4] + q[5] * q[5] + q[6] * q[6]);
      s.base.R = quat_to_rot<double>(q[3] / n, q[4] / n, q[5] / n, q[6] / n);
      for (int k = 0; k < 3; ++k) s.base.p[k] = q[k];
    } else {
      s.qa[m.idx_v[i]] = x[m.idx_q[i]];
    }
  }
  return s;
}
inline void x_from_state(const Model& m, const State<double>& s, double* x) {
  for (int i = 0; i < m.nj; ++i) {
    if (m.kind[i] == MPC_JOINT_FREEFLYER) {
      for (int k = 0; k < 3; ++k) x[m.idx_q[i] + k] = s.base.p[k];
      rot_to_quat(s.base.R, x + m.idx_q[i] + 3);
    } else {
      x[m.idx_q[i]] = s.qa[m.idx_v[i]];
    }
  }
  for (int k = 0; k < m.nv; ++k) x[m.nq + k] = s.v[k];
}

// Lift a double state to T, optionally seeding tangent directions: direction k < nv perturbs the
// configuration along e_k (right-multiplication on the free-flyer, Pinocchio's integrate), direction
// nv + k perturbs v_k.  `seed` = first dual slot to use, or -1 for no seeding.
template <class T> State<T> lift(const Model& m, const State<double>& s, int seed);
template <> inline State<double> lift<double>(const Model&, const State<double>& s, int) { return s; }
template <> inline State<Dual> lift<Dual>(const Model& m, const State<double>& s, int seed) {
  State<Dual> r;
  r.base = convert<Dual>(s.base);
  r.qa.resize(m.nv); r.v.resize(m.nv);
  for (int k = 0; k < m.nv; ++k) { r.qa[k] = Dual(s.qa[k]); r.v[k] = Dual(s.v[k]); }
  if (seed < 0) return r;
  for (int k = 0; k < m.nv; ++k) { r.qa[k].d[seed + k] = 1.0; r.v[k].d[seed + m.nv + k] = 1.0; }
  if (m.has_freeflyer()) {
    // d/de [ M exp6(e) ] at 0:  dp = R e_lin ,  dR = R [e_ang]x
    for (int k = 0; k < 3; ++k) {
      r.qa[k].d[seed + k] = 0.0; r.qa[3 + k].d[seed + 3 + k] = 0.0;
      for (int i = 0; i < 3; ++i) r.base.p[i].d[seed + k] = s.base.R(i, k);
      V3<double> e; e[k] = 1.0;
      const M3<double> dR = s.base.R * skew(e);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.base.R(i, j).d[seed + 3 + k] = dR(i, j);
    }
  }
  return r;
}

template <class T> SE3<T> joint_transform(int kind, const T& q) {
  SE3<T> M;
  const T c = cos(q), s = sin(q);
  const int a = kind - MPC_JOINT_RX, b = (a + 1) % 3, d = (a + 2) % 3;
  M.R(b, b) = c; M.R(b, d) = -s; M.R(d, b) = s; M.R(d, d) = c;
  return M;
}

template <class T>
struct Kin {
  std::vector<SE3<T>> liMi, oMi;
  std::vector<Mot<T>> v, a;  // joint-frame spatial velocity / acceleration (a includes the gravity field)
};

// forward kinematics + velocities + accelerations (with the -gravity trick), joint frames
template <class T>
void forward_pass(const Model& m, const State<T>& s, const T* qdd, Kin<T>& k, bool with_gravity = true) {
  k.liMi.resize(m.nj); k.oMi.resize(m.nj); k.v.resize(m.nj); k.a.resize(m.nj);
  Mot<T> a0;
  if (with_gravity) a0.lin = V3<T>(T(-m.gravity[0]), T(-m.gravity[1]), T(-m.gravity[2]));
  for (int i = 0; i < m.nj; ++i) {
    const int iv = m.idx_v[i];
    Mot<T> vJ, aJ;
    if (m.kind[i] == MPC_JOINT_FREEFLYER) {
      k.liMi[i] = convert<T>(m.placement[i]) * s.base;
      vJ.lin = V3<T>(s.v[iv], s.v[iv + 1], s.v[iv + 2]); vJ.ang = V3<T>(s.v[iv + 3], s.v[iv + 4], s.v[iv + 5]);
      if (qdd) { aJ.lin = V3<T>(qdd[iv], qdd[iv + 1], qdd[iv + 2]); aJ.ang = V3<T>(qdd[iv + 3], qdd[iv + 4], qdd[iv + 5]); }
    } else {
      k.liMi[i] = convert<T>(m.placement[i]) * joint_transform<T>(m.kind[i], s.qa[iv]);
      vJ.ang[m.kind[i] - MPC_JOINT_RX] = s.v[iv];
      if (qdd) aJ.ang[m.kind[i] - MPC_JOINT_RX] = qdd[iv];
    }
    const int p = m.parent[i];
    if (p < 0) {
      k.oMi[i] = k.liMi[i];
      k.v[i] = vJ;
      k.a[i] = actInv(k.liMi[i], a0) + aJ;
    } else {
      k.oMi[i] = k.oMi[p] * k.liMi[i];
      k.v[i] = actInv(k.liMi[i], k.v[p]) + vJ;
      k.a[i] = actInv(k.liMi[i], k.a[p]) + aJ + mcross(k.v[i], vJ);
    }
  }
}

// tau = RNEA given a finished forward pass; fext[i] (may be null) = external force on body i, joint frame
template <class T>
void rnea_backward(const Model& m, const Kin<T>& k, const std::vector<Frc<T>>* fext, T* tau) {
  std::vector<Frc<T>> f(m.nj);
  for (int i = 0; i < m.nj; ++i) {
    const Inertia<T> Y = convert<T>(m.inertia[i]);
    f[i] = Y * k.a[i] + fcross(k.v[i], Y * k.v[i]);
    if (fext) f[i] = f[i] - (*fext)[i];
  }
  for (int i = m.nj - 1; i >= 0; --i) {
    const int iv = m.idx_v[i];
    if (m.kind[i] == MPC_JOINT_FREEFLYER) {
      for (int c = 0; c < 3; ++c) { tau[iv + c] = f[i].lin[c]; tau[iv + 3 + c] = f[i].ang[c]; }
    } else {
      tau[iv] = f[i].ang[m.kind[i] - MPC_JOINT_RX];
    }
    if (m.parent[i] >= 0) f[m.parent[i]] = f[m.parent[i]] + act(k.liMi[i], f[i]);
  }
}

// centre of mass (world) and centroidal momentum hg = [linear; angular about the CoM] (world axes)
template <class T>
void centroidal(const Model& m, const Kin<T>& k, V3<T>& com, Frc<T>& hg) {
  Frc<T> h0;
  V3<T> mc;
  for (int i = 0; i < m.nj; ++i) {
    const Inertia<T> Y = convert<T>(m.inertia[i]);
    h0 = h0 + act(k.oMi[i], Y * k.v[i]);
    mc = mc + Y.mass * (k.oMi[i].R * Y.c + k.oMi[i].p);
  }
  com = T(1.0 / m.total_mass) * mc;
  hg.lin = h0.lin;
  hg.ang = h0.ang - cross(com, h0.lin);
}

}  // namespace orc
