// oracle/spatial.hpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
// Spatial algebra with Pinocchio's conventions ([linear; angular] motions, [force; torque] forces,
// SE3 right-multiplication for the free-flyer), templated on the scalar so the same primal code runs
// on double and on orc::Dual.  Restates the operations pin.* performs underneath the reference's
// residual/dynamics models (README.md:16); formulas from Featherstone's RBDA and Murray-Li-Sastry.
#pragma once
#include <cmath>
#include "dual.hpp"

namespace orc {

using std::atan2;
using std::cos;
using std::sin;
using std::sqrt;

template <class T>
struct V3 {
  T x[3];
  V3() { x[0] = x[1] = x[2] = T(0.0); }
  V3(const T& a, const T& b, const T& c) { x[0] = a; x[1] = b; x[2] = c; }
  T& operator[](int i) { return x[i]; }
  const T& operator[](int i) const { return x[i]; }
};
template <class T> V3<T> operator+(const V3<T>& a, const V3<T>& b) { return V3<T>(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
template <class T> V3<T> operator-(const V3<T>& a, const V3<T>& b) { return V3<T>(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
template <class T> V3<T> operator-(const V3<T>& a) { return V3<T>(-a[0], -a[1], -a[2]); }
template <class T> V3<T> operator*(const T& s, const V3<T>& a) { return V3<T>(s * a[0], s * a[1], s * a[2]); }
template <class T> T dot(const V3<T>& a, const V3<T>& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> V3<T> cross(const V3<T>& a, const V3<T>& b) {
  return V3<T>(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}

template <class T>
struct M3 {
  T m[3][3];
  M3() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = T(0.0); }
  static M3 identity() { M3 r; for (int i = 0; i < 3; ++i) r.m[i][i] = T(1.0); return r; }
  T& operator()(int i, int j) { return m[i][j]; }
  const T& operator()(int i, int j) const { return m[i][j]; }
};
template <class T> M3<T> operator*(const M3<T>& a, const M3<T>& b) {
  M3<T> r;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T s(0.0); for (int k = 0; k < 3; ++k) s += a(i, k) * b(k, j); r(i, j) = s; }
  return r;
}
template <class T> V3<T> operator*(const M3<T>& a, const V3<T>& v) {
  V3<T> r;
  for (int i = 0; i < 3; ++i) { T s(0.0); for (int k = 0; k < 3; ++k) s += a(i, k) * v[k]; r[i] = s; }
  return r;
}
template <class T> M3<T> transpose(const M3<T>& a) { M3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = a(j, i); return r; }
template <class T> V3<T> tmul(const M3<T>& a, const V3<T>& v) {  // a^T v
  V3<T> r;
  for (int i = 0; i < 3; ++i) { T s(0.0); for (int k = 0; k < 3; ++k) s += a(k, i) * v[k]; r[i] = s; }
  return r;
}
template <class T> M3<T> skew(const V3<T>& w) {
  M3<T> K;
  K(0, 1) = -w[2]; K(0, 2) = w[1]; K(1, 0) = w[2]; K(1, 2) = -w[0]; K(2, 0) = -w[1]; K(2, 1) = w[0];
  return K;
}

// ---- SO(3) / SE(3) exp and log -----------------------------------------------------------------
// Coefficients are evaluated from theta^2 with a Taylor branch below kSmall so that the dual parts
// stay finite at theta = 0 (sqrt is never differentiated at 0).
constexpr double kSmall2 = 1e-3;  // theta^2 threshold (theta ~ 0.0316)

template <class T> M3<T> exp3(const V3<T>& w) {
  const T t2 = dot(w, w);
  T A, B;
  if (value(t2) < kSmall2) {
    A = T(1.0) - t2 * (T(1.0 / 6) - t2 * (T(1.0 / 120) - t2 * T(1.0 / 5040)));
    B = T(0.5) - t2 * (T(1.0 / 24) - t2 * (T(1.0 / 720) - t2 * T(1.0 / 40320)));
  } else {
    const T t = sqrt(t2);
    const T sh = sin(T(0.5) * t);
    A = sin(t) / t;
    B = T(2.0) * sh * sh / t2;
  }
  const M3<T> K = skew(w), K2 = K * K;
  M3<T> R = M3<T>::identity();
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R(i, j) = R(i, j) + A * K(i, j) + B * K2(i, j);
  return R;
}

template <class T> V3<T> log3(const M3<T>& R) {
  const T half(0.5);
  V3<T> v(half * (R(2, 1) - R(1, 2)), half * (R(0, 2) - R(2, 0)), half * (R(1, 0) - R(0, 1)));  // sin(th) * axis
  const T c = half * (R(0, 0) + R(1, 1) + R(2, 2) - T(1.0));
  const T s2 = dot(v, v);
  T f;  // theta / sin(theta)
  if (value(s2) < kSmall2 && value(c) > 0.0) {
    f = T(1.0) + s2 * (T(1.0 / 6) + s2 * (T(3.0 / 40) + s2 * (T(15.0 / 336) + s2 * T(105.0 / 3456))));
  } else {
    const T s = sqrt(s2);
    f = atan2(s, c) / s;
  }
  return f * v;
}

template <class T>
struct SE3 {
  M3<T> R;
  V3<T> p;
  SE3() : R(M3<T>::identity()) {}
  SE3(const M3<T>& R_, const V3<T>& p_) : R(R_), p(p_) {}
};
template <class T> SE3<T> operator*(const SE3<T>& a, const SE3<T>& b) { return SE3<T>(a.R * b.R, a.R * b.p + a.p); }
template <class T> SE3<T> inverse(const SE3<T>& a) { return SE3<T>(transpose(a.R), -tmul(a.R, a.p)); }

// spatial motion [lin; ang] and force [f; n]
template <class T> struct Mot { V3<T> lin, ang; };
template <class T> struct Frc { V3<T> lin, ang; };
template <class T> Mot<T> operator+(const Mot<T>& a, const Mot<T>& b) { return Mot<T>{a.lin + b.lin, a.ang + b.ang}; }
template <class T> Mot<T> operator-(const Mot<T>& a, const Mot<T>& b) { return Mot<T>{a.lin - b.lin, a.ang - b.ang}; }
template <class T> Frc<T> operator+(const Frc<T>& a, const Frc<T>& b) { return Frc<T>{a.lin + b.lin, a.ang + b.ang}; }
template <class T> Frc<T> operator-(const Frc<T>& a, const Frc<T>& b) { return Frc<T>{a.lin - b.lin, a.ang - b.ang}; }

// aMb acting on a motion expressed in b -> expressed in a
template <class T> Mot<T> act(const SE3<T>& M, const Mot<T>& m) {
  const V3<T> w = M.R * m.ang;
  return Mot<T>{M.R * m.lin + cross(M.p, w), w};
}
template <class T> Mot<T> actInv(const SE3<T>& M, const Mot<T>& m) {
  return Mot<T>{tmul(M.R, m.lin - cross(M.p, m.ang)), tmul(M.R, m.ang)};
}
template <class T> Frc<T> act(const SE3<T>& M, const Frc<T>& f) {
  const V3<T> fl = M.R * f.lin;
  return Frc<T>{fl, M.R * f.ang + cross(M.p, fl)};
}
template <class T> Frc<T> actInv(const SE3<T>& M, const Frc<T>& f) {
  return Frc<T>{tmul(M.R, f.lin), tmul(M.R, f.ang - cross(M.p, f.lin))};
}
// motion x motion, motion x* force
template <class T> Mot<T> mcross(const Mot<T>& a, const Mot<T>& b) {
  return Mot<T>{cross(a.ang, b.lin) + cross(a.lin, b.ang), cross(a.ang, b.ang)};
}
template <class T> Frc<T> fcross(const Mot<T>& a, const Frc<T>& f) {
  return Frc<T>{cross(a.ang, f.lin), cross(a.ang, f.ang) + cross(a.lin, f.lin)};
}

template <class T> SE3<T> exp6(const Mot<T>& nu) {
  const V3<T>& w = nu.ang;
  const T t2 = dot(w, w);
  T B, C;
  if (value(t2) < kSmall2) {
    B = T(0.5) - t2 * (T(1.0 / 24) - t2 * (T(1.0 / 720) - t2 * T(1.0 / 40320)));
    C = T(1.0 / 6) - t2 * (T(1.0 / 120) - t2 * (T(1.0 / 5040) - t2 * T(1.0 / 362880)));
  } else {
    const T t = sqrt(t2);
    const T sh = sin(T(0.5) * t);
    B = T(2.0) * sh * sh / t2;
    C = (t - sin(t)) / (t2 * t);
  }
  const M3<T> K = skew(w), K2 = K * K;
  V3<T> p = nu.lin + B * (K * nu.lin) + C * (K2 * nu.lin);
  return SE3<T>(exp3(w), p);
}

template <class T> Mot<T> log6(const SE3<T>& M) {
  const V3<T> w = log3(M.R);
  const T t2 = dot(w, w);
  T C;  // (1/t^2) (1 - t sin t / (2 (1 - cos t)))
  if (value(t2) < kSmall2) {
    C = T(1.0 / 12) + t2 * (T(1.0 / 720) + t2 * (T(1.0 / 30240) + t2 * T(1.0 / 1209600)));
  } else {
    const T t = sqrt(t2);
    const T sh = sin(T(0.5) * t), ch = cos(T(0.5) * t);
    // t sin t / (2(1-cos t)) = t cos(t/2) / (2 sin(t/2))
    C = (T(1.0) - t * ch / (T(2.0) * sh)) / t2;
  }
  const M3<T> K = skew(w), K2 = K * K;
  V3<T> v = M.p - T(0.5) * (K * M.p) + C * (K2 * M.p);
  return Mot<T>{v, w};
}

// rotational inertia about the CoM + mass + CoM ("lever") in the joint frame
template <class T>
struct Inertia {
  T mass;
  V3<T> c;
  M3<T> I;
};
template <class T> Frc<T> operator*(const Inertia<T>& Y, const Mot<T>& v) {
  const V3<T> hl = Y.mass * (v.lin + cross(v.ang, Y.c));
  return Frc<T>{hl, Y.I * v.ang + cross(Y.c, hl)};
}

template <class T> M3<T> quat_to_rot(const T& x, const T& y, const T& z, const T& w) {
  M3<T> R;
  const T two(2.0), one(1.0);
  R(0, 0) = one - two * (y * y + z * z); R(0, 1) = two * (x * y - z * w); R(0, 2) = two * (x * z + y * w);
  R(1, 0) = two * (x * y + z * w); R(1, 1) = one - two * (x * x + z * z); R(1, 2) = two * (y * z - x * w);
  R(2, 0) = two * (x * z - y * w); R(2, 1) = two * (y * z + x * w); R(2, 2) = one - two * (x * x + y * y);
  return R;
}

inline void rot_to_quat(const M3<double>& R, double q[4]) {  // x y z w, w >= 0
  const double t = R(0, 0) + R(1, 1) + R(2, 2);
  double x, y, z, w;
  if (t > 0) {
    const double s = std::sqrt(t + 1.0) * 2; w = 0.25 * s; x = (R(2, 1) - R(1, 2)) / s; y = (R(0, 2) - R(2, 0)) / s; z = (R(1, 0) - R(0, 1)) / s;
  } else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) {
    const double s = std::sqrt(1.0 + R(0, 0) - R(1, 1) - R(2, 2)) * 2; w = (R(2, 1) - R(1, 2)) / s; x = 0.25 * s; y = (R(0, 1) + R(1, 0)) / s; z = (R(0, 2) + R(2, 0)) / s;
  } else if (R(1, 1) > R(2, 2)) {
    const double s = std::sqrt(1.0 + R(1, 1) - R(0, 0) - R(2, 2)) * 2; w = (R(0, 2) - R(2, 0)) / s; x = (R(0, 1) + R(1, 0)) / s; y = 0.25 * s; z = (R(1, 2) + R(2, 1)) / s;
  } else {
    const double s = std::sqrt(1.0 + R(2, 2) - R(0, 0) - R(1, 1)) * 2; w = (R(1, 0) - R(0, 1)) / s; x = (R(0, 2) + R(2, 0)) / s; y = (R(1, 2) + R(2, 1)) / s; z = 0.25 * s;
  }
  const double n = std::sqrt(x * x + y * y + z * z + w * w) * (w < 0 ? -1.0 : 1.0);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}

}  // namespace orc
