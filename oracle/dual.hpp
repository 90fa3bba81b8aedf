// oracle/dual.hpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
// Vector-mode forward automatic differentiation: a value plus up to MAXD directional derivatives.
// The oracle obtains every Jacobian of the rigid-body residuals by running the *primal* algorithms on
// this type (restating what pin.computeConstraintDynamicsDerivatives / aligator residual
// computeJacobians produce — SURVEY.md §8a-2 K2, K4 — without sharing any closed-form derivative code
// with the HIP kernels it is used to check).
#pragma once
#include <cmath>

namespace orc {

constexpr int MAXD = 128;  // >= 2*nv + nu: complete Talos full dynamics 76 + 32, kinodynamics 76 + 44

// number of active tangent directions of the running AD sweep (per thread)
inline int& dual_nd() {
  static thread_local int nd = 0;
  return nd;
}

struct Dual {
  double v;
  double d[MAXD];
  Dual() : v(0.0) {
    const int n = dual_nd();
    for (int i = 0; i < n; ++i) d[i] = 0.0;
  }
  Dual(double x) : v(x) {
    const int n = dual_nd();
    for (int i = 0; i < n; ++i) d[i] = 0.0;
  }
  static Dual seed(double x, int dir) {
    Dual r(x);
    r.d[dir] = 1.0;
    return r;
  }
};

inline Dual operator+(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v + b.v;
  const int n = dual_nd();
  for (int i = 0; i < n; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
inline Dual operator-(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v - b.v;
  const int n = dual_nd();
  for (int i = 0; i < n; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
inline Dual operator-(const Dual& a) {
  Dual r;
  r.v = -a.v;
  const int n = dual_nd();
  for (int i = 0; i < n; ++i) r.d[i] = -a.d[i];
  return r;
}
inline Dual operator*(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v * b.v;
  const int n = dual_nd();
  for (int i = 0; i < n; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
inline Dual operator/(const Dual& a, const Dual& b) {
  Dual r;
  const double inv = 1.0 / b.v;
  r.v = a.v * inv;
  const int n = dual_nd();
  for (int i = 0; i < n; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
inline Dual& operator+=(Dual& a, const Dual& b) { return a = a + b; }
inline Dual& operator-=(Dual& a, const Dual& b) { return a = a - b; }
inline Dual& operator*=(Dual& a, const Dual& b) { return a = a * b; }

inline Dual chain(const Dual& a, double f, double df) {
  Dual r;
  r.v = f;
  const int n = dual_nd();
  for (int i = 0; i < n; ++i) r.d[i] = df * a.d[i];
  return r;
}
inline Dual sin(const Dual& a) { return chain(a, std::sin(a.v), std::cos(a.v)); }
inline Dual cos(const Dual& a) { return chain(a, std::cos(a.v), -std::sin(a.v)); }
inline Dual sqrt(const Dual& a) {
  const double s = std::sqrt(a.v);
  return chain(a, s, 0.5 / s);
}
inline Dual atan2(const Dual& y, const Dual& x) {
  Dual r;
  r.v = std::atan2(y.v, x.v);
  const double den = x.v * x.v + y.v * y.v;
  const int n = dual_nd();
  for (int i = 0; i < n; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) / den;
  return r;
}

inline double value(double x) { return x; }
inline double value(const Dual& x) { return x.v; }

}  // namespace orc
