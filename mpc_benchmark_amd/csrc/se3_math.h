// se3_math.h — SO(3) / SE(3) and spatial 6-vector helpers shared by the HIP kernels (device_common.h: DEV = __device__
// __forceinline__) and by the CPU port of the stage evaluation that bench.py times as its cpu_baseline (oracle/cpu_port/: DEV =
// static inline).  Right Jacobians as used by Pinocchio's Jlog6 / Jexp6 (Barfoot's Q block), Taylor branches below theta^2 = 1e-3.
#pragma once
#include <math.h>
#ifndef DEV
#error "define DEV (function qualifier) before including se3_math.h"
#endif

// ---- small SO(3)/SE(3) helpers (multibody state space) --------------------------------------------
struct V3 { double x, y, z; };
DEV V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
struct M3 { double m[9]; };
DEV V3 mul(const M3& R, V3 v) { return v3(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z); }
DEV V3 tmul(const M3& R, V3 v) { return v3(R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z, R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z); }
DEV M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
DEV M3 tmul(const M3& A, const M3& B) {  // A^T B
  M3 C;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[i] * B.m[j] + A.m[3 + i] * B.m[3 + j] + A.m[6 + i] * B.m[6 + j];
  return C;
}
DEV M3 quat_to_rot(const double* q) {  // x y z w
  const double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double x = q[0] / nrm, y = q[1] / nrm, z = q[2] / nrm, w = q[3] / nrm;
  M3 R;
  R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - z * w); R.m[2] = 2 * (x * z + y * w);
  R.m[3] = 2 * (x * y + z * w); R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - x * w);
  R.m[6] = 2 * (x * z - y * w); R.m[7] = 2 * (y * z + x * w); R.m[8] = 1 - 2 * (x * x + y * y);
  return R;
}
DEV void rot_to_quat(const M3& R, double* q) {
  const double t = R.m[0] + R.m[4] + R.m[8];
  double x, y, z, w;
  if (t > 0) { const double s = sqrt(t + 1.0) * 2; w = 0.25 * s; x = (R.m[7] - R.m[5]) / s; y = (R.m[2] - R.m[6]) / s; z = (R.m[3] - R.m[1]) / s; }
  else if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) { const double s = sqrt(1.0 + R.m[0] - R.m[4] - R.m[8]) * 2; w = (R.m[7] - R.m[5]) / s; x = 0.25 * s; y = (R.m[1] + R.m[3]) / s; z = (R.m[2] + R.m[6]) / s; }
  else if (R.m[4] > R.m[8]) { const double s = sqrt(1.0 + R.m[4] - R.m[0] - R.m[8]) * 2; w = (R.m[2] - R.m[6]) / s; x = (R.m[1] + R.m[3]) / s; y = 0.25 * s; z = (R.m[5] + R.m[7]) / s; }
  else { const double s = sqrt(1.0 + R.m[8] - R.m[0] - R.m[4]) * 2; w = (R.m[3] - R.m[1]) / s; x = (R.m[2] + R.m[6]) / s; y = (R.m[5] + R.m[7]) / s; z = 0.25 * s; }
  const double n = sqrt(x * x + y * y + z * z + w * w) * (w < 0 ? -1.0 : 1.0);
  q[0] = x / n; q[1] = y / n; q[2] = z / n; q[3] = w / n;
}
DEV M3 skew_m(V3 w) { M3 K; K.m[0] = 0; K.m[1] = -w.z; K.m[2] = w.y; K.m[3] = w.z; K.m[4] = 0; K.m[5] = -w.x; K.m[6] = -w.y; K.m[7] = w.x; K.m[8] = 0; return K; }

constexpr double kSmall2 = 1e-3;  // theta^2 switch to Taylor series (same switch point as the oracle)

// sin(t)/t, (1-cos t)/t^2, (t - sin t)/t^3 from t^2
DEV void so3_coeffs(double t2, double& A, double& B, double& C) {
  if (t2 < kSmall2) {
    A = 1.0 - t2 * (1.0 / 6 - t2 * (1.0 / 120 - t2 * (1.0 / 5040)));
    B = 0.5 - t2 * (1.0 / 24 - t2 * (1.0 / 720 - t2 * (1.0 / 40320)));
    C = 1.0 / 6 - t2 * (1.0 / 120 - t2 * (1.0 / 5040 - t2 * (1.0 / 362880)));
  } else {
    const double t = sqrt(t2), sh = sin(0.5 * t);
    A = sin(t) / t; B = 2.0 * sh * sh / t2; C = (t - sin(t)) / (t2 * t);
  }
}
DEV M3 exp3(V3 w) {
  double A, B, C;
  so3_coeffs(dot(w, w), A, B, C);
  const M3 K = skew_m(w), K2 = mul(K, K);
  M3 R;
  for (int i = 0; i < 9; ++i) R.m[i] = ((i % 4 == 0) ? 1.0 : 0.0) + A * K.m[i] + B * K2.m[i];
  return R;
}
DEV V3 log3(const M3& R) {
  const V3 v = v3(0.5 * (R.m[7] - R.m[5]), 0.5 * (R.m[2] - R.m[6]), 0.5 * (R.m[3] - R.m[1]));
  const double c = 0.5 * (R.m[0] + R.m[4] + R.m[8] - 1.0), s2 = dot(v, v);
  double f;
  if (s2 < kSmall2 && c > 0.0) f = 1.0 + s2 * (1.0 / 6 + s2 * (3.0 / 40 + s2 * (15.0 / 336 + s2 * (105.0 / 3456))));
  else { const double s = sqrt(s2); f = atan2(s, c) / s; }
  return f * v;
}
// exp6 of nu = (v, w) -> (R, p)
DEV void exp6(V3 v, V3 w, M3& R, V3& p) {
  double A, B, C;
  so3_coeffs(dot(w, w), A, B, C);
  R = exp3(w);
  const V3 wv = cross(w, v);
  p = v + B * wv + C * cross(w, wv);
}
// log6 of (R, p) -> (v, w)
DEV void log6(const M3& R, V3 p, V3& v, V3& w) {
  w = log3(R);
  const double t2 = dot(w, w);
  double Cc;
  if (t2 < kSmall2) Cc = 1.0 / 12 + t2 * (1.0 / 720 + t2 * (1.0 / 30240 + t2 * (1.0 / 1209600)));
  else { const double t = sqrt(t2), sh = sin(0.5 * t), ch = cos(0.5 * t); Cc = (1.0 - t * ch / (2.0 * sh)) / t2; }
  const V3 wp = cross(w, p);
  v = p - 0.5 * wp + Cc * cross(w, wp);
}


// ---- 6-vectors --------------------------------------------------------------------------------------------
struct S6 { double v[6]; };
DEV S6 ld6(const double* p) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = p[i]; return r; }
DEV void st6(double* p, const S6& a) { for (int i = 0; i < 6; ++i) p[i] = a.v[i]; }
DEV S6 zero6() { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = 0.0; return r; }
DEV S6 add6(const S6& a, const S6& b) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = a.v[i] + b.v[i]; return r; }
DEV S6 sub6(const S6& a, const S6& b) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = a.v[i] - b.v[i]; return r; }
DEV S6 scale6(double s, const S6& a) { S6 r; for (int i = 0; i < 6; ++i) r.v[i] = s * a.v[i]; return r; }
DEV double dot6(const S6& a, const S6& b) { double s = 0; for (int i = 0; i < 6; ++i) s += a.v[i] * b.v[i]; return s; }
DEV V3 lin(const S6& a) { return v3(a.v[0], a.v[1], a.v[2]); }
DEV V3 ang(const S6& a) { return v3(a.v[3], a.v[4], a.v[5]); }
DEV S6 mk6(V3 l, V3 a) { S6 r; r.v[0] = l.x; r.v[1] = l.y; r.v[2] = l.z; r.v[3] = a.x; r.v[4] = a.y; r.v[5] = a.z; return r; }
// motion x motion and motion x* force
DEV S6 mcross(const S6& a, const S6& b) { return mk6(cross(ang(a), lin(b)) + cross(lin(a), ang(b)), cross(ang(a), ang(b))); }
DEV S6 fcross(const S6& a, const S6& f) { return mk6(cross(ang(a), lin(f)), cross(ang(a), ang(f)) + cross(lin(a), lin(f))); }
DEV S6 mat6_mul(const double* Y, const S6& x) { S6 r; for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += Y[6 * i + j] * x.v[j]; r.v[i] = s; } return r; }
DEV S6 mat6_tmul(const double* Y, const S6& x) { S6 r; for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 6; ++j) s += Y[6 * j + i] * x.v[j]; r.v[i] = s; } return r; }
DEV M3 ldm3(const double* p) { M3 r; for (int i = 0; i < 9; ++i) r.m[i] = p[i]; return r; }
DEV V3 ldv3(const double* p) { return v3(p[0], p[1], p[2]); }
// Ad(M)^-1 on a motion, M = (R, p)
DEV S6 adinv(const M3& R, V3 p, const S6& m) { return mk6(tmul(R, lin(m) - cross(p, ang(m))), tmul(R, ang(m))); }

// ---- SE(3) Jacobians (Barfoot's Q block; right Jacobians as used by Pinocchio's Jlog6 / Jexp6) ------------
DEV void q_coeffs(double t2, double& a1, double& a2, double& a3) {
  if (t2 < kSmall2) {
    a1 = 1.0 / 6 - t2 * (1.0 / 120 - t2 * (1.0 / 5040 - t2 * (1.0 / 362880)));
    a2 = 1.0 / 24 - t2 * (1.0 / 720 - t2 * (1.0 / 40320 - t2 * (1.0 / 3628800)));
    a3 = 1.0 / 120 - t2 * (1.0 / 2520 - t2 * (1.0 / 120960 - t2 * (1.0 / 9979200)));
  } else {
    const double t = sqrt(t2), s = sin(t), c = cos(t);
    a1 = (t - s) / (t2 * t); a2 = (t2 + 2 * c - 2) / (2 * t2 * t2); a3 = (2 * t - 3 * s + t * c) / (2 * t2 * t2 * t);
  }
}
DEV M3 add3(const M3& A, const M3& B) { M3 C; for (int i = 0; i < 9; ++i) C.m[i] = A.m[i] + B.m[i]; return C; }
DEV M3 scl3(double s, const M3& A) { M3 C; for (int i = 0; i < 9; ++i) C.m[i] = s * A.m[i]; return C; }
DEV M3 Qmat(V3 v, V3 w) {
  double a1, a2, a3;
  q_coeffs(dot(w, w), a1, a2, a3);
  const M3 P = skew_m(v), F = skew_m(w);
  const M3 FP = mul(F, P), PF = mul(P, F), FPF = mul(FP, F), FF = mul(F, F);
  M3 Q = scl3(0.5, P);
  Q = add3(Q, scl3(a1, add3(add3(FP, PF), FPF)));
  Q = add3(Q, scl3(a2, add3(add3(mul(FF, P), mul(P, FF)), scl3(-3.0, FPF))));
  Q = add3(Q, scl3(a3, add3(mul(FPF, F), mul(F, FPF))));
  return Q;
}
// out (6x6 row-major) = Jlog6 at M = (R, p)
DEV void Jlog6(const M3& R, V3 p, double* out) {
  V3 v, w;
  log6(R, p, v, w);
  const double t2 = dot(w, w);
  double c;
  if (t2 < kSmall2) c = 1.0 / 12 + t2 * (1.0 / 720 + t2 * (1.0 / 30240 + t2 * (1.0 / 1209600)));
  else { const double t = sqrt(t2); c = (1.0 - t * cos(0.5 * t) / (2.0 * sin(0.5 * t))) / t2; }
  const M3 K = skew_m(w), K2 = mul(K, K);
  M3 Ji;
  for (int i = 0; i < 9; ++i) Ji.m[i] = ((i % 4 == 0) ? 1.0 : 0.0) + 0.5 * K.m[i] + c * K2.m[i];
  const M3 Q = Qmat(v3(-v.x, -v.y, -v.z), v3(-w.x, -w.y, -w.z));
  const M3 B = mul(mul(Ji, Q), Ji);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    out[6 * i + j] = Ji.m[3 * i + j]; out[6 * (i + 3) + j + 3] = Ji.m[3 * i + j];
    out[6 * i + j + 3] = -B.m[3 * i + j]; out[6 * (i + 3) + j] = 0.0;
  }
}
DEV void Jexp6(V3 v, V3 w, double* out) {
  double A, B, C;
  so3_coeffs(dot(w, w), A, B, C);
  const M3 K = skew_m(w), K2 = mul(K, K);
  const M3 Q = Qmat(v3(-v.x, -v.y, -v.z), v3(-w.x, -w.y, -w.z));
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    const double jr = ((i == j) ? 1.0 : 0.0) - B * K.m[3 * i + j] + C * K2.m[3 * i + j];
    out[6 * i + j] = jr; out[6 * (i + 3) + j + 3] = jr; out[6 * i + j + 3] = Q.m[3 * i + j]; out[6 * (i + 3) + j] = 0.0;
  }
}

