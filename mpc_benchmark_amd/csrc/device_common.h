// device_common.h — workgroup-level dense fp64 building blocks shared by the HIP kernels.
// A "block routine" is called by every thread of the workgroup with its (tid, nthr); routines that end with
// data other threads will read finish with __syncthreads().
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mpc_abi.h"
#include "layout.h"

#define DEV __device__ __forceinline__

// projection on the normal cone of one constraint row (same tie-breaking as the oracle: strict inequalities)
DEV double proj_normal(int role, double z, double lo, double hi, bool& active) {
  if (role == MPC_ROLE_EQUALITY) { active = true; return z; }
  if (role == MPC_ROLE_NEG_ORTHANT) { active = z > 0.0; return active ? z : 0.0; }
  if (role == MPC_ROLE_BOX) {
    if (z < lo) { active = true; return z - lo; }
    if (z > hi) { active = true; return z - hi; }
  }
  active = false;
  return 0.0;
}

// In-place lower Cholesky of the n x n matrix A (leading dimension ld), right-looking, whole workgroup.
// Returns false (to every thread) when a pivot is not positive.
DEV bool chol_block(double* A, int n, int ld, int tid, int nthr, int* flag) {
  if (tid == 0) *flag = 1;
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    const double d = A[j * ld + j];
    if (!(d > 0.0)) { if (tid == 0) *flag = 0; }
    __syncthreads();
    if (*flag == 0) return false;
    const double inv = 1.0 / sqrt(d);
    for (int i = j + 1 + tid; i < n; i += nthr) A[i * ld + j] *= inv;
    __syncthreads();
    if (tid == 0) A[j * ld + j] = sqrt(d);
    // trailing update of the lower triangle
    const int r = n - j - 1;
    for (int idx = tid; idx < r * r; idx += nthr) {
      const int a = j + 1 + idx / r, b = j + 1 + idx % r;
      if (b <= a) A[a * ld + b] -= A[a * ld + j] * A[b * ld + j];
    }
    __syncthreads();
  }
  return true;
}

// Solve L X = B then L^T X = B in place for r right-hand sides stored row-major in B (n x r, ldb).
// One thread per column.
DEV void potrs_block(const double* L, int n, int ldl, double* B, int r, int ldb, int tid, int nthr) {
  for (int j = tid; j < r; j += nthr) {
    for (int i = 0; i < n; ++i) {
      double s = B[i * ldb + j];
      for (int k = 0; k < i; ++k) s -= L[i * ldl + k] * B[k * ldb + j];
      B[i * ldb + j] = s / L[i * ldl + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = B[i * ldb + j];
      for (int k = i + 1; k < n; ++k) s -= L[k * ldl + i] * B[k * ldb + j];
      B[i * ldb + j] = s / L[i * ldl + i];
    }
  }
  __syncthreads();
}
// forward substitution only (L X = B)
DEV void trsm_lower_block(const double* L, int n, int ldl, double* B, int r, int ldb, int tid, int nthr) {
  for (int j = tid; j < r; j += nthr)
    for (int i = 0; i < n; ++i) {
      double s = B[i * ldb + j];
      for (int k = 0; k < i; ++k) s -= L[i * ldl + k] * B[k * ldb + j];
      B[i * ldb + j] = s / L[i * ldl + i];
    }
  __syncthreads();
}
// backward substitution only (L^T X = B)
DEV void trsm_lower_t_block(const double* L, int n, int ldl, double* B, int r, int ldb, int tid, int nthr) {
  for (int j = tid; j < r; j += nthr)
    for (int i = n - 1; i >= 0; --i) {
      double s = B[i * ldb + j];
      for (int k = i + 1; k < n; ++k) s -= L[k * ldl + i] * B[k * ldb + j];
      B[i * ldb + j] = s / L[i * ldl + i];
    }
  __syncthreads();
}

// 6x6 inverse by Gauss-Jordan with partial pivoting (single thread)
DEV void inv6_serial(const double* A, double* Ainv) {
  double M[6][12];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = (i == j) ? 1.0 : 0.0; }
  for (int k = 0; k < 6; ++k) {
    int piv = k;
    double best = fabs(M[k][k]);
    for (int i = k + 1; i < 6; ++i) if (fabs(M[i][k]) > best) { best = fabs(M[i][k]); piv = i; }
    if (piv != k) for (int j = 0; j < 12; ++j) { const double t = M[k][j]; M[k][j] = M[piv][j]; M[piv][j] = t; }
    const double inv = 1.0 / M[k][k];
    for (int j = 0; j < 12; ++j) M[k][j] *= inv;
    for (int i = 0; i < 6; ++i) {
      if (i == k) continue;
      const double l = M[i][k];
      for (int j = 0; j < 12; ++j) M[i][j] -= l * M[k][j];
    }
  }
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) Ainv[i * 6 + j] = M[i][6 + j];
}

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

// x (+) alpha*dx for either state space.  Multibody: q = [p, quat(xyzw), joints], free-flyer at the root.
// the same by a group of threads (t0 of nt): the elementwise part in parallel, the free-flyer pose by thread 0
DEV void state_integrate_group(int space, int nx, int n, const double* x, const double* dx, double alpha, double* out, int t0, int nt) {
  if (space == MPC_SPACE_VECTOR) {
    for (int i = t0; i < nx; i += nt) out[i] = x[i] + alpha * dx[i];
    return;
  }
  const int nv = n / 2, nq = nx - nv;
  for (int i = 6 + t0; i < nv; i += nt) out[i + 1] = x[i + 1] + alpha * dx[i];
  for (int i = t0; i < nv; i += nt) out[nq + i] = x[nq + i] + alpha * dx[nv + i];
  if (t0 == 0) {
    const M3 R = quat_to_rot(x + 3);
    M3 dR;
    V3 dp;
    exp6(v3(alpha * dx[0], alpha * dx[1], alpha * dx[2]), v3(alpha * dx[3], alpha * dx[4], alpha * dx[5]), dR, dp);
    const V3 p = v3(x[0], x[1], x[2]) + mul(R, dp);
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
    rot_to_quat(mul(R, dR), out + 3);
  }
}

DEV void state_integrate(int space, int nx, int n, const double* x, const double* dx, double alpha, double* out) {
  if (space == MPC_SPACE_VECTOR) {
    for (int i = 0; i < nx; ++i) out[i] = x[i] + alpha * dx[i];
    return;
  }
  const int nv = n / 2, nq = nx - nv;
  const M3 R = quat_to_rot(x + 3);
  M3 dR;
  V3 dp;
  exp6(v3(alpha * dx[0], alpha * dx[1], alpha * dx[2]), v3(alpha * dx[3], alpha * dx[4], alpha * dx[5]), dR, dp);
  const V3 p = v3(x[0], x[1], x[2]) + mul(R, dp);
  out[0] = p.x; out[1] = p.y; out[2] = p.z;
  rot_to_quat(mul(R, dR), out + 3);
  for (int i = 6; i < nv; ++i) out[i + 1] = x[i + 1] + alpha * dx[i];
  for (int i = 0; i < nv; ++i) out[nq + i] = x[nq + i] + alpha * dx[nv + i];
}

// x / d for 0 <= x < 2^32 / d, d >= 2, with mg = magic_div(d) = ceil(2^32 / d) computed on the host: one v_mul_hi_u32 instead of the
// ~30 instructions of a 32-bit division by a run-time divisor (the LDS fills of the leg kernels divide an index per element)
DEV int qdiv(int x, unsigned mg) { return (int)__umulhi((unsigned)x, mg); }
static inline unsigned magic_div(int d) { return (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }
