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

#include "se3_math.h"

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
static inline constexpr unsigned magic_div(int d) { return (unsigned)((0x100000000ull + (unsigned long long)d - 1) / (unsigned long long)d); }
