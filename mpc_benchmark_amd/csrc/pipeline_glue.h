// pipeline_glue.h — the two small kernels that string the kinodynamic control pipeline together on the device (mpc_qp_low_level_steps,
// include/mpc_qp_abi.h; kinodynamic_talos.py:411-462): the feedback terms of the plan's knot 0 into the inputs of the inverse-dynamics QP, and the QP's
// torque — clamped — into the simulator step.  One workgroup of one wavefront per robot; the vectors are a few hundred bytes.
#pragma once
#include "solver_args.h"

struct PipeArgs {
  // the plan (kinodynamic MPC handle): solution of knot 0, its Riccati gain, xdot of its stage data
  const double *xs, *us, *gains, *knots;
  int N, nx, nq, nv, n, m, gain_stride, oK, knot_stride, oXD, slot0;
  const double* x;    // [B][nx] measured states (the simulator handle's)
  // inverse-dynamics QP: inputs of its assembly kernel, its solution (da, df, tau)
  double *xrob, *acc, *f;
  const double* sol;
  int nk, qn;
  const double* tau_max;  // [nv - 6]
  double *sim_u;          // [B][nv - 6] torques of the simulator step
  double *f_new;          // [B][6 nk] forces + df
};

#define PIPE_MAX_N 160  // tangent dimension 2 nv of the plan

// d = difference(x_measured, xs[0]) ; forces = us[0][:6 nk] - K_0[:6 nk] d ; a0 = [xdot(knot 0) base acceleration ; us[0][6 nk:] - K_0[6 nk:] d]
__global__ void __launch_bounds__(64) k_pipe_feedback(PipeArgs p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nx = p.nx, nq = p.nq, nv = p.nv, n = p.n, m = p.m, nf = 6 * p.nk;
  const double* x = p.x + (size_t)b * nx;
  const double* x0 = p.xs + (size_t)b * (p.N + 1) * nx;
  const double* us0 = p.us + (size_t)b * p.N * m;
  const double* K0 = p.gains + (size_t)b * (p.N + 1) * p.gain_stride + p.oK;
  const double* xd = p.knots + ((size_t)b * (p.N + 1) + p.slot0) * p.knot_stride + p.oXD;
  __shared__ double dd[PIPE_MAX_N];
  if (lane == 0) {
    const M3 Rx = quat_to_rot(x + 3), R0 = quat_to_rot(x0 + 3);
    V3 ev, ew;
    log6(tmul(Rx, R0), tmul(Rx, v3(x0[0] - x[0], x0[1] - x[1], x0[2] - x[2])), ev, ew);
    dd[0] = ev.x; dd[1] = ev.y; dd[2] = ev.z; dd[3] = ew.x; dd[4] = ew.y; dd[5] = ew.z;
  }
  for (int i = 6 + lane; i < nv; i += 64) dd[i] = x0[i + 1] - x[i + 1];
  for (int i = lane; i < nv; i += 64) dd[nv + i] = x0[nq + i] - x[nq + i];
  __syncthreads();
  for (int i = lane; i < m; i += 64) {
    double su = 0.0;
    for (int j = 0; j < n; ++j) su += K0[(size_t)i * n + j] * dd[j];
    su = us0[i] - su;
    if (i < nf) p.f[(size_t)b * nf + i] = su;
    else p.acc[(size_t)b * nv + 6 + (i - nf)] = su;
  }
  if (lane < 6) p.acc[(size_t)b * nv + lane] = xd[nv + lane];
  for (int i = lane; i < nx; i += 64) p.xrob[(size_t)b * nx + i] = x[i];
}

// tau = clamp(QP torque, +-tau_max) into the simulator's input ; forces + df
__global__ void __launch_bounds__(64) k_pipe_torque(PipeArgs p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int nv = p.nv, nf = 6 * p.nk, nu = nv - 6;
  const double* sol = p.sol + (size_t)b * p.qn;
  for (int i = lane; i < nu; i += 64) {
    const double t = sol[nv + nf + i], lim = p.tau_max[i];
    p.sim_u[(size_t)b * nu + i] = fmin(fmax(t, -lim), lim);
  }
  for (int i = lane; i < nf; i += 64) p.f_new[(size_t)b * nf + i] = p.f[(size_t)b * nf + i] + sol[nv + i];
}
