// eval_vector.h — per-knot evaluation of a vector-space stage: the centroidal OCP of
// centroidal_talos.py:202-247 (CentroidalFwdDynamics + IntegratorEuler, control / momentum / acceleration
// costs, CentroidalWrenchCone constraints).  One workgroup (64 lanes = one wavefront) per
// (knot, instance, linesearch candidate); the state is 9-dimensional so the whole stage lives in LDS.
#pragma once
#include "eval_common.h"

// trial == 0: evaluate value + derivatives at the current iterate and fill the LQ knot record.
// trial == 1: value only at (x (+) alpha dx, u + alpha du), candidate alpha = 2^-blockIdx.z; writes the
//             merit partial of the knot to trial_phi.
template <int TRIAL>
__global__ void __launch_bounds__(64) k_eval_vector(SolverArgs a, Layout KL, double* records, int cand0) {
  const Layout& L = a.L;
  const int k = (!TRIAL && a.only_knot >= 0) ? a.only_knot : (int)blockIdx.x, b = blockIdx.y, cand = blockIdx.z + cand0, tid = threadIdx.x, nthr = blockDim.x;  // (only_knot: the launch is that knot alone)
  const InstState& st = a.inst[b];
  if (st.done || (TRIAL && st.skip_step)) return;
  if (TRIAL && cand > 0 && !st.ls_more) return;  // the full step was accepted: no backtracking candidates needed
  if (!TRIAL && a.only_knot >= 0 && k != a.only_knot) return;  // (refinement of the appended knot)
  const int n = L.n, N = L.N, nx = L.nx, mfull = L.m;
  const int slot = stage_slot(a, k);
  const int32_t* desc = a.stage_desc + (size_t)slot * L.max_stage_ints;
  const double* P = (a.inst_params ? a.inst_params + (size_t)b * (L.N + 1) * L.max_stage_doubles : a.stage_params) + (size_t)slot * L.max_stage_doubles;
  const int dyn = desc[0];
  const int m = (dyn == MPC_DYN_NONE) ? 0 : mfull, nzk = n + m, nterms = desc[5], c = desc[6];
  const bool derivs = !TRIAL;
  const double alpha = TRIAL ? ldexp(1.0, -cand) : 0.0;
  double* kn = records + (TRIAL ? (((size_t)b * L.n_alpha + cand) * (N + 1) + k) : ((size_t)b * (N + 1) + k)) * KL.knot_stride;

  __shared__ double x[16], u[16], xn[16], r[24], Wr[24], J[24 * 24], WJ[24 * 24], red[128];
  __shared__ double s_cost;
  const int ldj = 24;
  // ---- load the evaluation point ----
  {
    const double* xs = a.xs + ((size_t)b * (N + 1) + k) * nx;
    const double* dx = a.dxs + ((size_t)b * (N + 1) + k) * n;
    for (int i = tid; i < nx; i += nthr) x[i] = xs[i] + (TRIAL ? alpha * dx[i] : 0.0);
    if (k < N) {
      const double* us = a.us + ((size_t)b * N + k) * mfull;
      const double* du = a.dus + ((size_t)b * N + k) * mfull;
      for (int i = tid; i < mfull; i += nthr) u[i] = us[i] + (TRIAL ? alpha * du[i] : 0.0);
      for (int i = tid; i < nx; i += nthr) xn[i] = xs[nx + i] + (TRIAL ? alpha * dx[n + i] : 0.0);
    }
    if (tid == 0) s_cost = 0.0;
  }
  __syncthreads();
  clear_knot(KL, kn, nzk, derivs, tid, nthr);

  // ---- dynamics: x+ = x + dt * [h/m ; m g + sum f ; sum (p - c) x f + tau] ----
  if (dyn == MPC_DYN_CENTROIDAL_EULER) {
    const double* dp = P + desc[4];
    const double mass = dp[0], dt = dp[4];
    if (derivs) {
      for (int idx = tid; idx < n * nzk; idx += nthr) kn[KL.oAB + (idx / nzk) * KL.nz + idx % nzk] = (idx / nzk == idx % nzk) ? 1.0 : 0.0;
    }
    __syncthreads();
    if (tid == 0) {
      double xd[9];
      for (int i = 0; i < 3; ++i) { xd[i] = x[3 + i] / mass; xd[3 + i] = mass * dp[1 + i]; xd[6 + i] = 0.0; }
      if (derivs) for (int i = 0; i < 3; ++i) kn[KL.oAB + i * KL.nz + 3 + i] += dt / mass;
      for (int kc = 0; kc < desc[1]; ++kc) {
        if (!desc[2 + kc]) continue;
        const double* p = dp + 5 + 3 * kc;
        const double* fk = u + 6 * kc;
        const double rr[3] = {p[0] - x[0], p[1] - x[1], p[2] - x[2]};
        for (int i = 0; i < 3; ++i) xd[3 + i] += fk[i];
        xd[6] += rr[1] * fk[2] - rr[2] * fk[1] + fk[3];
        xd[7] += rr[2] * fk[0] - rr[0] * fk[2] + fk[4];
        xd[8] += rr[0] * fk[1] - rr[1] * fk[0] + fk[5];
        if (derivs) {
          const double F[3][3] = {{0, -fk[2], fk[1]}, {fk[2], 0, -fk[0]}, {-fk[1], fk[0], 0}};
          const double Rx[3][3] = {{0, -rr[2], rr[1]}, {rr[2], 0, -rr[0]}, {-rr[1], rr[0], 0}};
          for (int i = 0; i < 3; ++i) {
            kn[KL.oAB + (3 + i) * KL.nz + n + 6 * kc + i] += dt;
            kn[KL.oAB + (6 + i) * KL.nz + n + 6 * kc + 3 + i] += dt;
            for (int j = 0; j < 3; ++j) {
              kn[KL.oAB + (6 + i) * KL.nz + j] += dt * F[i][j];
              kn[KL.oAB + (6 + i) * KL.nz + n + 6 * kc + j] += dt * Rx[i][j];
            }
          }
        }
      }
      for (int i = 0; i < n; ++i) {
        const double xp = x[i] + dt * xd[i];
        kn[KL.oF + i] = xp - xn[i];
        if (derivs) { kn[KL.oXD + i] = xd[i]; kn[KL.oXN + i] = xp; }
      }
      if (derivs) for (int i = 0; i < 36; ++i) { kn[KL.oE6 + i] = (i % 7 == 0) ? -1.0 : 0.0; kn[KL.oT6k + i] = (i % 7 == 0) ? 1.0 : 0.0; }
    }
    __syncthreads();
  }

  // ---- cost stack and constraints ----
  int row = 0;
  for (int t = 0; t < nterms; ++t) {
    const TermRec tr = load_term(desc, t);
    const double* tp = P + tr.poff;
    const int d = tr.dim;
    for (int idx = tid; idx < d * ldj; idx += nthr) J[idx] = 0.0;
    for (int i = tid; i < d; i += nthr) r[i] = 0.0;
    __syncthreads();
    if (tr.type == MPC_TERM_STATE_ERROR) {
      for (int i = tid; i < d; i += nthr) { r[i] = tp[tr.i0 + i] - x[tr.i0 + i]; J[i * ldj + tr.i0 + i] = -1.0; }  // x_ref (-) x
    } else if (tr.type == MPC_TERM_CONTROL_ERROR) {
      for (int i = tid; i < d; i += nthr) { r[i] = u[tr.i0 + i] - tp[tr.i0 + i]; J[i * ldj + n + tr.i0 + i] = 1.0; }
    } else if (tr.type == MPC_TERM_CENTROIDAL_WRENCH_CONE) {
      for (int i = tid; i < d; i += nthr) {
        double s = 0;
        for (int j = 0; j < 6; ++j) { s += tp[i * 6 + j] * u[6 * tr.i0 + j]; J[i * ldj + n + 6 * tr.i0 + j] = tp[i * 6 + j]; }
        r[i] = s;
      }
    } else if (tr.type == MPC_TERM_CENTROIDAL_LIN_ACC) {
      if (tid < 3) {
        const int i = tid;
        double s = tp[1 + i];
        for (int kc = 0; kc < tr.i0; ++kc) {
          if (tp[4 + 4 * kc] == 0.0) continue;
          s += u[6 * kc + i] / tp[0];
          J[i * ldj + n + 6 * kc + i] = 1.0 / tp[0];
        }
        r[i] = s;
      }
    } else if (tr.type == MPC_TERM_CENTROIDAL_ANG_ACC) {
      if (tid == 0) {
        for (int kc = 0; kc < tr.i0; ++kc) {
          if (tp[4 + 4 * kc] == 0.0) continue;
          const double* p = tp + 4 + 4 * kc + 1;
          const double* fk = u + 6 * kc;
          const double rr[3] = {p[0] - x[0], p[1] - x[1], p[2] - x[2]};
          r[0] += rr[1] * fk[2] - rr[2] * fk[1] + fk[3];
          r[1] += rr[2] * fk[0] - rr[0] * fk[2] + fk[4];
          r[2] += rr[0] * fk[1] - rr[1] * fk[0] + fk[5];
          const double F[3][3] = {{0, -fk[2], fk[1]}, {fk[2], 0, -fk[0]}, {-fk[1], fk[0], 0}};
          const double Rx[3][3] = {{0, -rr[2], rr[1]}, {rr[2], 0, -rr[0]}, {-rr[1], rr[0], 0}};
          for (int i = 0; i < 3; ++i) {
            J[i * ldj + n + 6 * kc + 3 + i] += 1.0;
            for (int j = 0; j < 3; ++j) { J[i * ldj + j] += F[i][j]; J[i * ldj + n + 6 * kc + j] += Rx[i][j]; }
          }
        }
      }
    }
    __syncthreads();
    if (tr.role == MPC_ROLE_COST) {
      accumulate_cost(KL, kn, tr, P + tr.woff, r, J, ldj, nzk, Wr, WJ, derivs, s_cost, tid, nthr);
    } else {
      emit_constraint(KL, kn, tr, P, row, r, J, ldj, nzk, derivs, tid, nthr);
      row += d;
    }
  }
  if (derivs) {
    for (int z = tid; z < nzk; z += nthr) kn[KL.oH + z * KL.nz + z] += a.opt.reg_init;
  }
  __syncthreads();

  // ---- projections, AL penalty, infeasibility ----
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const size_t vo = ((size_t)b * (N + 1) + k) * L.c, lo = ((size_t)b * (N + 1) + k + 1) * n;
  double pen = 0, prim = 0;
  knot_merit(KL, kn, c, (k < N) ? kn + KL.oF : nullptr, a.vs + vo, TRIAL ? a.dvs + vo : nullptr, a.vs_e + vo,
             a.lams + lo, TRIAL ? a.dlams + lo : nullptr, a.lams_e + lo, alpha, mu, mud, derivs, red, pen, prim, tid, nthr);
  if (tid == 0) {
    if (TRIAL) {
      a.trial_phi[((size_t)b * L.n_alpha + cand) * (N + 1) + k] = s_cost + pen;
    } else {
      double* ms = kn + KL.oMISC;
      ms[MISC_COST] = s_cost; ms[MISC_PEN] = pen; ms[MISC_PRIM] = prim; ms[MISC_NC] = (double)c; ms[MISC_M] = (double)m;
    }
  }
}
