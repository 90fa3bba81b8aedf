/*
 * mpc_qp_abi.h — C-ABI of the batched dense QP solver ("next" row N3 of SURVEY.md §8f).
 *
 * What this replaces on the reference side: `proxsuite.proxqp.dense.QP` as QP_utils.py uses it (PrimalDualLDLT backend)
 *     qp = proxsuite.proxqp.dense.QP(n, neq, nin, box, ...); qp.settings.eps_abs / max_iter / max_iter_in      QP_utils.py:500-507, 651-658
 *     qp.init(H, g, A, b, C, l, u[, l_box, u_box])                                                              QP_utils.py:508, 659
 *     qp.update(A=..., b=..., C=..., l=...[, H=..., g=...]); qp.solve(); qp.results.x                          QP_utils.py:556-567, 736-752
 * for the whole-body inverse-dynamics QPs of the 1 kHz loop (IDSolver_ulim :437-575, IKIDSolver_f6 :584-762):
 *     min 1/2 x^T H x + g^T x   s.t.  A x = b,   l <= C x <= u,   l_box <= x <= u_box
 * with n = 2 nv - 6 + 6 nk (62 for the 28-dof model), neq = nv + 6 nk (40), nin = 9 nk (18).
 *
 * One handle solves B independent QPs of the same shape per call (B robots, one QP each): plain pointers and sizes, every
 * array with a leading batch dimension, row-major float64.  Exported by the same two libraries as mpc_abi.h
 * (libmpc_hip.so: one workgroup per QP on the GPU; libmpc_oracle.so: CPU restatement, test infrastructure only).
 *
 * Algorithm (both libraries): proximal augmented Lagrangian in the ProxQP family — outer bound-constrained-Lagrangian loop
 * on (mu_eq, mu_in) with proximal weight rho on x; the inner problem (convex, piecewise quadratic, C^1) by semismooth
 * Newton on the active inequality rows with an exact line search; every Newton system
 *     [ H + rho I + C_I^T C_I / mu_in (+ active box rows / mu_in)    A^T      ] [dx]
 *     [ A                                                          -mu_eq I  ] [y+]
 * by Cholesky of the primal block and of the Schur complement on the equality rows.
 */
#ifndef MPC_QP_ABI_H
#define MPC_QP_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpc_qp_solver mpc_qp_solver; /* opaque */

typedef struct mpc_qp_dims {
  int32_t batch, n, neq, nin;
  int32_t box;     /* 1: l_box <= x <= u_box present (QP(n, neq, nin, True)) */
  int32_t device;  /* HIP device ordinal (oracle: ignored) */
} mpc_qp_dims;

typedef struct mpc_qp_settings {
  double eps_abs;           /* qp.settings.eps_abs (1e-3 in QP_utils.py)                                   */
  double rho;               /* proximal weight on x              (proxsuite default 1e-6)                  */
  double mu_eq, mu_in;      /* initial penalty parameters        (proxsuite defaults 1e-3, 1e-1)           */
  double mu_min_eq, mu_min_in; /* lower bounds of the BCL updates (1e-9, 1e-8)                             */
  double mu_update_factor;  /* 0.1                                                                          */
  double alpha_bcl, beta_bcl; /* 0.1, 0.9                                                                   */
  int32_t max_iter;         /* outer iterations  (qp.settings.max_iter)                                     */
  int32_t max_iter_in;      /* Newton iterations per outer iteration (qp.settings.max_iter_in)              */
  int32_t warm_start;       /* 1: start from the handle's previous solution, 0: from zero                   */
  int32_t reserved;
} mpc_qp_settings;

typedef struct mpc_qp_info {
  double prim_res, dual_res;  /* inf-norms at the returned point */
  double mu_eq, mu_in;
  int32_t iters, iters_in;    /* outer iterations, Newton steps in total */
  int32_t status;             /* 0 solved to eps_abs, 1 iteration limit, 2 factorisation failed */
  int32_t n_active;           /* active inequality + box rows at the solution */
} mpc_qp_info;

int mpc_qp_create(const mpc_qp_dims* dims, mpc_qp_solver** out);
void mpc_qp_destroy(mpc_qp_solver* s);
const char* mpc_qp_last_error(mpc_qp_solver* s);
void mpc_qp_default_settings(mpc_qp_settings* out);
/* Solve B QPs.  H[B][n][n] (symmetric), g[B][n], A[B][neq][n], b[B][neq], C[B][nin][n], l[B][nin], u[B][nin],
 * l_box / u_box [B][n] (ignored unless dims.box).  Outputs x[B][n], y[B][neq], z[B][nin], z_box[B][n] (may be NULL),
 * info[B].  Sign convention of the multipliers: H x + g + A^T y + C^T z + z_box = 0, z_i > 0 on an active upper bound. */
int mpc_qp_solve(mpc_qp_solver* s, const mpc_qp_settings* settings, const double* H, const double* g, const double* A, const double* b,
                 const double* C, const double* l, const double* u, const double* l_box, const double* u_box,
                 double* x, double* y, double* z, double* z_box, mpc_qp_info* info);

/* ---- on-device assembly of the whole-body inverse-dynamics QP (QP_utils.py IDSolver.computeMatrice / solve) ----
 * The reference builds every QP on the host from Pinocchio quantities (QP_utils.py:120-158: computeAllTerms, frame
 * Jacobians, M, nle, frame accelerations) and hands dense matrices to proxqp; here the robot model is uploaded once
 * and one kernel per batch builds A, b, C, l in HBM from (x, a, forces, contact states), after which the QP kernel
 * runs on them without the matrices crossing PCIe.
 * mpc_qp_set_model: the same two tables as mpc_set_model (include/mpc_abi.h, MPC_MODEL_* layout).
 * mpc_qp_solve_id: the handle must have n = 2 nv - 6 + 6 nk, neq = nv + 6 nk, nin = 9 nk, box = 0.
 *   frames[nk]: model frame indices of the contacts; weights[2]: the diagonal weights of H on da and on df
 *   (QP_utils.py:98-103; the torque block is zero); cone[2][9][6]: the wrench-cone rows Cmin that go into C (QP_utils.py:466-490), then the rows cone_l that
 *   form the lower bound l = - cone_l f (QP_utils.py:538-548 writes l out by hand: its rows 2, 3 are f_y -+ mu f_z although rows 2, 3 of Cmin repeat
 *   the f_x rows — pass both as the reference has them for a bit-for-bit drop-in, or twice the same matrix for a consistent cone);
 *   kd: Baumgarte velocity gain (QP_utils.py:105); xrob[B][nq+nv], acc[B][nv], forces[B][6 nk],
 *   contact_states[B][nk] (0/1).  Outputs x[B][n] = (da, df, tau), y, z (may be NULL), info[B]; A_out / b_out /
 *   C_out / l_out (may be NULL) read the assembled matrices back for inspection. */
int mpc_qp_set_model(mpc_qp_solver* s, const int32_t* itab, int32_t n_i, const double* dtab, int32_t n_d);
int mpc_qp_solve_id(mpc_qp_solver* s, const mpc_qp_settings* settings, int32_t nk, const int32_t* frames, const double* weights,
                    const double* cone, double kd, const double* xrob, const double* acc, const double* forces,
                    const int32_t* contact_states, double* x, double* y, double* z, mpc_qp_info* info, double* A_out, double* b_out,
                    double* C_out, double* l_out);

/* The same for the IK + ID QP of the centroidal pipeline (QP_utils.py:584-762 IKIDSolver_f6, centroidal_talos.py:326, 435): unknowns
 * (a, df, tau); posture, foot-acceleration, centroidal-momentum-rate and base / torso orientation tasks in the cost, dynamics and
 * contact-acceleration equalities, wrench cones, torque box.  The handle must have n = 2 nv - 6 + 6 nk, neq = nv + 6 nk, nin = 9 nk,
 * box = 1, nk = 2.  weights[5] (posture, feet, momentum, orientation, force increments); gains: Kp, Kd of the posture task (nv x nv
 * each, row-major), of the foot tasks (6 x 6 each), of the orientation tasks (3 x 3 each); cone[2][9][6] as in mpc_qp_solve_id; l_box / u_box [n] (the torque box, +-inf
 * elsewhere as large numbers); ik[B][2 nv + 42] per robot: q_diff, dq_diff | per contact: pose error (6), its rate (6) | base_diff,
 * dbase_diff, torso_diff, dtorso_diff (3 each) | dH (6): the task errors the script computes from its references.  Outputs as
 * mpc_qp_solve plus the assembled H, g, A, b, C, l (each may be NULL). */
int mpc_qp_solve_ikid(mpc_qp_solver* s, const mpc_qp_settings* settings, int32_t nk, const int32_t* frames, int32_t base_frame, int32_t torso_frame,
                      const double* weights, const double* gains, const double* cone, const double* l_box, const double* u_box,
                      const double* xrob, const double* ik, const double* forces, const int32_t* contact_states,
                      double* x, double* y, double* z, double* z_box, mpc_qp_info* info,
                      double* H_out, double* g_out, double* A_out, double* b_out, double* C_out, double* l_out);

/* ---- device-side glue of the kinodynamic control pipeline (kinodynamic_talos.py:411-462) ----
 * `steps` periods of the 1 kHz low-level loop for every robot of the batch, without the host in between.  Per period, on the device:
 *     d       = difference(x_measured, xs[0])                                             kinodynamic_talos.py:412-418, 424
 *     a0      = [ base part of xdot(knot 0)'s acceleration ; us[0][6 nk:] - K_0[6 nk:] d ]  :420-431
 *     forces  =   us[0][:6 nk] - K_0[:6 nk] d                                              :432-434
 *     (da, df, tau) = the inverse-dynamics QP of mpc_qp_solve_id at (x_measured, a0, forces, contact_states)   :438-446
 *     tau     = clamp(tau, -tau_max, tau_max)                                              :448-456
 *     x_measured <- one simulator step of length dt under tau (mpc_simulate_torque)        :458 (device.execute)
 * `plan`: the MPC handle (include/mpc_abi.h) of the kinodynamic problem, controls u = (contact wrenches [6 nk], joint accelerations
 * [nv - 6]); its solution xs[0], us[0], the Riccati gain K_0 and xdot of knot 0 are read where the last run left them.  `sim`: the
 * simulator handle (horizon 1, whole-body contact dynamics with nu = nv - 6, the contact set of its stage 0 — the handle of
 * mpc_simulate_torque) ; the three handles live on one device and share the batch size.
 * x[B][nq+nv]: the measured states to start from (NULL: the simulator handle's, so that calls continue one another) ; frames, weights, cone,
 * kd, contact_states[B][nk] as in mpc_qp_solve_id ; tau_max[nv - 6].  Outputs (each may be NULL): x_prev[B][nq+nv] the measured states BEFORE
 * the last period (what the script keeps as the next solve's initial condition, :414-415, 482-486), x_out[B][nq+nv] after it,
 * tau[B][nv-6] and forces[B][6 nk] (= forces + df) of the last period, info[B] of its QP. */
typedef struct mpc_solver mpc_solver;
int mpc_qp_low_level_steps(mpc_qp_solver* s, const mpc_qp_settings* settings, mpc_solver* plan, mpc_solver* sim, int32_t nk, const int32_t* frames,
                           const double* weights, const double* cone, double kd, const int32_t* contact_states, const double* tau_max,
                           const double* x, int32_t steps, double dt, double* x_prev, double* x_out, double* tau, double* forces, mpc_qp_info* info);

#ifdef __cplusplus
}
#endif
#endif
