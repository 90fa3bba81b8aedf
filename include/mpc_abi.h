/*
 * mpc_abi.h — C-ABI of the MI355X-native receding-horizon ProxDDP solver (drop-in boundary).
 *
 * What this replaces on the reference side: the Boost.Python/eigenpy module `aligator`
 * (README.md:10) as it is used by the three driver scripts, i.e. the calls
 *     solver = aligator.SolverProxDDP(TOL, mu_init)          fulldynamic_talos.py:379, kinodynamic_talos.py:285, centroidal_talos.py:270
 *     solver.setup(problem); solver.run(problem, xs, us)     fulldynamic_talos.py:539-540, kinodynamic_talos.py:490, centroidal_talos.py:461-462
 *     problem.replaceStageCircular / solver.cycleProblem     fulldynamic_talos.py:496-497, kinodynamic_talos.py:488, centroidal_talos.py:459-460
 *     residual.setReference(...), contact_poses[i] = ...     fulldynamic_talos.py:461-463, centroidal_talos.py:374-384
 *     results.xs / results.us / controlFeedbacks()[0]        fulldynamic_talos.py:548-550
 *     workspace...stage_data[0]...xdot / contact_force       fulldynamic_talos.py:465-485, kinodynamic_talos.py:432, centroidal_talos.py:409
 * The Python package `mpc_benchmark_amd.aligator` keeps that Python surface and lowers a
 * TrajOptProblem to the flat tables below; nothing but plain pointers and sizes crosses this boundary.
 *
 * Two shared libraries export exactly these symbols:
 *   mpc_benchmark_amd/csrc/libmpc_hip.so   — the product: hand-written HIP kernels for gfx950
 *   oracle/libmpc_oracle.so                — TEST INFRASTRUCTURE ONLY: CPU restatement used as the checker
 *
 * Conventions: all functions return 0 on success, <0 on error (mpc_last_error gives text); no
 * exceptions cross the boundary; host buffers are caller-owned, C-contiguous float64/int32, copied
 * during the call and never retained.  One handle = one device stream = one host thread at a time.
 * Every array carries a leading ensemble ("batch") dimension B: B independent MPC instances that
 * share the stage tables and differ in x0 / warm start / multipliers.
 */
#ifndef MPC_ABI_H
#define MPC_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_ABI_VERSION 3

/* ---- state manifolds -------------------------------------------------------------------- */
#define MPC_SPACE_VECTOR 0     /* aligator.manifolds.VectorSpace(n)            centroidal_talos.py:46   */
#define MPC_SPACE_MULTIBODY 1  /* aligator.manifolds.MultibodyPhaseSpace(model) fulldynamic_talos.py:62 */

/* ---- joint kinds in the model table ------------------------------------------------------- */
#define MPC_JOINT_FREEFLYER 0
#define MPC_JOINT_RX 1
#define MPC_JOINT_RY 2
#define MPC_JOINT_RZ 3

/* ---- dynamics kinds (stage descriptor word 0) ---------------------------------------------- */
#define MPC_DYN_NONE 0              /* terminal node */
#define MPC_DYN_CENTROIDAL_EULER 1  /* CentroidalFwdDynamics + IntegratorEuler            centroidal_talos.py:203-204 */
#define MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER 2 /* MultibodyConstraintFwdDynamics + IntegratorSemiImplEuler  fulldynamic_talos.py:103-110 */
#define MPC_DYN_KINODYNAMICS_SEMIEULER 3         /* KinodynamicsFwdDynamics + IntegratorSemiImplEuler         kinodynamic_talos.py:108-111 */

/* ---- residual ("term") types --------------------------------------------------------------- */
#define MPC_TERM_STATE_ERROR 1          /* r = (x_ref (-) x)[i0 : i0+dim]      params: x_ref[nx]   (= space.difference(x, x_ref)) */
#define MPC_TERM_CONTROL_ERROR 2        /* r = (u - u_ref)[i0 : i0+dim]        params: u_ref[nu]                         */
#define MPC_TERM_FRAME_PLACEMENT 3      /* r = log6(Mref^-1 oMf)               iarg0=frame, params: R[9] p[3]            */
#define MPC_TERM_FRAME_TRANSLATION 4    /* r = (oMf.t - p_ref)[i1:i1+dim]      iarg0=frame, params: p[3]                 */
#define MPC_TERM_FRAME_VELOCITY 5       /* r = v_frame(LOCAL) - v_ref          iarg0=frame, params: v_ref[6]             */
#define MPC_TERM_COM_TRANSLATION 6      /* r = (com(q) - p_ref)[i1:i1+dim]     params: p[3]                              */
#define MPC_TERM_CENTROIDAL_MOMENTUM 7  /* r = hg(q,v) - h_ref                 params: h_ref[6]                          */
#define MPC_TERM_CONTACT_FORCE 8        /* r = lambda_c(x,u) - f_ref           iarg0=slot in stage contact list, params: f_ref[6] */
#define MPC_TERM_MB_WRENCH_CONE 9       /* r = A lambda_c(x,u)                 iarg0=slot, params: A[dim*6]              */
#define MPC_TERM_CENTROIDAL_WRENCH_CONE 10 /* r = A u[6k:6k+6]                 iarg0=k,    params: A[dim*6]              */
#define MPC_TERM_CENTROIDAL_LIN_ACC 11  /* r = sum_active f_i / m + g          params: m, g[3], then per contact: state, p[3] */
#define MPC_TERM_CENTROIDAL_ANG_ACC 12  /* r = sum_active (p_i - c) x f_i + tau_i   same params                          */
#define MPC_TERM_CENTROIDAL_MOMENTUM_DER 13 /* r = d/dt hg from contact wrenches (kinodynamic) params: g[3], states, frames */

/* ---- term roles ---------------------------------------------------------------------------- */
#define MPC_ROLE_COST 0            /* 1/2 r^T W r ; weight at woff: dense dim*dim row-major, or diag[dim] if flags&1 */
#define MPC_ROLE_EQUALITY 1        /* r = 0        (constraints.EqualityConstraintSet) */
#define MPC_ROLE_NEG_ORTHANT 2     /* r <= 0       (constraints.NegativeOrthant)       */
#define MPC_ROLE_BOX 3             /* lo <= r <= hi (constraints.BoxConstraint) ; lo[dim], hi[dim] at woff */

#define MPC_TERM_FLAG_DIAG_WEIGHT 1

/* Stage descriptor, int32 words:
 *   [0] dynamics kind   [1] number of contacts listed   [2],[3] contact entries
 *       (multibody: indices into the model's contact table of the ACTIVE contacts;
 *        centroidal / kinodynamic: contact state flag of foot 0 / foot 1)
 *   [4] offset of the dynamics parameters in the stage's double table
 *       centroidal : mass, g[3], dt, p0[3], p1[3]     multibody: dt     kinodynamic: dt, g[3], frame0, frame1
 *   [5] number of terms T   [6] total constraint rows   [7] reserved
 *   then T records of MPC_TERM_WORDS words: type, role, dim, iarg0, iarg1, poff, woff, flags
 */
#define MPC_STAGE_HEADER_WORDS 8
#define MPC_TERM_WORDS 8

/* Model table.
 *  int32: [0] njoints (moving joints, joint 0 is the first moving joint) [1] nq [2] nv [3] nframes [4] ncontacts
 *         then per joint: parent (-1 = world), kind, idx_q, idx_v ; per frame: parent joint ; per contact: joint
 *  f64  : gravity[3], prox_mu, then per joint: R[9] p[3] mass lever[3] I_com[9]  (25 doubles);
 *         per frame: R[9] p[3] (12) ; per contact: R1[9] p1[3] R2[9] p2[3] Kp[6] Kd[6] (36)
 */
#define MPC_MODEL_HEADER_WORDS 5
#define MPC_MODEL_JOINT_WORDS 4
#define MPC_MODEL_HEADER_DOUBLES 4
#define MPC_MODEL_JOINT_DOUBLES 25
#define MPC_MODEL_FRAME_DOUBLES 12
#define MPC_MODEL_CONTACT_DOUBLES 36

typedef struct mpc_dims {
  int32_t horizon;   /* N: number of stages (knots 0..N, N is terminal)      */
  int32_t batch;     /* B: ensemble size                                     */
  int32_t space;     /* MPC_SPACE_*                                          */
  int32_t nx, ndx, nu;
  int32_t nc_max;    /* upper bound on constraint rows of any knot           */
  int32_t max_stage_ints, max_stage_doubles; /* capacity of one stage table  */
  int32_t device;    /* HIP device ordinal (ignored by the oracle)           */
} mpc_dims;

/* SolverProxDDP knobs (fulldynamic_talos.py:374-386); values not set by the scripts keep the
 * defaults documented in DESIGN.md. */
typedef struct mpc_options {
  double tol;              /* SolverProxDDP(tol, .) target tolerance                  */
  double mu_init;          /* SolverProxDDP(., mu_init)                                */
  double dyn_al_scale;     /* mu_dyn = mu * dyn_al_scale                               */
  double reg_init;         /* primal regularisation added to diag(Q), diag(R)          */
  double ls_armijo_c1;
  double ls_alpha_min;
  double bcl_prim_alpha, bcl_prim_beta, bcl_dual_alpha, bcl_dual_beta;
  double bcl_mu_update_factor, bcl_mu_lower_bound;
  double inner_tol0, prim_tol0;
  double corrector_prim_tol; /* > 0: an instance whose run has used up max_iters takes ONE more iteration (a "corrector") when the iterate its last
                            * iteration started from was primal-infeasible by more than this (mpc_stats.prim_infeas: largest constraint violation /
                            * dynamics gap; N, N m, rad), or when that iteration's step had to be shortened by the linesearch (alpha < 1).  For the MPC loops (max_iters = 1, fulldynamic_talos.py:407): the warm start of a tick is
                            * the previous solution shifted by one knot with the last control DUPLICATED (:532-534); on the few ticks where the
                            * appended stage has another contact pattern that duplicate violates the new stage by 100 - 300 N / N m, one Newton
                            * step of the 1 / mu = 1e8 penalty problem lands on rows its active set did not know, and ensembles of perturbed
                            * instances are lost within two walking cycles.  A second iteration on exactly those ticks
                            * (~2 - 5 % of the instance-ticks at 20.0) keeps them.  0 (default of the C-ABI): off.  Not something Aligator is
                            * known to do: this build's globalisation of an iteration budget of one. */
  int32_t max_iters;       /* solver.max_iters (100 cold, 1 in the MPC loop)           */
  int32_t max_al_iters;
  int32_t force_initial_condition;
  int32_t rollout_linear;  /* ROLLOUT_LINEAR = 1 (only mode implemented)               */
  int32_t ls_max_steps;    /* number of backtracking candidates alpha = 2^-i           */
  int32_t num_threads;     /* oracle: OpenMP threads ; HIP: ignored                    */
  int32_t riccati_legs;    /* linear_solver_choice = LQ_SOLVER_PARALLEL + setNumThreads(n) (fulldynamic_talos.py:383,385): the horizon is cut
                            * into this many legs (clamped to 32 and to the horizon; leg j starts at knot floor(j N / legs)) whose Riccati
                            * sweeps run side by side — parallel-in-time, same KKT system, results equal to the serial sweep (1) up to
                            * round-off.  The cuts are resolved by a tree of pairwise compositions (ceil(log2 legs) rounds of independent
                            * solves; MPC_LEGS_CHAIN=1: one after the other, a chain of legs - 1 solves — the first form, kept for tests).
                            * controlFeedbacks()[0] is the exact gain; the gains of later knots are those of their leg.  HIP:
                            * problems with more than 80 tangent dimensions or 48 controls keep the serial sweep */
  int32_t forward_mode;    /* HIP forward sweep: 0 = automatic, 1 = one workgroup per instance walks the knots (least CU time: ensembles
                            * sharded over several handles of one GPU), 2 = knot-parallel closed-loop transitions first (shortest latency:
                            * a single small ensemble) ; oracle: ignored                */
  int32_t refine_appended_knot; /* 0 (default): mpc_run_shifted duplicates the last control into the knot that mpc_cycle appended, as the scripts'
                            * own warm-start shift does (us = us[1:] + [us[-1]], fulldynamic_talos.py:533).  R > 0: when the appended stage has a
                            * different contact pattern than the stage before it (the duplicated torques then violate the new stage's wrench cone
                            * by ~150 N: a foot that comes back to the ground pulls), the warm start of that ONE knot is made consistent with its
                            * own stage first: R Newton steps on u_{N-1} alone, x_{N-1} fixed —
                            *     du = -(H_uu + D_a^T D_a / mu)^-1 (g_u + D_a^T Pi_N(z)_a / mu)    (the knot's own penalty problem, active rows a)
                            * then x_N = phi(x_{N-1}, u_{N-1}).  Costs R + 1 evaluations of one knot per instance on the ~1.4 % of the ticks
                            * where the pattern changes; with it ensembles of randomised instances walk the whole schedule on ONE ProxDDP
                            * iteration per tick.  R < 0: the same |R| steps after EVERY mpc_cycle, whatever the
                            * appended stage is.  For OCPs whose controls contain the contact forces (the kinodynamic one): the references of
                            * the appended stage differ from those of the stage before it as a rule (force references ramp before a
                            * take-off), the duplicated control leaves the appended knot's own active set, and the full step of the tick
                            * then violates rows the Newton system did not know — the kinodynamic walk spends five consecutive ticks
                            * backtracking on that (alpha = 1/2 ... 1/128), and none with R = -1.  EXPERIMENTAL: later in the schedule it makes
                            * most ticks backtrack, and on flat ground it loses instances the plain warm start keeps.
                            * NOT for the whole-body OCP: there the
                            * refinement problem of a knot whose duplicated control is already feasible is ill-conditioned (dependent cone
                            * rows; the nominal instance is lost within 30 ticks) — use R > 0.  Not
                            * something Aligator does: a choice of initial guess, off unless asked for. */
  int32_t corrector_window; /* 0: corrector_prim_tol applies to every run.  K > 0: only to the K runs that follow an mpc_cycle whose appended stage has
                            * another contact pattern than the stage before it (the ticks on which the duplicated control is inconsistent) — lets
                            * mpc_run_shifted_async, which enqueues the passes of a tick ahead of their results, enqueue the corrector pass on those
                            * ticks only.  Same rule in both libraries. */
} mpc_options;

typedef struct mpc_stats {
  int32_t num_iters;
  int32_t converged;
  int32_t al_iters;
  int32_t ls_steps;        /* index of the accepted backtracking candidate in the last iteration */
  double traj_cost;
  double merit;
  double prim_infeas;
  double dual_infeas;
  double mu;
  double alpha;
} mpc_stats;

typedef struct mpc_solver mpc_solver;

int mpc_abi_version(void);
const char* mpc_backend_name(void); /* "hip-gfx950" or "oracle-cpu" */

/* aligator.SolverProxDDP(TOL, mu_init, ...) (fulldynamic_talos.py:379) for an ensemble of dims->batch independent copies
 * of one problem structure; allocates every device buffer once. */
int mpc_create(const mpc_dims* dims, mpc_solver** out);
void mpc_destroy(mpc_solver* s);
const char* mpc_last_error(mpc_solver* s);

/* solver.rollout_type / .max_iters / .force_initial_condition ... (fulldynamic_talos.py:380-386) */
int mpc_set_options(mpc_solver* s, const mpc_options* opt);
/* the pin.Model the residuals and dynamics were built on (talos_utils.py:31-41 loadTalos), its frames and the
 * RigidConstraintModels of fulldynamic_talos.py:84-98, lowered to the model table documented above */
int mpc_set_model(mpc_solver* s, const int32_t* itab, int32_t n_i, const double* dtab, int32_t n_d);

/* TrajOptProblem(x0, stages, term_cost) (fulldynamic_talos.py:153-232 createStage, :372): k in [0, N]; k == N is the
 * terminal node (cost + terminal constraints, :234-245, :499-507). */
int mpc_set_stage(mpc_solver* s, int32_t k, const int32_t* desc, int32_t n_desc, const double* params, int32_t n_params);
/* setReference / contact_poses[i] = ... : overwrite n doubles of stage k's parameter table. */
int mpc_update_stage_params(mpc_solver* s, int32_t k, int32_t offset, const double* vals, int32_t n);
/* The per-tick form of the above (fulldynamic_talos.py:461-463 calls setReference on two residuals of every stage of
 * the horizon): `count` updates in one call — update i overwrites lens[i] doubles at offsets[i] of stage ks[i]; `vals`
 * holds the new values back to back. */
int mpc_update_stage_params_batch(mpc_solver* s, int32_t count, const int32_t* ks, const int32_t* offsets, const int32_t* lens,
                                  const double* vals);
/* replaceStageCircular + cycleAppend/cycleProblem (fulldynamic_talos.py:496-497, kinodynamic_talos.py:395, :488): drop stage 0,
 * shift, install the new stage at N-1. */
int mpc_cycle(mpc_solver* s, const int32_t* desc, int32_t n_desc, const double* params, int32_t n_params);

/* problem.x0_init = x (fulldynamic_talos.py:536): x0[B][nx].  x0 == NULL selects "perfect-model feedback": every later
 * mpc_run_shifted takes the state the previous solution predicted for the next tick (xs[1]) as the new
 * initial condition, so a closed receding-horizon loop runs without any host<->device traffic. */
int mpc_set_x0(mpc_solver* s, const double* x0);
/* "Next" row N2 — closed-loop stand-in for the 1 kHz low-level loop + simulator of fulldynamic_talos.py:512-530
 * (bullet_robot.py:138-145, 172-196): starting from xs[0], knot 0's contact dynamics are integrated `substeps` times with
 * step `dt` (semi-implicit Euler, the scheme of the stage dynamics) under the state-feedback law
 *     u = us[0] - controlFeedbacks()[0] * difference(x, xs[0]) ;
 * the final state becomes the measured state x0 of every instance (as if set by mpc_set_x0), so the next mpc_run_shifted
 * starts from it.  Whole-body contact dynamics only.  mpc_get_x0 reads the measured states back: x0[B][nx]. */
int mpc_simulate(mpc_solver* s, int32_t substeps, double dt);
/* The same with a disturbance: f_ext[B][3], a world-frame force applied at the origin of the base link during the whole call (the
 * 300 N push of fulldynamic_talos.py:433-435, 524-526: device.apply_force(f_disturbance, [0, 0, 0]) on ticks 160 - 170); NULL = none. */
int mpc_simulate_push(mpc_solver* s, int32_t substeps, double dt, const double* f_ext);
/* Torque-driven form — the stand-in for BulletRobot.execute(torques) + p.stepSimulation() and measureState()
 * (bullet_robot.py:138-145, 172-196; the kinodynamic and centroidal loops apply the torque of their whole-body QP this way,
 * kinodynamic_talos.py:458, centroidal_talos.py:447): x[B][nx] = the states to start from (NULL: the handle's measured states x0, so that
 * successive calls continue one another), tau[B][nu] joint torques held during the call.  Knot 0's contact dynamics — its contact set, the
 * Baumgarte-corrected rigid contacts of the model table — are integrated `substeps` times with step dt (semi-implicit Euler); the final state
 * becomes the measured state x0 of every instance (mpc_get_x0), wrenches[B][2][6] (may be NULL) receives the contact wrenches of the
 * last sub-step (LOCAL frame, model contact order, inactive contacts zero).  Whole-body contact dynamics only. */
int mpc_simulate_torque(mpc_solver* s, const double* x, const double* tau, int32_t substeps, double dt, double* wrenches);
int mpc_get_x0(mpc_solver* s, double* x0);
/* solver.setup(problem) (fulldynamic_talos.py:539): reset multipliers, penalty and tolerances (no re-allocation). */
int mpc_setup(mpc_solver* s);
/* solver.run(problem, xs, us) (fulldynamic_talos.py:540): xs[B][N+1][nx], us[B][N][nu]; stats[B] (may be NULL). */
int mpc_run(mpc_solver* s, const double* xs_init, const double* us_init, mpc_stats* stats);
/* Re-run from the solver's own shifted solution: xs <- [xs[1:], xs[-1]], us likewise, xs[0] <- x0
 * (the warm-start shift of fulldynamic_talos.py:532-534 done on the device). */
int mpc_run_shifted(mpc_solver* s, mpc_stats* stats);
/* Asynchronous form for pipelining several handles (ensemble shards) on one device: enqueues the shift, one solver pass
 * (max_iters = 1) and an asynchronous status read-back on the handle's stream and returns without waiting.  Up to TWO
 * ticks may be in flight per handle (enqueue tick t + 1, then wait for tick t: the stream never runs dry).  mpc_wait
 * completes the OLDEST tick in flight and fills stats[B] (may be NULL); when no younger tick is queued behind it, an
 * instance whose pass was a BCL update without a step gets its further passes then (as mpc_run_shifted does), otherwise
 * it carries on in the next tick.  The oracle runs the tick inside mpc_run_shifted_async. */
int mpc_run_shifted_async(mpc_solver* s);
/* Tick reuse for MPC ticks with max_iters = 1 on whole-body problems (HIP; the oracle accepts and ignores it): the full step
 * of a tick is evaluated WITH derivatives into the knot records; when it is accepted, mpc_run_shifted of the next tick finds
 * the records of its knots 0 .. N-2 in place (one knot on) and only refreshes the multiplier-dependent part.  Results are
 * bit-identical to the plain path.  mpc_update_stage_params(_batch) compares the incoming values with the host mirror: unchanged
 * ranges cost nothing, changed ones invalidate the record of THEIR knot only (it is evaluated afresh by the next tick, the others
 * stay reused); set_stage, set_options and mpc_run invalidate all kept records.
 * A no-op on vector-space problems (centroidal): accepted, nothing changes. */
int mpc_set_tick_reuse(mpc_solver* s, int32_t on);
int mpc_wait(mpc_solver* s, mpc_stats* stats);
/* mpc_wait that also hands over x_next[B][nx] = xs[1] of every instance after the completed tick: the state the next tick takes as its
 * measurement under perfect-model feedback, i.e. what the reference generators of the loop (foot poses of the measured state,
 * fulldynamic_talos.py:441-455) need to plan that tick — snapshotted with the status, so the host never waits for a younger tick.
 * With nothing in flight both return the status (and states) of the last completed tick. */
int mpc_wait_state(mpc_solver* s, mpc_stats* stats, double* x_next);
/* Non-blocking look at the asynchronous ticks: *in_flight = ticks enqueued and not yet collected by mpc_wait, *completed = how
 * many of those have already finished on the device (a host-side pacer uses it to tell whether the device keeps up). */
int mpc_poll(mpc_solver* s, int32_t* in_flight, int32_t* completed);

/* results.xs / results.us / controlFeedbacks() (fulldynamic_talos.py:403-405, :522, :548-550) / feed-forwards / multipliers. Any pointer may be NULL.
 * xs[B][N+1][nx] us[B][N][nu] K[B][N][nu][ndx] kff[B][N][nu] vs[B][N+1][nc_max] lams[B][N+1][ndx] */
int mpc_get_results(mpc_solver* s, double* xs, double* us, double* K, double* kff, double* vs, double* lams);
/* The gains of ONE knot: K_k[B][nu][ndx], kff_k[B][nu] (either may be NULL).  The scripts read controlFeedbacks()[0] only
 * (fulldynamic_talos.py:522, :550): fetching that one block instead of all N (1.9 MB per instance on the complete model) is what
 * the Python mirror does after every run. */
int mpc_get_gain(mpc_solver* s, int32_t k, double* K_k, double* kff_k);
/* workspace.problem_data.stage_data[k].dynamics_data.continuous_data.{xdot, constraint_datas[i].contact_force}
 * (fulldynamic_talos.py:465-480, kinodynamic_talos.py:432)
 * xdot[B][ndx], wrenches[B][2][6] (inactive contacts zero). */
int mpc_get_stage_data(mpc_solver* s, int32_t k, double* xdot, double* wrenches);

/* Solver-state checkpoint (the reference itself only logs, talos_utils.py:113-154): everything a handle needs to
 * continue a receding-horizon run — the stage tables of the horizon in knot order (ring unrolled), the iterate xs / us, the
 * multipliers vs / lams, the measured state x0 and the feedback mode, and per instance the penalty and the BCL tolerances — as one
 * array of doubles (integers stored exactly), portable between the libraries that export this ABI with the same dimensions
 * (a state saved by the HIP library restores into the oracle and vice versa).  mpc_state_size: doubles needed.  mpc_get_state: fills
 * buf (cap doubles), returns the count written or -1.  mpc_set_state: restores (dimensions must match mpc_create's); kept records
 * of tick reuse and the cut-Hessian guesses of the legs are invalidated, so the next tick evaluates everything afresh.  Per-instance parameter
 * tables (mpc_enable_instance_params) are NOT part of the state: every instance is handed the shared tables of the restored stages again, and with the
 * reference generator in the library (mpc_walk_init) the next mpc_walk_update rewrites the references of every knot from its plan (mpc_walk_get_state /
 * mpc_walk_set_state carry the plan itself).  Both libraries behave the same (tests/test_checkpoint.py). */
int64_t mpc_state_size(mpc_solver* s);
int64_t mpc_get_state(mpc_solver* s, double* buf, int64_t cap);
int mpc_set_state(mpc_solver* s, const double* buf, int64_t len);

/* Per-instance stage parameters.  The stage tables (descriptors AND parameters) of an ensemble are shared by its instances; after
 * mpc_enable_instance_params every instance has its own copy of the PARAMETER tables, so that references, targets, bounds and weights
 * can differ from robot to robot (the descriptors — which terms a stage has — stay shared).  mpc_set_stage / mpc_cycle /
 * mpc_update_stage_params(_batch) keep acting on every instance (a slot that receives a stage table is reset to it in every copy);
 * mpc_update_instance_params_batch patches the copies of single instances: patch i writes lens[i] doubles at offsets[i] of stage
 * ks[i] of instance insts[i]; values concatenated in `vals`.  Tick reuse invalidates a patched knot for the whole ensemble.
 * (mpc_get_state stores the shared tables only.)  May be mixed with mpc_walk_update on the same offsets: a patch that follows device-generated ticks always
 * reaches the device (the HIP library compares patches with a host mirror and skips unchanged ones; the generator marks what it wrote as unknown there). */
int mpc_enable_instance_params(mpc_solver* s);
int mpc_update_instance_params_batch(mpc_solver* s, int32_t count, const int32_t* insts, const int32_t* ks, const int32_t* offsets, const int32_t* lens, const double* vals);

/* Reference generation in the library (round 5; SURVEY.md section 8f N1, second half): the swing-foot generator of the walking loops
 * (talos_utils.py:187-327 footTrajectory.updateTrajectory / foot_trajectory, called at fulldynamic_talos.py:444-463) for EVERY instance of an ensemble
 * with per-instance parameter tables, from each instance's own predicted next state xs[1] — forward kinematics of the two sole frames, the foothold
 * rules, the Bezier swing curve and the 12-double placement references written straight into the instance's tables: no per-tick host work that grows
 * with the ensemble, nothing but four countdown integers crosses the boundary per tick (HIP: one kernel, a workgroup per instance).
 *   mpc_walk_init    after mpc_enable_instance_params.  frames: the two sole frames of the model table; offsets: where a running stage / the terminal
 *                    node keep the references (the lowering's slots; -1 = the problem has no such reference).
 *   mpc_walk_update  once per tick BEFORE mpc_cycle (as the loops call setReference before replaceStageCircular): the countdowns of
 *                    talos_utils.update_timings; `forward` != NULL = footTrajectory.updateForward (t_left[3], t_right[3], swing_apex) first.  On a tick
 *                    whose generator replans (a foot without a pending landing, a take-off inside the double-support window) every knot's reference is
 *                    rewritten and no record of the previous tick is reused; otherwise only the knot appended by the last mpc_cycle (which came with
 *                    the shared table's reference) and the terminal node.  Same rules and the same arithmetic in the oracle.
 *   mpc_walk_get_state / mpc_walk_set_state  start / final poses of both feet per instance, [B][4][12]: tests, and the checkpoint of a loop that is restarted
 *                    (after set_state the next update rewrites the references of every knot). */
typedef struct mpc_walk_config {
  int32_t T_ss, T_ds;
  int32_t frame_lf, frame_rf;
  int32_t off_lf, off_rf;        /* running stages: placement references (R row-major 9, p 3)                                  */
  int32_t off_xref_z;            /* running stages: base height of the posture reference (the stairs variant), or -1          */
  int32_t toff_com, toff_lf, toff_rf;  /* terminal node: CoM target (3) ; foot references (12 each) or -1                       */
  double swing_apex;
  double t_left[3], t_right[3];  /* footTrajectory.translationLeft / translationRight                                         */
  double rot_diff[9];            /* footTrajectory.rotationDiff                                                               */
  double com0[3];                /* terminal CoM target = (mid-point of the last foot references in x, y ; com0 z)              */
  double feet_z0, xref_z0;       /* stairs: posture height = xref_z0 + mean height of the knot's foot references - feet_z0      */
  double z_follow;               /* 1: stairs variant (off_xref_z, CoM target height follow the feet) ; 0: flat ground          */
  double lf0[12], rf0[12];       /* initial sole placements                                                                   */
  double floor_z;                /* (ABI 3) no foothold is planned below this height: the floor stops a foot.  Loops that feed the solver's own prediction
                                  * back (no simulator, no ground) otherwise sink: fulldynamic_talos.py:449 aims the left foot 1 cm BELOW the right one's
                                  * height at every step, which a floor stops and a prediction does not (5 - 6 cm over the reference's seven swings).
                                  * <= -1e300: no floor (stairs).                                                                    */
} mpc_walk_config;
int mpc_walk_init(mpc_solver* s, const mpc_walk_config* cfg);
int mpc_walk_update(mpc_solver* s, int32_t takeoff_RF, int32_t takeoff_LF, int32_t land_RF, int32_t land_LF, const double* forward);
int mpc_walk_get_state(mpc_solver* s, double* out);
int mpc_walk_set_state(mpc_solver* s, const double* in);

/* Failure policy of an ensemble.  isolate = 0 (default): a failed factorisation on any instance makes the run return an error, as a
 * single solver would.  isolate = 1: the instance is reported (mpc_stats.converged = -code: 2 / 3 / 4 Riccati blocks, 5 / 6 contact
 * dynamics), keeps the iterate it had when the pass started failing and is skipped by every later run until it is revived; the other
 * instances are not affected.  mpc_revive_instance(dst, src): dst takes over the iterate, multipliers and measured state of instance
 * src (e.g. the nominal one) and takes part again from the next run on. */
int mpc_set_failure_policy(mpc_solver* s, int32_t isolate);
int mpc_revive_instance(mpc_solver* s, int32_t dst, int32_t src);

/* Phase dumps for parity tests: copies the named per-knot quantity of instance b, knot k into out
 * (capacity cap doubles) and returns the number of doubles written (<0 on error).  Names:
 * "H" "grad" "AB" "f" "E6" "cval" "CD" "cost" "P" "p" "K" "kff" "Knu" "knu" "dx" "du" "dvs" "dlams" "xnext".
 * Parallel-in-time sweep (riccati_legs > 1), knots of a leg other than the last: "Mu" "Znu" "Lm" and, HIP, "Phi" "phi" "Gam" "Ku" "Knup" /
 * oracle, "Mx" "Mth" "Kth" "Knuth" "Kexact" (same quantities: Phi = Mx everywhere, Gam = Mth, Ku = Kth, Knup = Knuth at the last knot of a leg) ;
 * with k = index of the leg instead of a knot: "Sg" "sg" (HIP; the oracle keeps them per knot) "Zx" "zc" "calP" "calp" "theta" ;
 * HIP only: "ls_knot" (knot k of the last pass: the merit of every linesearch candidate alpha_i = 2^-i at this knot, then the knot's cost
 * and penalty at the current point), "fixed_dims" (one value: which fixed-dimension instantiations of the hot kernels serve this handle,
 * 0 = the generic ones ). */
int mpc_debug_get(mpc_solver* s, const char* name, int32_t b, int32_t k, double* out, int32_t cap);
/* Evaluate (value + derivatives) at the current iterate without stepping; fills the LQ knots. */
int mpc_debug_evaluate(mpc_solver* s, const double* xs, const double* us);

/* Per-kernel timing with device events on the solver's own stream (the reference only has wall-clock
 * timers around run(): fulldynamic_talos.py:538-543).  mpc_profile(s, 1) starts recording, (s, 0) stops,
 * (s, 2) clears; (s, 16 * mask), mask != 0, records only the kernel slots whose bit is set in mask (an event pair between two
 * kernels costs stream time: a timed region that wants one kernel's duration should not pay for all of them);
 * (s, 3) / (s, 4) switch the in-kernel phase timers (shader-clock counters of the Riccati and stage
 * kernels, read with mpc_debug_get("ric_prof")) on / off — developer tooling, off by default.  mpc_profile_read returns the number of kernel slots; for slot i it fills the kernel
 * name, the number of launches recorded and their summed duration in milliseconds.  The oracle reports
 * zero slots. */
int mpc_profile(mpc_solver* s, int32_t mode);
int mpc_profile_read(mpc_solver* s, int32_t slot, char* name, int32_t name_cap, int32_t* launches, double* total_ms);
/* Occupancy table of the kernels one pass of this handle launches (the "LDS / wave occupancy study" of BASELINE.json's kinodynamic
 * configuration: tools/occupancy_report.py).  Entry idx: info[8] = {threads per workgroup, VGPRs, scratch bytes per lane, static LDS bytes,
 * dynamic LDS bytes, workgroups one CU can hold (the runtime's occupancy calculator), workgroups per launch, wavefronts per SIMD at
 * that residency}.  Returns the number of entries (for any idx, also one out of range); the oracle reports zero. */
int mpc_kernel_info(mpc_solver* s, int32_t idx, char* name, int32_t name_cap, int32_t* info);

#ifdef __cplusplus
}
#endif
#endif /* MPC_ABI_H */
