// solver_args.h — argument block of the on-device ProxDDP kernels and the device helpers every kernel file shares
// (record addressing, tick-reuse predicate, DPP wave reductions).  Kernel definitions live in solver_kernels.h and the
// per-family headers, so that each translation unit of the library (Makefile) compiles only its own kernels.
#pragma once
#include "device_common.h"

// |dphi0| <= MPC_STALL_TOL (1 + |phi0|): no descent left in the inner problem (shared with oracle/solver.hpp)
#define MPC_STALL_TOL 1e-13

#define MPC_DIRTY_WORDS 4  // per-knot invalidation of tick reuse covers horizons up to 255 knots (longer ones: any update clears all)

struct SolverArgs {
  Layout L;
  mpc_options opt;
  int head;  // ring-buffer head of the stage table
  // stage tables
  const int32_t* stage_desc;
  const double* stage_params;
  const double* inst_params;  // per-instance copies of the parameter tables [B][N + 1][max_stage_doubles], or nullptr (mpc_enable_instance_params)
  const int32_t* model_i;
  const double* model_d;
  // iterate
  double *xs, *us, *vs, *lams, *vs_e, *lams_e, *x0;
  double *dxs, *dus, *dvs, *dlams;
  // Tick reuse (MPC ticks with max_iters = 1): the full step of tick t is evaluated WITH derivatives straight into the knot
  // records, which are ring-indexed like the stage tables; when it is accepted, tick t + 1 finds the records of its knots
  // 0 .. N-2 already there and only re-projects the constraint values under the fresh multipliers.
  int isolate;    // mpc_set_failure_policy: a failed instance is remembered (InstState::converged = -code) and skipped instead of failing the run
  int khead;      // ring head of the N running knot records (the terminal record has its own slot)
  int spec_on;    // the alpha = 1 candidate of this pass writes full records (k_eval_multibody<3>)
  int reuse_on;   // this tick may reuse records marked valid in spec[]
  int reuse_k0;   // ... knot 0 included (perfect-model feedback: the measured state is the predicted one)
  int reuse_same; // a further pass of the same run: the records were written by the previous pass's full-step candidate for these very knots (no shift in between)
  int* spec;      // [B] 1: the records of instance b hold the evaluation of its current iterate shifted by one knot
  double* abdz;  // [B][N][n]: [A B] [dx; du] per knot, written by the forward sweep for k_duals (nullptr: k_duals forms it itself)
  double *knots, *gains, *work;
  double *trial_phi;  // [B][n_alpha][N+1]
  InstState* inst;
  int* all_done;  // unused: the host reads the per-instance status instead
  double* prof;  // [B][64] phase cycle counters: 0..31 Riccati kernel, 32..63 whole-body stage kernel (knot 1)
  // parallel-in-time Riccati (legs.h): number of legs of this pass (1 = serial sweep) and the per-(instance, leg) records
  int nlegs;
  int leg_cap;     // capacity (in legs) of legbuf / treebuf / work: grown by mpc_set_options when more legs are asked for
  int leg_guess;   // 1: leg j starts from the Hessian calP_{j+1} its record holds from the previous pass / tick (0: from zero)
  double* legbuf;  // [B][MPC_MAX_LEGS - 1][leg_stride]
  double* treebuf; // [B][MPC_MAX_LEGS - 1][tree_stride]: inner nodes of the tree over the cuts (legs_tree.h)
  // tick reuse: spare knot record per instance for the speculative evaluation of the knot the next tick appends (eval_multibody.h) ;
  // spec_next: this tick's appended stage has the table the speculation assumed (knots N - 1 and N are reused too)
  double* spec_knot;
  int spec_next;
  // knots (bit k of word k / 64) whose stage parameters changed since their record was written (setReference, a rebuilt terminal
  // constraint: mpc_update_stage_params): the launch of the current point evaluates them afresh, the others stay reused
  unsigned long long dirty[MPC_DIRTY_WORDS];
  // >= 0: the launch of the current point evaluates THIS knot only, whatever tick reuse says (the warm-start refinement of the appended
  // knot, mpc_options.refine_appended_knot) ; -1: all knots
  int only_knot;
  int corrector_on;  // mpc_options.corrector_prim_tol applies to this run (corrector_window: the host knows which runs follow a change of the contact pattern)
  int reject_failed;  // developer experiment (MPC_HIP_REJECT_FAILED=1): a linesearch whose last candidate still fails the Armijo test takes NO step instead of that candidate
  int tree_pivoted;  // 1: k_leg_compose skips its blocked elimination on the matrix cores and goes straight to the pivoted Gauss-Jordan (MPC_HIP_TREE_PIVOTED=1: developer comparison)
};

// first knot of leg j (leg nlegs - 1 ends with the terminal knot) — the rule of oracle/solver.hpp leg_start
DEV int leg_start(const SolverArgs& a, int j) { return (int)((long long)j * a.L.N / a.nlegs); }
DEV int leg_of_knot(const SolverArgs& a, int k) { int j = a.nlegs - 1; while (j > 0 && leg_start(a, j) > k) --j; return j; }
DEV double* leg_ptr(const SolverArgs& a, int b, int j) { return a.legbuf + ((size_t)b * (a.leg_cap - 1) + j) * a.L.leg_stride; }  // leg_cap: legs the buffers were sized for

DEV int knot_slot(const SolverArgs& a, int k) { return k < a.L.N ? (a.khead + k) % a.L.N : a.L.N; }
DEV double* knot_ptr(const SolverArgs& a, int b, int k) { return a.knots + ((size_t)b * (a.L.N + 1) + knot_slot(a, k)) * a.L.knot_stride; }
// true if tick reuse applies to knot k of instance b: its record is already the evaluation of the current iterate
// (knot N - 1, the appended one: if the previous tick evaluated it speculatively with the table that was then appended — spec_next ;
// knot N: the terminal state and its table are those of the previous tick)
DEV bool knot_reused(const SolverArgs& a, int b, int k) {
  if (!(a.reuse_on && a.spec[b])) return false;
  if ((a.dirty[(k >> 6) & (MPC_DIRTY_WORDS - 1)] >> (k & 63)) & 1ull) return false;
  if (a.reuse_same) return true;
  if (k < a.L.N - 1) return k > 0 || a.reuse_k0;
  return a.spec_next != 0 && a.spec[b] == 1;
}
DEV double* gain_ptr(const SolverArgs& a, int b, int k) { return a.gains + ((size_t)b * (a.L.N + 1) + k) * a.L.gain_stride; }
DEV int stage_slot(const SolverArgs& a, int k) { return k < a.L.N ? (a.head + k) % a.L.N : a.L.N; }

// Sum over the 64 lanes on the DPP network (row shifts inside 16-lane rows, then the two row broadcasts of gfx9) — no
// LDS permutes (ds_bpermute, what __shfl_down compiles to, costs an LDS round trip per step); result broadcast from lane 63.
template <int CTRL, int ROW_MASK>
DEV double dpp_add(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xf, true);
  return v + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
DEV double wave_sum(double v) {
  v = dpp_add<0x111, 0xf>(v);  // row_shr:1
  v = dpp_add<0x112, 0xf>(v);  // row_shr:2
  v = dpp_add<0x114, 0xf>(v);  // row_shr:4
  v = dpp_add<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of every row holds the row sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROW_MASK>
DEV double dpp_max(double v) {  // v >= 0 (infeasibility measures): lanes without a source read 0
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(bits & 0xffffffffll), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, ROW_MASK, 0xf, true);
  return fmax(v, __longlong_as_double(((long long)hi << 32) | (unsigned int)lo));
}
DEV double wave_max_nonneg(double v) {
  v = dpp_max<0x111, 0xf>(v);
  v = dpp_max<0x112, 0xf>(v);
  v = dpp_max<0x114, 0xf>(v);
  v = dpp_max<0x118, 0xf>(v);
  v = dpp_max<0x142, 0xa>(v);
  v = dpp_max<0x143, 0xc>(v);
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(bits >> 32), 63);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
