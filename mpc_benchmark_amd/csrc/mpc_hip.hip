// mpc_hip.hip — the product library: C-ABI of include/mpc_abi.h on top of hand-written HIP kernels for
// gfx950 (MI355X).  Host side = thin orchestration (allocation, table upload, kernel sequence of one
// ProxDDP iteration, result download); every floating-point operation of the hot path runs on the device.
// There is no CPU fallback in this file: any HIP failure is reported through the return code.
#include <hip/hip_runtime.h>
#include <mutex>
#include <map>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

// (-DCHOL16_MFMA: the 16 x 16 Cholesky + inverse of this translation unit — the sweep's Ruu and Schur blocks, the leg kernels — on the matrix cores, mfma_blocks.h
// chol16_wave_mfma.  Round 6 measured it again: sweep 1.527 -> 1.507 ms per launch, 1.461 with its identity panels skipped — and the kinodynamic stairs ensemble, whose
// closing step is decided at round-off level, lost an instance it keeps with the register form (tests/test_gpu_walk_all_problems.py).  Not the default: the register form
// with the padding columns left out — chol16_wave(..., ncols), the same bits on the real ones — gets the Schur block's share without touching a single result.)
#include "eval_multibody_host.h"
#include "eval_reuse_kernels.h"
#include "eval_vector.h"
#include "riccati_mfma.h"
#include "closed_loop.h"
#include "legs.h"
#include "legs_tree.h"
#include "pipeline_glue.h"
#include "qp_device_api.h"

#define HIP_OK(expr)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (expr);                                                                           \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a setting of the (device, kernel) pair, not of a handle: a process may hold handles
// whose carve-outs differ (another number of constraint rows, another model) — the largest request so far stays
static void lds_attr_max(const void* fn, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> have;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  int& h = have[{dev, fn}];
  if (h < bytes) {
    HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    h = bytes;
  }
}

// Problems whose dimensions get compile-time instantiations of the sweep and the leg kernels (riccati_mfma.h "FN, FM"):
//   X(id, n, m, gfull, st_lds, NP, MP)   — (n, m): state / control dimension ; gfull, st_lds: the sweep's LDS plan for them (make_ric_lds) ;
//   NP, MP: padded state / control dimension (the template arguments of the tree and of k_leg_knot)
// 1: complete Talos (38 dofs), full dynamics   2: complete Talos, kinodynamic   3: Talos with the upper body locked as the scripts lock it
// (28 dofs), full dynamics   4: the same, kinodynamic.  The stage kernel has its own list (eval_multibody.hip).
#define MPC_FIXED_MODELS(X) X(1, 76, 32, 1, 1, 80, 32) X(2, 76, 44, 3, 0, 80, 48) X(3, 56, 22, 3, 1, 64, 32) X(4, 56, 34, 3, 1, 64, 48)

struct mpc_solver {
  mpc_dims dims{};
  mpc_options opt{};
  Layout L{}, LT{};
  int head = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // device buffers
  int32_t* d_stage_desc = nullptr;
  double* d_stage_params = nullptr;
  int32_t* d_model_i = nullptr;
  double* d_model_d = nullptr;
  double *d_xs = nullptr, *d_us = nullptr, *d_vs = nullptr, *d_lams = nullptr, *d_vs_e = nullptr, *d_lams_e = nullptr, *d_x0 = nullptr;
  double *d_dxs = nullptr, *d_dus = nullptr, *d_dvs = nullptr, *d_dlams = nullptr, *d_abdz = nullptr;
  double *d_xs_alt = nullptr, *d_us_alt = nullptr;  // second pair of iterate buffers: the warm-start shift writes out of place, then the pairs swap
  double *d_knots = nullptr, *d_tknots = nullptr, *d_gains = nullptr, *d_work = nullptr, *d_trial_phi = nullptr, *d_mbwork = nullptr;
  InstState* d_inst = nullptr;
  int* d_all_done = nullptr;
  int async_passes[2] = {1, 1};  // passes enqueued by mpc_run_shifted_async per slot in flight
  double* d_prof = nullptr;
  bool phase_timers = false;
  std::vector<void*> allocs;
  // host mirrors of the stage tables (needed for ring-buffer bookkeeping and debug)
  std::vector<int32_t> h_desc;
  std::vector<double> h_params;            // host mirror of the stage parameters (an unchanged stage is not uploaded again)
  std::vector<int32_t> h_len;              // per slot: n_desc, n_params of the mirror (0, 0 = nothing uploaded yet)
  std::vector<int> h_model_i;
  bool have_model = false;
  size_t mb_work_stride = 0;
  bool perfect_feedback = false;
  // tick reuse (mpc_set_tick_reuse): see SolverArgs
  bool isolate = false;  // mpc_set_failure_policy
  // per-instance stage parameters (mpc_enable_instance_params): every instance has its own copy of the parameter tables [B][N + 1][max_stage_doubles]
  double* d_inst_params = nullptr;
  std::vector<double> h_inst_params;
  bool contact_dyn = false;  // some stage uploaded so far has contact-constrained dynamics (sticky): the stage kernel's LDS carve-out holds the factor of M
  bool tick_reuse = false, reuse_this_pass = false, reuse_same_now = false;
  int pass_in_run = 0;  // index of the pass being enqueued within its run
  bool pass_is_corrector = false;  // the pass being enqueued is the extra one mpc_run_shifted_async adds for the corrector rule (most workgroups sit it out)
  // speculative evaluation of the appended knot (eval_multibody.h): the spare records hold one made with the table of the then last
  // stage (spec_rec_valid) ; the stage appended since is that table (spec_next_pending, set by mpc_cycle) ; this pass may use it
  double* d_spec_knot = nullptr;
  bool spec_rec_valid = false, spec_next_pending = false, spec_next_now = false;
  int cycles_since_run = 0;  // the speculation assumes ONE mpc_cycle per tick (replaceStageCircular + cycleAppend of the scripts)
  int khead = 0;
  int* d_spec = nullptr;
  double* d_fext = nullptr;  // [B][3] disturbance force of mpc_simulate_push
  int only_knot = -1;              // SolverArgs::only_knot of the launches being enqueued
  bool appended_changed = false;   // the stage of the last mpc_cycle has another contact pattern than its predecessor (refine_appended_knot)
  bool appended_any = false;       // a stage was appended since the last run (refine_appended_knot < 0: refine after every cycle)
  // mpc_walk_*: reference generation on the device (k_walk_refs)
  bool walk_on = false, walk_force_all = false;
  std::vector<uint8_t> walk_poisoned;  // per ring slot: the host mirror of the ranges k_walk_refs writes holds NaN (see mpc_walk_update)
  mpc_walk_config walk{};
  double* d_walk_state = nullptr;  // [B][48]
  int since_change = 1 << 20;      // mpc_cycle calls since the appended stage last changed its contact pattern (corrector_window)
  bool refine_now = false;         // ... and this run refines the warm start of the appended knot after k_begin_run
  double* d_simu = nullptr;  // [B][nu] torques, [B][12] wrenches of mpc_simulate_torque
  double* d_simwr = nullptr;
  // per-slot invalidation (mpc_update_stage_params*): slots whose parameters changed since the last pass was enqueued ; dirty_all:
  // an update on a horizon too long for the mask of SolverArgs
  std::vector<uint8_t> slot_dirty;
  bool dirty_all = false;
  bool spec_skip_pass = false;  // this pass neither reuses records nor writes speculative ones (begin_reuse_pass)
  unsigned long long dirty_now[MPC_DIRTY_WORDS] = {};  // knot mask of the pass being enqueued (set in begin_reuse_pass)
  // parameter patches travel through a pinned ring as ONE host-to-device copy + a scatter kernel, stream-ordered (no host wait)
  static constexpr int PATCH_RING = 4;
  char* patch_pin[PATCH_RING] = {};
  size_t patch_cap[PATCH_RING] = {};
  hipEvent_t patch_ev[PATCH_RING] = {};
  int patch_next = 0;
  char* d_patch = nullptr;
  size_t d_patch_cap = 0;
  // the measured state of the NEXT tick (xs[1] of every instance) snapshotted with the status of an asynchronous tick
  double* h_xnext[2] = {nullptr, nullptr};  // [ASYNC_DEPTH]
  std::vector<double> leg_mu;  // penalty of every instance when the cut Hessians of the legs were last refreshed
  // asynchronous ticks (mpc_run_shifted_async / mpc_wait): status snapshots in pinned host memory, one event each; up to
  // ASYNC_DEPTH ticks may be in flight, so that the next tick is already queued while the host looks at the previous one
  static constexpr int ASYNC_DEPTH = 2;
  InstState* h_status[ASYNC_DEPTH] = {nullptr, nullptr};
  hipEvent_t status_ev[ASYNC_DEPTH] = {nullptr, nullptr};
  int async_head = 0, async_pending = 0;
  // pinned staging ring for stage-table uploads: the copy is stream-ordered and the host does not wait for the ticks in flight
  // (a stage that differs from its slot's mirror arrives at every contact-phase switch)
  static constexpr int STAGE_RING = 8;
  char* stage_pin[STAGE_RING] = {};
  hipEvent_t stage_ev[STAGE_RING] = {};
  int stage_next = 0;
  RicLds ric{};
  int ric_fixed = 0;  // id of the fixed-dimension instantiations of the sweep and the leg kernels that serve this handle (MPC_FIXED_MODELS below), 0: the generic kernels
  ClLds cl{};
  bool use_mfma_riccati = false;
  // parallel-in-time legs (legs.h)
  LkLds lk{};
  LcLds lc{};
  LxLds lx{};
  bool legs_ok = false;  // the dimensions fit the leg kernels (np <= 80, mp <= 48, at most 256 constraint rows, LDS carve-outs) and MPC_HIP_NO_LEGS is unset; otherwise riccati_legs > 1 silently keeps the serial sweep (mpc_abi.h)
  // developer knobs, read from the environment ONCE when the handle is created (MPC_LEGS_CHAIN, MPC_LEGS_PLAIN, MPC_TREE_SWEEPS, MPC_HIP_TRACE)
  bool env_chain = false, env_plain = false, env_reject_failed = false;
  int env_tree_sweeps = 0, env_trace = -1;
  double* d_legbuf = nullptr;
  double* d_treebuf = nullptr;
  int leg_cap = 0;  // legs the three leg-indexed buffers (d_work, d_legbuf, d_treebuf) are sized for: they grow on demand (ensure_leg_capacity)
  TreeDesc tree{};       // tree over the cuts for the current number of legs (legs_tree.h)
  // three legs or more: the cuts are resolved by a tree of pairwise compositions (legs_tree.h) instead of the chain of k_leg_consensus
  // (fewer solves in a row, and the compositions of a level run side by side: 64 x 4 legs 0.38 -> 0.27 ms, batch 1 with 16 legs is only
  // worth it this way) ; MPC_LEGS_CHAIN=1 keeps the chain (tests: its intermediates are compared with the oracle's one to one)
  bool use_tree() const {
    return eff_legs() >= 3 && d_treebuf != nullptr && !env_chain;
  }
  bool leg_guess_valid = false;  // the leg records hold the cut Hessians of an earlier pass (terminal costs of the legs)
  int leg_guess_now = 0;
  int eff_legs() const {
    if (!legs_ok) return 1;
    int J = opt.riccati_legs;
    if (J > MPC_MAX_LEGS) J = MPC_MAX_LEGS;
    if (J > L.N) J = L.N;
    return J < 1 ? 1 : J;
  }
  // per-kernel timing (mpc_profile): event pairs recorded around every launch while enabled
  struct ProfSlot {
    const char* name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
  };
  bool profiling = false;
  unsigned prof_mask = ~0u;  // kernel slots that get event pairs while profiling (an event pair costs stream time)
  std::vector<ProfSlot> prof;

  template <class F> void timed(int slot, const char* name, F&& launch) {
    // (the launches of a corrector pass — mpc_options.corrector_prim_tol: most workgroups sit it out — are not timed: the per-kernel figures of the
    // profile describe launches that process every instance)
    // (only that pass: in the synchronous run the passes after max_iters are BCL updates / extra iterations that serve every instance)
    if (!profiling || !((prof_mask >> slot) & 1u) || pass_is_corrector) { launch(); return; }
    if ((int)prof.size() <= slot) prof.resize(slot + 1);
    ProfSlot& p = prof[slot];
    p.name = name;
    if (p.used == p.ev.size()) {
      hipEvent_t e0, e1;
      HIP_OK(hipEventCreate(&e0));
      HIP_OK(hipEventCreate(&e1));
      p.ev.emplace_back(e0, e1);
    }
    HIP_OK(hipEventRecord(p.ev[p.used].first, stream));
    launch();
    HIP_OK(hipEventRecord(p.ev[p.used].second, stream));
    p.used++;
  }

  template <class T> T* alloc(size_t count) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, count * sizeof(T) + 64));
    HIP_OK(hipMemsetAsync(p, 0, count * sizeof(T) + 64, stream));
    allocs.push_back(p);
    return (T*)p;
  }

  // mpc_options.corrector_prim_tol / corrector_window: does the corrector rule apply to the run that starts now? (same rule: oracle/capi.cpp)
  bool corrector_armed() const { return opt.corrector_prim_tol > 0.0 && (opt.corrector_window <= 0 || since_change < opt.corrector_window); }
  SolverArgs args() const {
    SolverArgs a;
    a.L = L; a.opt = opt; a.head = head;
    a.stage_desc = d_stage_desc; a.stage_params = d_stage_params; a.model_i = d_model_i; a.model_d = d_model_d;
    a.xs = d_xs; a.us = d_us; a.vs = d_vs; a.lams = d_lams; a.vs_e = d_vs_e; a.lams_e = d_lams_e; a.x0 = d_x0;
    a.dxs = d_dxs; a.dus = d_dus; a.dvs = d_dvs; a.dlams = d_dlams; a.abdz = nullptr;
    a.khead = khead; a.spec = d_spec; a.isolate = isolate ? 1 : 0; a.inst_params = d_inst_params;
    // an MPC tick: one iteration (the reference loop) or a few (max_iters <= 4).  The last pass of a replanning tick (spec_skip_pass)
    // evaluates its candidate value-only; its first pass reuses nothing.
    const bool mpc_tick = tick_reuse && opt.max_iters <= 4 && L.space == MPC_SPACE_MULTIBODY;
    a.spec_on = (mpc_tick && !(spec_skip_pass && pass_in_run >= opt.max_iters - 1)) ? 1 : 0;
    a.reuse_on = (mpc_tick && reuse_this_pass && !(spec_skip_pass && pass_in_run == 0)) ? 1 : 0;
    a.reuse_k0 = perfect_feedback ? 1 : 0;
    a.reuse_same = (a.reuse_on && reuse_same_now) ? 1 : 0;
    a.spec_knot = a.spec_on ? d_spec_knot : nullptr; a.spec_next = (a.reuse_on && spec_next_now) ? 1 : 0;
    for (int w = 0; w < MPC_DIRTY_WORDS; ++w) a.dirty[w] = a.reuse_on ? dirty_now[w] : 0ull;
    a.only_knot = only_knot;
    a.corrector_on = corrector_armed() ? 1 : 0;
    a.reject_failed = env_reject_failed ? 1 : 0;
    a.tree_pivoted = getenv("MPC_HIP_TREE_PIVOTED") ? 1 : 0;
    a.nlegs = eff_legs(); a.leg_cap = leg_cap; a.legbuf = d_legbuf; a.treebuf = d_treebuf; a.leg_guess = leg_guess_now;
    a.knots = d_knots; a.gains = d_gains; a.work = d_work; a.trial_phi = d_trial_phi; a.inst = d_inst; a.all_done = d_all_done; a.prof = phase_timers ? d_prof : nullptr;
    return a;
  }
  size_t riccati_lds() const {
    return ((size_t)L.n * L.n + (size_t)L.m * L.m + (size_t)L.sc_cap * L.sc_cap) * sizeof(double) + ((size_t)L.c + 4) * sizeof(int);
  }
};

// Blocking copy on the handle's own stream.  (A plain hipMemcpy goes through the null stream, which waits for the work of
// every other handle on the device: shards of one GPU would then run in lock step.)
static void copy_sync(mpc_solver* s, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  HIP_OK(hipMemcpyAsync(dst, src, bytes, kind, s->stream));
  HIP_OK(hipStreamSynchronize(s->stream));
}

// The leg-indexed buffers are sized for the number of legs in use (4 in the benchmark: 0.3 GB for 64 instances of the complete model
// instead of 2.3 GB for MPC_MAX_LEGS = 32) and re-allocated when more are asked for.  Their contents are scratch (d_work) or guesses
// that a change of the number of legs invalidates anyway (cut Hessians, tree nodes).
static void ensure_leg_capacity(mpc_solver* s, int legs) {
  const Layout& L = s->L;
  int need = legs < 1 ? 1 : legs;
  if (need >= 3 && need < 4) need = 4;  // (the tree keeps one scratch record per instance beside its inner nodes)
  if (need <= s->leg_cap) return;
  HIP_OK(hipStreamSynchronize(s->stream));
  auto regrow = [&](double*& p, size_t count) {
    if (p) HIP_OK(hipFree(p));
    p = nullptr;
    HIP_OK(hipMalloc((void**)&p, (count ? count : 1) * sizeof(double)));
    HIP_OK(hipMemsetAsync(p, 0, (count ? count : 1) * sizeof(double), s->stream));
  };
  const size_t B = L.B;
  regrow(s->d_work, B * need * (size_t)L.work_stride);
  if (s->legs_ok) {
    regrow(s->d_legbuf, B * (size_t)(need > 1 ? need - 1 : 1) * L.leg_stride);
    regrow(s->d_treebuf, B * (size_t)need * L.tree_stride);  // inner nodes + one scratch record per instance
  }
  s->leg_cap = need;
  s->leg_guess_valid = false;
  HIP_OK(hipStreamSynchronize(s->stream));
}

static void create_impl(mpc_solver* s, const mpc_dims& d) {
  s->dims = d;
  { const char* e;
    e = getenv("MPC_LEGS_CHAIN"); s->env_chain = e && atoi(e) > 0;
    e = getenv("MPC_LEGS_PLAIN"); s->env_plain = e && atoi(e) > 0;
    e = getenv("MPC_HIP_REJECT_FAILED"); s->env_reject_failed = e && atoi(e) > 0;
    e = getenv("MPC_TREE_SWEEPS"); s->env_tree_sweeps = e ? atoi(e) : 0;
    e = getenv("MPC_HIP_TRACE"); s->env_trace = e ? atoi(e) : -1; }
  HIP_OK(hipSetDevice(d.device));
  // Own non-blocking stream per handle: the shards of an ensemble on one GPU (several handles) run out of phase, the
  // sequential Riccati sweep of one shard (B workgroups) beside the wide per-knot kernels of the others.  Equal priorities:
  // alternating high / low priorities starved the low-priority shards (their tick took 16 - 21 ms against 14.5 ms).
  {
    // developer experiment: MPC_HIP_STREAM_PRIO=1 gives the handles of a process descending stream priorities
    static int created = 0;
    const char* pe = getenv("MPC_HIP_STREAM_PRIO");
    if (pe && atoi(pe) > 0) {
      int lo = 0, hi = 0;
      HIP_OK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo = least priority (largest number)
      const int levels = lo - hi + 1;
      const int prio = hi + (created++ % levels);
      HIP_OK(hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, prio));
      if (atoi(pe) > 1) fprintf(stderr, "[mpc] stream priority %d (range %d..%d)\n", prio, hi, lo);
    } else {
      HIP_OK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    }
  }
  Layout& L = s->L;
  L.N = d.horizon; L.B = d.batch; L.space = d.space; L.nx = d.nx; L.n = d.ndx; L.m = d.nu; L.c = d.nc_max > 0 ? d.nc_max : 1;
  L.nj = 0;
  L.max_stage_ints = d.max_stage_ints; L.max_stage_doubles = d.max_stage_doubles;
  L.n_alpha = 8;
  make_layout(L);
  // compact record for linesearch candidates: only what the value-only pass writes
  Layout& T = s->LT;
  T = L;
  {
    int o = 0;
    auto take = [&](int cnt) { int r = o; o += align2(cnt); return r; };
    T.oCV = take(L.c); T.oCT = take(L.c); T.oLO = take(L.c); T.oHI = take(L.c); T.oDT = take(L.c); T.oACT = take(L.c);
    T.oF = take(L.n); T.oMISC = take(MISC_COUNT);
    T.oH = T.oG = T.oAB = T.oE6 = T.oT6k = T.oD12 = T.oCD = T.oXD = T.oWR = T.oXN = 0;  // never written in value-only mode
    T.knot_stride = o;
  }
  if (s->riccati_lds() > 160 * 1024) throw std::runtime_error("problem dimensions exceed the LDS budget of the Riccati kernel");
  const size_t B = d.batch, N1 = d.horizon + 1, N = d.horizon;
  s->d_stage_desc = s->alloc<int32_t>(N1 * L.max_stage_ints);
  s->d_stage_params = s->alloc<double>(N1 * L.max_stage_doubles);
  s->d_xs = s->alloc<double>(B * N1 * L.nx); s->d_us = s->alloc<double>(B * N * L.m + 1);
  s->d_xs_alt = s->alloc<double>(B * N1 * L.nx); s->d_us_alt = s->alloc<double>(B * N * L.m + 1);
  s->d_vs = s->alloc<double>(B * N1 * L.c); s->d_lams = s->alloc<double>(B * (N1 + 1) * L.n);
  s->d_vs_e = s->alloc<double>(B * N1 * L.c); s->d_lams_e = s->alloc<double>(B * (N1 + 1) * L.n);
  s->d_x0 = s->alloc<double>(B * L.nx);
  s->d_dxs = s->alloc<double>(B * (N1 + 1) * L.n); s->d_dus = s->alloc<double>(B * N * L.m + 1);
  s->d_dvs = s->alloc<double>(B * N1 * L.c); s->d_dlams = s->alloc<double>(B * (N1 + 1) * L.n); s->d_abdz = s->alloc<double>(B * N1 * L.n);
  s->d_knots = s->alloc<double>(B * N1 * L.knot_stride);
  s->d_tknots = s->alloc<double>(B * L.n_alpha * N1 * T.knot_stride);
  s->d_gains = s->alloc<double>(B * N1 * L.gain_stride);
  // d_work (per (leg, instance): the legs of one instance run side by side), d_legbuf, d_treebuf: ensure_leg_capacity
  s->d_trial_phi = s->alloc<double>(B * L.n_alpha * N1);
  s->d_inst = s->alloc<InstState>(B);
  s->d_all_done = s->alloc<int>(4);
  s->d_spec = s->alloc<int>(B);
  s->d_prof = s->alloc<double>(B * 64);
  s->h_desc.assign(N1 * L.max_stage_ints, 0);
  s->h_params.assign(N1 * (size_t)L.max_stage_doubles, 0.0);
  s->h_len.assign(2 * N1, 0);
  s->slot_dirty.assign(N1, (uint8_t)0);
  s->leg_mu.assign((size_t)L.B, -1.0);
  // default options
  mpc_options& o = s->opt;
  o.tol = 1e-5; o.mu_init = 1e-8; o.dyn_al_scale = 1e-3; o.reg_init = 1e-9; o.ls_armijo_c1 = 1e-4; o.ls_alpha_min = 1e-7;
  o.bcl_prim_alpha = 0.1; o.bcl_prim_beta = 0.9; o.bcl_dual_alpha = 1.0; o.bcl_dual_beta = 1.0;
  o.bcl_mu_update_factor = 0.01; o.bcl_mu_lower_bound = 1e-8; o.inner_tol0 = 1.0; o.prim_tol0 = 1.0;
  o.max_iters = 100; o.max_al_iters = 100; o.force_initial_condition = 1; o.rollout_linear = 1; o.ls_max_steps = 8;
  o.num_threads = 1; o.riccati_legs = 1; o.forward_mode = 0;
  lds_attr_max((const void*)k_riccati_backward, 160 * 1024);
  s->ric = make_ric_lds(L.n, L.m, L.c, 1);
  // whole-body dynamics rows come in factored form (layout.h, oD12): the sweep multiplies with the v rows of [A B] only
  const bool sq_ok = L.space == MPC_SPACE_MULTIBODY && L.n % 2 == 0 && L.n >= 24 && !getenv("MPC_HIP_DENSE_AB");
  if (sq_ok && (s->ric.total_bytes > 160 * 1024 || !s->ric.ovl) && !getenv("MPC_HIP_FULL_AB")) {
    // ... and then only those rows need to be in LDS (plan 3 of make_ric_lds): room for G_u and for the factor of Ruu beside [A B] where plan 1 has none (m > 32)
    RicLds c3 = make_ric_lds(L.n, L.m, L.c, 3);
    if (c3.total_bytes > 160 * 1024) c3 = make_ric_lds(L.n, L.m, L.c, 3, 0);
    if (c3.total_bytes <= 160 * 1024 && c3.ovl && (c3.st_lds || c3.mp * c3.np <= L.n * L.nz)) s->ric = c3;
  }
  if (s->ric.total_bytes > 160 * 1024 && ((L.n + 15) & ~15) * ((L.m + 15) & ~15) <= L.n * L.n) {
    s->ric = make_ric_lds(L.n, L.m, L.c, 2);                                         // whole G, its u part in the L2 scratch (large m)
    if (s->ric.total_bytes > 160 * 1024) s->ric = make_ric_lds(L.n, L.m, L.c, 2, 0);
  }
  if (s->ric.total_bytes > 160 * 1024) s->ric = make_ric_lds(L.n, L.m, L.c, 0);     // panel-wise G when the whole G does not fit
  if (s->ric.total_bytes > 160 * 1024) s->ric = make_ric_lds(L.n, L.m, L.c, 0, 0);  // large m: Sh^T out of LDS as well
  if (sq_ok && s->ric.gfull) { s->ric.sq = 1; s->ric.nv = L.n / 2; }
  s->cl = make_cl_lds(L.n, L.m);
  if (s->cl.total_bytes <= 160 * 1024)
    lds_attr_max((const void*)k_closed_loop, s->cl.total_bytes);
  s->use_mfma_riccati = s->ric.total_bytes <= 160 * 1024 && (s->ric.st_lds || s->ric.mp * s->ric.np <= L.n * L.nz) && !getenv("MPC_HIP_GENERIC_RICCATI");
  if (s->use_mfma_riccati)
  {
    lds_attr_max((const void*)k_riccati_mfma<RIC_THREADS, 96>, s->ric.total_bytes);
    lds_attr_max((const void*)k_riccati_mfma<RIC_THREADS, 80>, s->ric.total_bytes);
    lds_attr_max((const void*)k_riccati_mfma<RIC_THREADS, 96, true>, s->ric.total_bytes);
    lds_attr_max((const void*)k_riccati_mfma<RIC_THREADS, 80, true>, s->ric.total_bytes);
    lds_attr_max((const void*)k_riccati_mfma<RIC_SMALL_THREADS, 16>, s->ric.total_bytes);
  }
  // parallel-in-time legs: every kernel of legs.h keeps three n x n operands (or [A B] + Pt) in LDS
  s->lk = make_lk_lds(L.n, L.m); s->lc = make_lc_lds(L.n); s->lx = make_lx_lds(L.n, L.m);
  s->legs_ok = s->use_mfma_riccati && s->ric.np <= 80 && s->ric.mp <= 48 && L.c <= 256 && s->ric.gfull >= 1 && s->lk.total_bytes <= 160 * 1024 &&
               s->lc.total_bytes <= 160 * 1024 && s->lx.total_bytes <= 160 * 1024 && !getenv("MPC_HIP_NO_LEGS");
  if (s->legs_ok) {
    lds_attr_max((const void*)k_riccati_mfma<RIC_THREADS, 80, true, true>, s->ric.total_bytes);
    // fixed-dimension instantiations (MPC_FIXED_MODELS): only when the handle's LDS plan IS the one they were compiled for
    s->ric_fixed = 0;
    if (!getenv("MPC_HIP_GENERIC_DIMS") && s->ric.sq) {
#define X(ID, FN, FM, GF, ST, NPV, MPV) if (!s->ric_fixed && L.n == FN && L.m == FM && L.nz == FN + FM && ric_same_layout(s->ric, ric_fixed_layout(FN, FM, true, GF, ST))) s->ric_fixed = ID;
      MPC_FIXED_MODELS(X)
#undef X
    }
#define X(ID, FN, FM, GF, ST, NPV, MPV) if (s->ric_fixed == ID) { \
      lds_attr_max((const void*)k_riccati_mfma<RIC_THREADS, 80, true, true, FN, FM, GF, ST>, s->ric.total_bytes); \
      lds_attr_max((const void*)k_leg_knot<MPV, FN, FM>, s->lk.total_bytes); \
      lds_attr_max((const void*)k_leg_condense<FN>, s->lc.total_bytes); \
      lds_attr_max((const void*)k_leg_compose<NPV, FN, FM>, s->lx.total_bytes); \
      lds_attr_max((const void*)k_leg_tree_down<NPV, FN, FM>, s->lx.total_bytes); }
    MPC_FIXED_MODELS(X)
#undef X
    lds_attr_max((const void*)k_riccati_mfma<RIC_THREADS, 80, false, true>, s->ric.total_bytes);
    lds_attr_max((const void*)k_riccati_mfma<RIC_SMALL_THREADS, 16, false, true>, s->ric.total_bytes);
    lds_attr_max((const void*)k_leg_knot<16>, s->lk.total_bytes);
    lds_attr_max((const void*)k_leg_knot<32>, s->lk.total_bytes);
    lds_attr_max((const void*)k_leg_knot<48>, s->lk.total_bytes);
    lds_attr_max((const void*)k_leg_condense<0>, s->lc.total_bytes);
    lds_attr_max((const void*)k_leg_consensus<16>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_consensus<32>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_consensus<48>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_consensus<64>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_consensus<80>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_compose<16>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_compose<32>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_compose<48>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_compose<64>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_compose<80>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_tree_down<16>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_tree_down<32>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_tree_down<48>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_tree_down<64>, s->lx.total_bytes);
    lds_attr_max((const void*)k_leg_tree_down<80>, s->lx.total_bytes);
  }
  ensure_leg_capacity(s, s->eff_legs());
  HIP_OK(hipStreamSynchronize(s->stream));
}

// per-instance parameter tables: a slot that receives a (shared) stage table is reset to it for every instance
__global__ void __launch_bounds__(256) k_bcast_params(double* inst, const double* shared, int slot, int nslots, int stride) {
  const int b = blockIdx.y;
  const double* src = shared + (size_t)slot * stride;
  double* dst = inst + ((size_t)b * nslots + slot) * stride;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < stride; i += gridDim.x * 256) dst[i] = src[i];
}
static void bcast_slot_to_instances(mpc_solver* s, int slot) {
  if (!s->d_inst_params) return;
  const Layout& L = s->L;
  hipLaunchKernelGGL(k_bcast_params, dim3(4, L.B), dim3(256), 0, s->stream, s->d_inst_params, s->d_stage_params, slot, L.N + 1, L.max_stage_doubles);
  for (int b = 0; b < L.B; ++b)
    std::memcpy(s->h_inst_params.data() + ((size_t)b * (L.N + 1) + slot) * L.max_stage_doubles, s->h_params.data() + (size_t)slot * L.max_stage_doubles,
                (size_t)L.max_stage_doubles * sizeof(double));
  if (!s->walk_poisoned.empty()) s->walk_poisoned[slot] = 0;  // (the mirror of this slot holds real values again)
}

static void upload_stage(mpc_solver* s, int slot, const int32_t* desc, int n_desc, const double* params, int n_params) {
  const Layout& L = s->L;
  if (n_desc > L.max_stage_ints || n_params > L.max_stage_doubles) throw std::runtime_error("stage table exceeds the capacity given at mpc_create");
  if (n_desc < MPC_STAGE_HEADER_WORDS || n_desc < MPC_STAGE_HEADER_WORDS + MPC_TERM_WORDS * desc[5]) throw std::runtime_error("stage descriptor truncated");
  if (desc[6] > L.c) throw std::runtime_error("stage has more constraint rows than nc_max");
  int nc = 0, nse3 = 0;
  if (s->dims.space == MPC_SPACE_MULTIBODY && desc[5] > 24) throw std::runtime_error("more than 24 terms in one stage (whole-body kernel's LDS term table)");
  for (int t = 0; t < desc[5]; ++t) {
    const int32_t* w = desc + MPC_STAGE_HEADER_WORDS + MPC_TERM_WORDS * t;
    if (w[1] != MPC_ROLE_COST) nc += w[2];
    if (w[2] > 24 && s->dims.space == MPC_SPACE_VECTOR) throw std::runtime_error("residual dimension too large for the vector-space kernel");
    if (s->dims.space == MPC_SPACE_MULTIBODY) {
      // whole-body kernel limits: LDS staging rows of a constraint term, slots of the SE(3) table
      const bool selector = (w[0] == MPC_TERM_STATE_ERROR && w[3] >= 6) || w[0] == MPC_TERM_CONTROL_ERROR;
      if (w[1] != MPC_ROLE_COST && !selector && w[2] > MB_STAGE_CONSTRAINT_ROWS) throw std::runtime_error("constraint term too large for the whole-body kernel's LDS staging rows");
      const bool diag_sel = (w[0] == MPC_TERM_STATE_ERROR || w[0] == MPC_TERM_CONTROL_ERROR) && (w[7] & MPC_TERM_FLAG_DIAG_WEIGHT);
      if (w[1] == MPC_ROLE_COST && !diag_sel && w[2] > 24) throw std::runtime_error("cost term too large for the whole-body kernel's LDS staging rows (dense weights: dim <= 24)");
      if (w[0] == MPC_TERM_FRAME_PLACEMENT || (w[0] == MPC_TERM_STATE_ERROR && w[3] < 6)) ++nse3;
    }
  }
  if (nse3 > MB_SE3_SLOTS) throw std::runtime_error("too many SE(3)-valued terms in one stage for the whole-body kernel");
  if (nc != desc[6]) throw std::runtime_error("stage descriptor: constraint row count mismatch");
  if (desc[0] == MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER) s->contact_dyn = true;
  // receding horizon: the stage that enters the ring usually equals the one that left the slot (same contact phase,
  // same references) — nothing to move then
  int32_t* hd = s->h_desc.data() + (size_t)slot * L.max_stage_ints;
  double* hp = s->h_params.data() + (size_t)slot * L.max_stage_doubles;
  if (s->h_len[2 * slot] == n_desc && s->h_len[2 * slot + 1] == n_params && std::memcmp(hd, desc, n_desc * sizeof(int32_t)) == 0 &&
      (n_params == 0 || std::memcmp(hp, params, n_params * sizeof(double)) == 0)) {
    bcast_slot_to_instances(s, slot);
    return;
  }
  std::memcpy(hd, desc, n_desc * sizeof(int32_t));
  if (n_params > 0) std::memcpy(hp, params, n_params * sizeof(double));
  s->h_len[2 * slot] = n_desc; s->h_len[2 * slot + 1] = n_params;
  // through a pinned ring slot (the caller's buffers are not retained past the call): waits only if the copy that used this slot
  // STAGE_RING uploads ago has not run yet
  const int rs = s->stage_next;
  s->stage_next = (s->stage_next + 1) % mpc_solver::STAGE_RING;
  const size_t ints_bytes = (size_t)L.max_stage_ints * sizeof(int32_t), dbl_bytes = (size_t)L.max_stage_doubles * sizeof(double);
  if (!s->stage_pin[rs]) {
    HIP_OK(hipHostMalloc((void**)&s->stage_pin[rs], ints_bytes + dbl_bytes + 16, hipHostMallocDefault));
    HIP_OK(hipEventCreateWithFlags(&s->stage_ev[rs], hipEventDisableTiming));
  } else {
    HIP_OK(hipEventSynchronize(s->stage_ev[rs]));
  }
  char* pin = s->stage_pin[rs];
  char* pin_d = pin + ((ints_bytes + 15) & ~(size_t)15);
  std::memcpy(pin, desc, n_desc * sizeof(int32_t));
  if (n_params > 0) std::memcpy(pin_d, params, n_params * sizeof(double));
  HIP_OK(hipMemcpyAsync(s->d_stage_desc + (size_t)slot * L.max_stage_ints, pin, n_desc * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
  if (n_params > 0)
    HIP_OK(hipMemcpyAsync(s->d_stage_params + (size_t)slot * L.max_stage_doubles, pin_d, n_params * sizeof(double), hipMemcpyHostToDevice, s->stream));
  HIP_OK(hipEventRecord(s->stage_ev[rs], s->stream));
  bcast_slot_to_instances(s, slot);
}

// tick reuse: whatever changes the problem or the iterate behind the solver's back invalidates the kept records
static void spec_clear(mpc_solver* s) {
  if (s->d_spec) HIP_OK(hipMemsetAsync(s->d_spec, 0, s->L.B * sizeof(int), s->stream));
  s->spec_rec_valid = s->spec_next_pending = s->spec_next_now = false;
  s->cycles_since_run = 0;
  std::fill(s->slot_dirty.begin(), s->slot_dirty.end(), (uint8_t)0);
  s->dirty_all = false;
}

// Start of a pass that may reuse records (mpc_run_shifted*): the slots updated since the last pass become the knot mask of this
// pass; the records the pass leaves behind are consistent with the current parameters again.
static void begin_reuse_pass(mpc_solver* s) {
  const int N = s->L.N;
  for (int w = 0; w < MPC_DIRTY_WORDS; ++w) s->dirty_now[w] = 0ull;
  bool any = false;
  for (int sl = 0; sl <= N; ++sl) if (s->slot_dirty[sl]) {
    any = true;
    const int k = sl == N ? N : (sl - s->head + N) % N;
    if (k < 64 * MPC_DIRTY_WORDS) s->dirty_now[k >> 6] |= 1ull << (k & 63);
  }
  if (s->dirty_all || (any && N + 1 > 64 * MPC_DIRTY_WORDS)) s->reuse_this_pass = false;  // no per-knot mask for this horizon: evaluate everything
  // Most knots updated (a replanning tick of the walk: every foot reference changes): the next tick will most likely find its records
  // stale again, so this pass evaluates its full-step candidate value-only instead of with derivatives into the records (the
  // derivative work would be thrown away) and reuses nothing — same results, the plain path for one tick.
  int ndirty = 0;
  for (int w = 0; w < MPC_DIRTY_WORDS; ++w) ndirty += __builtin_popcountll(s->dirty_now[w]);
  s->spec_skip_pass = 2 * ndirty > N;
  std::fill(s->slot_dirty.begin(), s->slot_dirty.end(), (uint8_t)0);
  s->dirty_all = false;
}

// One host-to-device copy + a scatter kernel for a set of parameter patches: [count | (dst offset, src offset, length) x count | values].
// (inst != nullptr: a patch of the shared tables also goes into every instance's copy — blockIdx.y = instance, inst_stride doubles apart)
__global__ void __launch_bounds__(64) k_scatter_params(double* dst, const char* packed, double* inst, size_t inst_stride) {
  const int* hdr = (const int*)packed;
  const int count = hdr[0];
  const int* tri = hdr + 4 + 3 * blockIdx.x;
  const double* vals = (const double*)(packed + (((size_t)(4 + 3 * count) * sizeof(int) + 15) & ~(size_t)15));
  double* d = (blockIdx.y == 0) ? dst : inst + (size_t)(blockIdx.y - 1) * inst_stride;
  for (int i = threadIdx.x; i < tri[2]; i += 64) d[(size_t)tri[0] + i] = vals[(size_t)tri[1] + i];
}

struct ParamPatch { int slot, offset, len; const double* vals; int inst = -1; };  // inst >= 0: a patch of that instance's own table

// Patches that differ from the host mirror go to the device (stream-ordered, no host wait) and mark their slot dirty; unchanged
// ones cost a memcmp.  Returns the number of patches that travelled.
static int apply_param_patches(mpc_solver* s, const std::vector<ParamPatch>& in) {
  const Layout& L = s->L;
  std::vector<ParamPatch> ch;
  size_t nval = 0;
  const size_t istride = (size_t)(L.N + 1) * L.max_stage_doubles;
  const bool per_inst = !in.empty() && in[0].inst >= 0;  // (a call carries patches of one kind)
  for (const ParamPatch& p : in) {
    double* hp = (p.inst >= 0 ? s->h_inst_params.data() + (size_t)p.inst * istride : s->h_params.data()) + (size_t)p.slot * L.max_stage_doubles + p.offset;
    if (p.len == 0 || std::memcmp(hp, p.vals, p.len * sizeof(double)) == 0) continue;
    std::memcpy(hp, p.vals, p.len * sizeof(double));
    if (!s->walk_poisoned.empty()) s->walk_poisoned[p.slot] = 0;  // (a host patch put real values into the mirror of this slot: the next device-generated tick poisons it again)
    if (p.inst < 0 && s->d_inst_params)
      for (int b = 0; b < L.B; ++b) std::memcpy(s->h_inst_params.data() + (size_t)b * istride + (size_t)p.slot * L.max_stage_doubles + p.offset, p.vals, p.len * sizeof(double));
    s->slot_dirty[p.slot] = 1;  // (per slot, for every instance: the knot mask of tick reuse is shared)
    ch.push_back(p);
    nval += p.len;
  }
  if (ch.empty()) return 0;
  const size_t hdr_bytes = ((size_t)(4 + 3 * ch.size()) * sizeof(int) + 15) & ~(size_t)15, bytes = hdr_bytes + nval * sizeof(double);
  const int rs = s->patch_next;
  s->patch_next = (s->patch_next + 1) % mpc_solver::PATCH_RING;
  if (s->patch_pin[rs]) HIP_OK(hipEventSynchronize(s->patch_ev[rs]));  // (the copy that used this slot PATCH_RING calls ago)
  else HIP_OK(hipEventCreateWithFlags(&s->patch_ev[rs], hipEventDisableTiming));
  if (s->patch_cap[rs] < bytes) {
    if (s->patch_pin[rs]) HIP_OK(hipHostFree(s->patch_pin[rs]));
    s->patch_cap[rs] = bytes * 2 + 4096;
    HIP_OK(hipHostMalloc((void**)&s->patch_pin[rs], s->patch_cap[rs], hipHostMallocDefault));
  }
  if (s->d_patch_cap < bytes) {  // grows rarely (first ticks): the old buffer may still be read by a queued scatter kernel
    HIP_OK(hipStreamSynchronize(s->stream));
    if (s->d_patch) HIP_OK(hipFree(s->d_patch));
    s->d_patch_cap = bytes * 2 + 4096;
    HIP_OK(hipMalloc((void**)&s->d_patch, s->d_patch_cap));
  }
  char* pin = s->patch_pin[rs];
  int* hdr = (int*)pin;
  hdr[0] = (int)ch.size(); hdr[1] = hdr[2] = hdr[3] = 0;
  double* vals = (double*)(pin + hdr_bytes);
  size_t pos = 0;
  for (size_t i = 0; i < ch.size(); ++i) {
    hdr[4 + 3 * i] = (int)((ch[i].inst >= 0 ? (size_t)ch[i].inst * istride : (size_t)0) + (size_t)ch[i].slot * L.max_stage_doubles + ch[i].offset);
    hdr[4 + 3 * i + 1] = (int)pos;
    hdr[4 + 3 * i + 2] = ch[i].len;
    std::memcpy(vals + pos, ch[i].vals, ch[i].len * sizeof(double));
    pos += ch[i].len;
  }
  HIP_OK(hipMemcpyAsync(s->d_patch, pin, bytes, hipMemcpyHostToDevice, s->stream));
  HIP_OK(hipEventRecord(s->patch_ev[rs], s->stream));
  if (per_inst) hipLaunchKernelGGL(k_scatter_params, dim3((unsigned)ch.size(), 1), dim3(64), 0, s->stream, s->d_inst_params, (const char*)s->d_patch, (double*)nullptr, (size_t)0);
  else hipLaunchKernelGGL(k_scatter_params, dim3((unsigned)ch.size(), s->d_inst_params ? 1 + L.B : 1), dim3(64), 0, s->stream, s->d_stage_params, (const char*)s->d_patch, s->d_inst_params, istride);
  HIP_OK(hipGetLastError());
  return (int)ch.size();
}

// warm-start shift of the iterate (k_shift): from the current pair of buffers into the other one, which becomes the current pair
static void launch_shift(mpc_solver* s) {
  const double *xin = s->d_xs, *uin = s->d_us;
  std::swap(s->d_xs, s->d_xs_alt);
  std::swap(s->d_us, s->d_us_alt);
  hipLaunchKernelGGL(k_shift, dim3(s->L.N + 1, s->L.B), dim3(64), 0, s->stream, s->args(), xin, uin, s->perfect_feedback ? 1 : 0);
}

static int slot_of(const mpc_solver* s, int k) { return k < s->L.N ? (s->head + k) % s->L.N : s->L.N; }

// ---- kernel sequences -----------------------------------------------------------------------------
static void launch_eval(mpc_solver* s, bool trial, int cand0 = 0, int ncand = 1, bool with_derivs = false) {
  const Layout& L = s->L;
  SolverArgs a = s->args();
  if (ncand <= 0) return;
  if (L.space == MPC_SPACE_VECTOR) {
    if (!trial) hipLaunchKernelGGL(k_eval_vector<0>, dim3(a.only_knot >= 0 ? 1 : L.N + 1, L.B, 1), dim3(64), 0, s->stream, a, s->L, s->d_knots, 0);
    else hipLaunchKernelGGL(k_eval_vector<1>, dim3(L.N + 1, L.B, ncand), dim3(64), 0, s->stream, a, s->LT, s->d_tknots, cand0);
  } else {
    launch_eval_multibody(s->stream, a, s->LT, (trial && !with_derivs) ? s->d_tknots : s->d_knots, s->d_mbwork, s->mb_work_stride, trial, cand0, ncand, 0, 0.0, with_derivs, nullptr, s->contact_dyn);
  }
  HIP_OK(hipGetLastError());
}

// one pass of the inner loop for every instance that is not done
static void launch_pass(mpc_solver* s) {
  const Layout& L = s->L;
  if (s->pass_in_run > 0 && s->tick_reuse && s->opt.max_iters <= 4) {
    // a further iteration of the same tick: instances whose full step was accepted find their records already written, for the same knots
    s->reuse_this_pass = true; s->reuse_same_now = true;
    for (int w = 0; w < MPC_DIRTY_WORDS; ++w) s->dirty_now[w] = 0ull;
  }
  SolverArgs a = s->args();
  if (a.spec_next) hipLaunchKernelGGL(k_copy_spec, dim3(32, L.B), dim3(256), 0, s->stream, a);  // the appended knot: evaluated speculatively by the previous tick
  s->timed(0, "k_eval_stage", [&] { launch_eval(s, false); });
  if (a.reuse_on) hipLaunchKernelGGL(k_reproject, dim3(L.N + 1, L.B), dim3(256), 0, s->stream, a);  // records kept from the last tick: fresh projections
  s->reuse_this_pass = false; s->reuse_same_now = false;
  s->spec_next_now = false;
  s->timed(1, "k_lagrangian", [&] { hipLaunchKernelGGL(k_lagrangian, dim3(L.N + 1, L.B), dim3(64), 0, s->stream, a); });
  s->timed(2, "k_decide", [&] { hipLaunchKernelGGL(k_decide, dim3(L.B), dim3(128), 0, s->stream, a); });
  // forward sweep: closed-loop transitions Phi / phi for all knots in parallel, then one mat-vec per knot (closed_loop.h);
  // the three-mat-vec sweep remains for dimensions whose operands do not fit the LDS of k_closed_loop
  // The knot-parallel kernel costs (N x B) workgroups: it pays when the GPU is mostly idle during the sweep (small
  // ensembles, latency-bound: 0.95 -> 0.53 ms at batch 1), not when the ensemble already fills it (B = 64: 0.97 -> 1.3 ms)
  // (options.forward_mode overrides: a handle cannot see the other handles that share its GPU — with 4 shards of 16 instances the
  // sweep gives 4714 solves/s against 4551)
  a.abdz = nullptr;
  const int J = a.nlegs;
  // legs without a value-function guess at the cuts (first pass of the handle): two sweeps, the second from the Hessians the first
  // one found (legs.h) ; MPC_LEGS_PLAIN=1: always one sweep from zero (intermediates comparable with the oracle's)
  const bool plain = s->env_plain;
  const bool tree = s->use_tree();
  if (tree && s->tree.J != J) s->tree = make_tree_desc(J);
  // (the tree refreshes the guess of a cut from the node that starts there: every cut has seen the true terminal cost after one sweep per level)
  int sweeps = (J > 1 && !plain && !s->leg_guess_valid) ? (tree ? s->tree.nlev + 1 : 2) : 1;
  if (tree && !plain && s->env_tree_sweeps > sweeps) sweeps = s->env_tree_sweeps;  // developer knob: sweeps per pass
  for (int sweep = 0; sweep < sweeps; ++sweep) {
  s->leg_guess_now = (J > 1 && !plain && (s->leg_guess_valid || sweep > 0)) ? 1 : 0;
  a = s->args();
  a.abdz = nullptr;
  s->timed(3, "k_riccati_backward", [&] {
    if (J > 1) {
      // parallel-in-time: workgroup (instance, leg), the last leg first
      if (s->ric.np == 16 && s->ric.mp == 16 && L.c <= RIC_SMALL_THREADS)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_SMALL_THREADS, 16, false, true>), dim3(L.B * J), dim3(RIC_SMALL_THREADS), s->ric.total_bytes, s->stream, a, s->ric);
#define X(ID, FN, FM, GF, ST, NPV, MPV) else if (s->ric_fixed == ID) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_THREADS, 80, true, true, FN, FM, GF, ST>), dim3(L.B * J), dim3(RIC_THREADS), s->ric.total_bytes, s->stream, a, s->ric);
      MPC_FIXED_MODELS(X)
#undef X
      else if (s->ric.sq) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_THREADS, 80, true, true>), dim3(L.B * J), dim3(RIC_THREADS), s->ric.total_bytes, s->stream, a, s->ric);
      else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_THREADS, 80, false, true>), dim3(L.B * J), dim3(RIC_THREADS), s->ric.total_bytes, s->stream, a, s->ric);
    }
    else if (s->use_mfma_riccati && s->ric.np == 16 && s->ric.mp == 16 && L.c <= RIC_SMALL_THREADS)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_SMALL_THREADS, 16>), dim3(L.B), dim3(RIC_SMALL_THREADS), s->ric.total_bytes, s->stream, a, s->ric);  // small problems: one wavefront
    else if (s->use_mfma_riccati && s->ric.sq && s->ric.np <= 80) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_THREADS, 80, true>), dim3(L.B), dim3(RIC_THREADS), s->ric.total_bytes, s->stream, a, s->ric);
    else if (s->use_mfma_riccati && s->ric.sq) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_THREADS, 96, true>), dim3(L.B), dim3(RIC_THREADS), s->ric.total_bytes, s->stream, a, s->ric);
    else if (s->use_mfma_riccati && s->ric.np <= 80) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_THREADS, 80>), dim3(L.B), dim3(RIC_THREADS), s->ric.total_bytes, s->stream, a, s->ric);  // fewer tiles per wavefront: lower register pressure
    else if (s->use_mfma_riccati) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_riccati_mfma<RIC_THREADS, 96>), dim3(L.B), dim3(RIC_THREADS), s->ric.total_bytes, s->stream, a, s->ric);
    else hipLaunchKernelGGL(k_riccati_backward, dim3(L.B), dim3(256), s->riccati_lds(), s->stream, a);
  });
  if (J > 1) {
    // legs: per-knot closed-loop transitions and parametric terms, condensation of every leg, consensus over the cuts, final
    // affine terms, then the forward sweeps of the legs side by side
    s->timed(12, "k_closed_loop", [&] {
      // knots per workgroup: 1 (measured: with the operands of the next knot prefetched, chunks of 4 .. 25 knots take the same time per
      // knot — the kernel is bound by its own instruction stream, not by the loads; MPC_LEG_KNOT_CHUNK for experiments)
      static const int chunk_env = getenv("MPC_LEG_KNOT_CHUNK") ? atoi(getenv("MPC_LEG_KNOT_CHUNK")) : 0;
      const int chunk = chunk_env > 0 ? chunk_env : 1;
      const dim3 grid((L.N + chunk - 1) / chunk, L.B);
      if (false) {}
#define X(ID, FN, FM, GF, ST, NPV, MPV) else if (s->ric_fixed == ID) hipLaunchKernelGGL((k_leg_knot<MPV, FN, FM>), grid, dim3(LK_THREADS), s->lk.total_bytes, s->stream, a, s->lk, chunk);
      MPC_FIXED_MODELS(X)
#undef X
      else if (s->lk.mp <= 16) hipLaunchKernelGGL(k_leg_knot<16>, grid, dim3(LK_THREADS), s->lk.total_bytes, s->stream, a, s->lk, chunk);
      else if (s->lk.mp <= 32) hipLaunchKernelGGL(k_leg_knot<32>, grid, dim3(LK_THREADS), s->lk.total_bytes, s->stream, a, s->lk, chunk);
      else hipLaunchKernelGGL(k_leg_knot<48>, grid, dim3(LK_THREADS), s->lk.total_bytes, s->stream, a, s->lk, chunk);
    });
    s->timed(13, "k_leg_condense", [&] {
      if (false) {}
#define X(ID, FN, FM, GF, ST, NPV, MPV) else if (s->ric_fixed == ID) hipLaunchKernelGGL(k_leg_condense<FN>, dim3(J - 1, L.B), dim3(LK_THREADS), s->lc.total_bytes, s->stream, a, s->lc);
      MPC_FIXED_MODELS(X)
#undef X
      else hipLaunchKernelGGL(k_leg_condense<0>, dim3(J - 1, L.B), dim3(LK_THREADS), s->lc.total_bytes, s->stream, a, s->lc);
    });
    if (tree) s->timed(14, "k_leg_consensus", [&] {
      const TreeDesc& T = s->tree;
#define MPC_TREE_LAUNCH(NPV) do { \
        for (int lev = 0; lev < T.nlev; ++lev) hipLaunchKernelGGL(k_leg_compose<NPV>, dim3(T.lev_cnt[lev] + 1, L.B, 2), dim3(LCMP_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev); } while (0)
      if (false) {
#define X(ID, FN, FM, GF, ST, NPV, MPV) } else if (s->ric_fixed == ID) { for (int lev = 0; lev < T.nlev; ++lev) hipLaunchKernelGGL((k_leg_compose<NPV, FN, FM>), dim3(T.lev_cnt[lev] + 1, L.B, 2), dim3(LCMP_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev);
      MPC_FIXED_MODELS(X)
#undef X
      } else switch (s->lx.np) {
        case 16: MPC_TREE_LAUNCH(16); break;
        case 32: MPC_TREE_LAUNCH(32); break;
        case 48: MPC_TREE_LAUNCH(48); break;
        case 64: MPC_TREE_LAUNCH(64); break;
        default: MPC_TREE_LAUNCH(80); break;
      }
#undef MPC_TREE_LAUNCH
    });
    if (tree) s->timed(16, "k_leg_tree_down", [&] {
      const TreeDesc& T = s->tree;
      if (false) {
#define X(ID, FN, FM, GF, ST, NPV, MPV) } else if (s->ric_fixed == ID) { for (int lev = T.nlev - 1; lev >= 0; --lev) hipLaunchKernelGGL((k_leg_tree_down<NPV, FN, FM>), dim3(T.lev_cnt[lev] + (lev == T.nlev - 1 ? 1 : 0), L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev);
      MPC_FIXED_MODELS(X)
#undef X
      } else switch (s->lx.np) {
        case 16: for (int lev = T.nlev - 1; lev >= 0; --lev) hipLaunchKernelGGL(k_leg_tree_down<16>, dim3(T.lev_cnt[lev] + (lev == T.nlev - 1 ? 1 : 0), L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev); break;
        case 32: for (int lev = T.nlev - 1; lev >= 0; --lev) hipLaunchKernelGGL(k_leg_tree_down<32>, dim3(T.lev_cnt[lev] + (lev == T.nlev - 1 ? 1 : 0), L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev); break;
        case 48: for (int lev = T.nlev - 1; lev >= 0; --lev) hipLaunchKernelGGL(k_leg_tree_down<48>, dim3(T.lev_cnt[lev] + (lev == T.nlev - 1 ? 1 : 0), L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev); break;
        case 64: for (int lev = T.nlev - 1; lev >= 0; --lev) hipLaunchKernelGGL(k_leg_tree_down<64>, dim3(T.lev_cnt[lev] + (lev == T.nlev - 1 ? 1 : 0), L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev); break;
        default: for (int lev = T.nlev - 1; lev >= 0; --lev) hipLaunchKernelGGL(k_leg_tree_down<80>, dim3(T.lev_cnt[lev] + (lev == T.nlev - 1 ? 1 : 0), L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx, T, lev); break;
      }
    });
    else s->timed(14, "k_leg_consensus", [&] {
      switch (s->lx.np) {
        case 16: hipLaunchKernelGGL(k_leg_consensus<16>, dim3(L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx); break;
        case 32: hipLaunchKernelGGL(k_leg_consensus<32>, dim3(L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx); break;
        case 48: hipLaunchKernelGGL(k_leg_consensus<48>, dim3(L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx); break;
        case 64: hipLaunchKernelGGL(k_leg_consensus<64>, dim3(L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx); break;
        default: hipLaunchKernelGGL(k_leg_consensus<80>, dim3(L.B), dim3(LK_THREADS), s->lx.total_bytes, s->stream, a, s->lx); break;
      }
    });
  }
  }
  if (J > 1) {
    s->leg_guess_valid = true;
    s->timed(15, "k_leg_apply", [&] { hipLaunchKernelGGL(k_leg_apply, dim3(L.N, L.B), dim3(256), 0, s->stream, a); });
    s->timed(4, "k_forward", [&] {
      if (L.m <= 32) hipLaunchKernelGGL((k_forward_phi<4, 10>), dim3(L.B * J), dim3(512), 2 * L.n * sizeof(double), s->stream, a);
      else hipLaunchKernelGGL((k_forward_phi<6, 10>), dim3(L.B * J), dim3(512), 2 * L.n * sizeof(double), s->stream, a);
    });
  } else {
  const bool fw_phi = s->cl.total_bytes <= 160 * 1024 && L.n <= 80 && L.m <= 32 && (s->opt.forward_mode == 0 ? L.B * L.N <= 2048 : s->opt.forward_mode == 2);
  if (fw_phi) {
    s->timed(12, "k_closed_loop", [&] { hipLaunchKernelGGL(k_closed_loop, dim3(L.N, L.B), dim3(CL_THREADS), s->cl.total_bytes, s->stream, a, s->cl); });
    s->timed(4, "k_forward", [&] { hipLaunchKernelGGL((k_forward_phi<4, 10>), dim3(L.B), dim3(512), 2 * L.n * sizeof(double), s->stream, a); });
  } else {
    s->timed(4, "k_forward", [&] {
      const size_t fw_lds = (L.nz + 2 * L.n) * sizeof(double);
      const bool fits = L.nz <= 128 && L.n + L.m <= L.nz;
      if (fits && L.n <= 80 && L.m <= 48) a.abdz = s->d_abdz;  // the register-prefetch sweeps leave [A B] [dx; du] of every knot for k_duals
      if (fits && L.n <= 80 && L.m <= 32) hipLaunchKernelGGL((k_forward_prefetch<4, 10>), dim3(L.B), dim3(512), fw_lds, s->stream, a);
      else if (fits && L.n <= 80 && L.m <= 48) hipLaunchKernelGGL((k_forward_prefetch<6, 10>), dim3(L.B), dim3(512), fw_lds, s->stream, a);
      else hipLaunchKernelGGL(k_forward, dim3(L.B), dim3(1024), fw_lds, s->stream, a);
    });
  }
  }
  s->timed(5, "k_duals", [&] { hipLaunchKernelGGL(k_duals, dim3(L.N + 1, L.B), dim3(256), (L.nz + 3 * L.n + 16 + 3 * L.c) * sizeof(double), s->stream, a); });
  // linesearch: evaluate the full step first; the backtracking candidates alpha = 2^-i, i >= 1, are only
  // evaluated for instances whose full step failed the Armijo test (their workgroups exit immediately otherwise)
  // (two profile slots: with derivatives into the knot records — the launch the roofline is quoted on — or values only)
  if (a.spec_on || L.space != MPC_SPACE_MULTIBODY) s->timed(6, "k_eval_stage_trial", [&] { launch_eval(s, true, 0, 1, a.spec_on != 0); });
  else s->timed(17, "k_eval_stage_trial_values", [&] { launch_eval(s, true, 0, 1, false); });
  s->spec_rec_valid = a.spec_on != 0 && a.spec_knot != nullptr;  // (per instance it counts only if the full step is accepted: a.spec[b])
  s->timed(7, "k_linesearch", [&] { hipLaunchKernelGGL(k_linesearch, dim3(L.B), dim3(64), 0, s->stream, a, 1, 0); });
  // backtracking in two launches: alpha = 1/2 and 1/4 for every instance whose full step failed, the remaining candidates only for those
  // that are still undecided (a workgroup walks its candidates one after the other: evaluating all seven at once cost a tick of the
  // kinodynamic ensemble 17 ms whenever one instance backtracked)
  // (a small ensemble is latency-bound: one launch for all candidates, as before)
  const int nc1 = (L.n_alpha - 1 < 2 || (size_t)L.B * (L.N + 1) < 2048) ? L.n_alpha - 1 : 2;
  s->timed(10, "k_eval_stage_backtrack", [&] { launch_eval(s, true, 1, nc1); });
  s->timed(11, "k_linesearch_backtrack", [&] { hipLaunchKernelGGL(k_linesearch, dim3(L.B), dim3(64), 0, s->stream, a, 0, nc1); });
  if (L.n_alpha - 1 - nc1 > 0) {
    s->timed(10, "k_eval_stage_backtrack", [&] { launch_eval(s, true, 1 + nc1, L.n_alpha - 1 - nc1); });
    s->timed(11, "k_linesearch_backtrack", [&] { hipLaunchKernelGGL(k_linesearch, dim3(L.B), dim3(64), 0, s->stream, a, 0, L.n_alpha); });
  }
  s->timed(8, "k_accept", [&] { hipLaunchKernelGGL(k_accept, dim3(L.N + 1, L.B), dim3(64), 0, s->stream, a); });
  s->timed(9, "k_after_step", [&] { hipLaunchKernelGGL(k_after_step, dim3(L.B), dim3(1), 0, s->stream, a); });
  HIP_OK(hipGetLastError());
}

// passes_enqueued: passes already put on the stream by the asynchronous entry point (their completion flag is
// checked first); the loop then continues synchronously until every instance is done.
static void report_status(mpc_solver* s, int B, const InstState* st, mpc_stats* stats) {
  for (int b = 0; b < B; ++b) {
    if (st[b].done >= 2) s->leg_guess_valid = false;  // do not start the legs of the next pass from what a failed sweep left
    if (st[b].done >= 2 && !s->isolate) throw std::runtime_error("Riccati factorisation failed on instance " + std::to_string(b) + " (code " + std::to_string(st[b].done) + ")");
    if (!stats) continue;
    mpc_stats& o = stats[b];
    o.num_iters = st[b].num_iters; o.converged = st[b].done >= 2 ? -st[b].done : st[b].converged; o.al_iters = st[b].al_iters; o.ls_steps = st[b].ls_step;
    o.traj_cost = st[b].cost; o.merit = st[b].phi0; o.prim_infeas = st[b].prim; o.dual_infeas = st[b].dual; o.mu = st[b].mu; o.alpha = st[b].alpha;
  }
}

// mpc_options.refine_appended_knot: R x (evaluate knot N - 1 alone, Newton step on its control), one more evaluation, x_N = phi(...).
// Enqueued after k_begin_run (the stage kernel skips instances that are `done`) and before the first pass, which evaluates knots N - 1
// and N afresh (begin_refine marked their slots dirty before the knot mask of the pass was taken).
static void begin_refine(mpc_solver* s) {
  s->refine_now = s->L.N >= 1 && ((s->opt.refine_appended_knot > 0 && s->appended_changed) || (s->opt.refine_appended_knot < 0 && s->appended_any));
  s->appended_changed = false; s->appended_any = false;
  if (s->refine_now) { s->slot_dirty[slot_of(s, s->L.N - 1)] = 1; s->slot_dirty[s->L.N] = 1; }
}
static void launch_refine(mpc_solver* s) {
  if (!s->refine_now) return;
  s->refine_now = false;
  const Layout& L = s->L;
  const bool reuse_keep = s->reuse_this_pass;
  s->reuse_this_pass = false;  // (plain evaluation of the one knot: no record is taken over)
  s->only_knot = L.N - 1;
  const int R = s->opt.refine_appended_knot < 0 ? -s->opt.refine_appended_knot : s->opt.refine_appended_knot;
  for (int it = 0; it <= R; ++it) {
    s->timed(18, "k_refine_appended_knot", [&] {
      launch_eval(s, false);  // (only_knot: a grid of one knot per instance)
      hipLaunchKernelGGL(k_refine_knot, dim3(L.B), dim3(256), 0, s->stream, s->args(), it < R ? 0 : 1);
    });
  }
  s->only_knot = -1;
  s->reuse_this_pass = reuse_keep;
  HIP_OK(hipGetLastError());
}

static void run_impl(mpc_solver* s, mpc_stats* stats, int passes_enqueued = 0) {
  const Layout& L = s->L;
  SolverArgs a = s->args();
  if (passes_enqueued == 0) { hipLaunchKernelGGL(k_begin_run, dim3(L.B), dim3(64), 0, s->stream, a); launch_refine(s); }
  const int max_passes = s->opt.max_iters + s->opt.max_al_iters + 2;  // (+ 1: the corrector iteration of mpc_options.corrector_prim_tol)
  std::vector<InstState> st(L.B);
  for (int pass = 0; pass < max_passes; ++pass) {
    if (pass >= passes_enqueued) { s->pass_in_run = pass; launch_pass(s); }
    else if (pass + 1 < passes_enqueued) continue;  // only the flag of the last enqueued pass is meaningful
    // one read-back per pass: the per-instance status (the device-side all_done flag says the same as "every done != 0")
    copy_sync(s, st.data(), s->d_inst, L.B * sizeof(InstState), hipMemcpyDeviceToHost);
    int done = 1;
    for (int b = 0; b < L.B; ++b) if (!st[b].done) done = 0;
    // a BCL update changed the penalty of an instance: the cut Hessians of the legs (calP ~ ... + C^T C / mu) were made for the old
    // one — the next pass refreshes them with its extra sweeps instead of starting single-sweep legs from a guess for another mu
    for (int b = 0; b < L.B; ++b) if (!st[b].done && st[b].mu != s->leg_mu[b]) { if (s->leg_mu[b] >= 0.0) s->leg_guess_valid = false; s->leg_mu[b] = st[b].mu; }
    if (s->env_trace >= 0) {  // developer aid: per-pass solver state of one instance
      const int tb = s->env_trace;
      if (tb >= 0 && tb < L.B) {
        const InstState& t = st[tb];
        fprintf(stderr, "[trace b=%d pass %d] it %d al %d mu %.1e phi0 %.10e dphi0 %.3e alpha %.4g ls %d prim %.3e dual %.3e crit %.3e inner_tol %.1e prim_tol %.1e skip %d stalled %d done %d\n",
                tb, pass, t.num_iters, t.al_iters, t.mu, t.phi0, t.dphi0, t.alpha, t.ls_step, t.prim, t.dual, t.crit, t.inner_tol, t.prim_tol, t.skip_step, t.stalled, t.done);
      }
    }
    if (done != 0) break;
  }
  report_status(s, L.B, st.data(), stats);
}

// every entry point runs on the handle's own device (several handles of one process may live on different devices)
#define MPC_TRY(h, ...)                 \
  try {                                 \
    if (h) HIP_OK(hipSetDevice((h)->dims.device)); \
    __VA_ARGS__;                        \
    return 0;                           \
  } catch (const std::exception& e) {   \
    if (h) (h)->err = e.what();         \
    return -1;                          \
  }

extern "C" {

int mpc_abi_version(void) { return MPC_ABI_VERSION; }
const char* mpc_backend_name(void) { return "hip-gfx950"; }

int mpc_create(const mpc_dims* dims, mpc_solver** out) {
  if (!dims || !out) return -2;
  mpc_solver* s = new mpc_solver();
  try {
    create_impl(s, *dims);
    *out = s;
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "mpc_create: %s\n", e.what());
    for (void* p : s->allocs) (void)hipFree(p);
    for (double* p : {s->d_work, s->d_legbuf, s->d_treebuf}) if (p) (void)hipFree(p);
    delete s;
    return -1;
  }
}

void mpc_destroy(mpc_solver* s) {
  if (!s) return;
  (void)hipStreamSynchronize(s->stream);
  for (auto& p : s->prof) for (auto& e : p.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (int i = 0; i < mpc_solver::ASYNC_DEPTH; ++i) { if (s->h_status[i]) (void)hipHostFree(s->h_status[i]); if (s->status_ev[i]) (void)hipEventDestroy(s->status_ev[i]); }
  for (int i = 0; i < mpc_solver::STAGE_RING; ++i) { if (s->stage_pin[i]) (void)hipHostFree(s->stage_pin[i]); if (s->stage_ev[i]) (void)hipEventDestroy(s->stage_ev[i]); }
  for (int i = 0; i < mpc_solver::PATCH_RING; ++i) { if (s->patch_pin[i]) (void)hipHostFree(s->patch_pin[i]); if (s->patch_ev[i]) (void)hipEventDestroy(s->patch_ev[i]); }
  for (int i = 0; i < mpc_solver::ASYNC_DEPTH; ++i) if (s->h_xnext[i]) (void)hipHostFree(s->h_xnext[i]);
  if (s->d_patch) (void)hipFree(s->d_patch);
  for (void* p : s->allocs) (void)hipFree(p);
  for (double* p : {s->d_work, s->d_legbuf, s->d_treebuf}) if (p) (void)hipFree(p);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

const char* mpc_last_error(mpc_solver* s) { return s ? s->err.c_str() : "null handle"; }

int mpc_set_options(mpc_solver* s, const mpc_options* opt) {
  MPC_TRY(s, {
    spec_clear(s);
    if (!opt->rollout_linear || !opt->force_initial_condition) throw std::runtime_error("only ROLLOUT_LINEAR with force_initial_condition is implemented");
    const int legs_before = s->eff_legs();
    s->opt = *opt;
    if (s->opt.ls_max_steps > s->L.n_alpha) s->opt.ls_max_steps = s->L.n_alpha;
    if (s->eff_legs() != legs_before) s->leg_guess_valid = false;  // the cuts moved: the kept Hessians belong to other knots
    ensure_leg_capacity(s, s->eff_legs());
  })
}

int mpc_set_model(mpc_solver* s, const int32_t* itab, int32_t n_i, const double* dtab, int32_t n_d) {
  MPC_TRY(s, {
    if (n_i < MPC_MODEL_HEADER_WORDS) throw std::runtime_error("model table too short");
    const int nj = itab[0], nf = itab[3], ncn = itab[4];
    if (n_i < MPC_MODEL_HEADER_WORDS + MPC_MODEL_JOINT_WORDS * nj + nf + ncn ||
        n_d < MPC_MODEL_HEADER_DOUBLES + MPC_MODEL_JOINT_DOUBLES * nj + MPC_MODEL_FRAME_DOUBLES * nf + MPC_MODEL_CONTACT_DOUBLES * ncn)
      throw std::runtime_error("model table size mismatch");
    HIP_OK(hipStreamSynchronize(s->stream));
    // device copy of the int table = the caller's table followed by the 64-bit tree masks the whole-body kernel walks
    // (ancestors of a body, bodies of its subtree, dofs on its root path), two int32 words each
    const int nvm = itab[2];
    std::vector<int32_t> ext(itab, itab + n_i);
    if (s->dims.space == MPC_SPACE_MULTIBODY) {
      const int32_t* mj = itab + MPC_MODEL_HEADER_WORDS;
      std::vector<unsigned long long> anc(nj, 0ull), sub(nj, 0ull), dm(nj, 0ull), below(nj, 0ull);
      std::vector<int> dof_body(nvm, 0);
      for (int i = 0; i < nj; ++i) {
        for (int j = i; j >= 0; j = mj[4 * j]) anc[i] |= 1ull << j;
        const int ndof = (mj[4 * i + 1] == MPC_JOINT_FREEFLYER) ? 6 : 1;
        for (int d = 0; d < ndof; ++d) dof_body[mj[4 * i + 3] + d] = i;
      }
      for (int i = 0; i < nj; ++i) {
        for (int j = i; j < nj; ++j) if ((anc[j] >> i) & 1ull) sub[i] |= 1ull << j;
        for (int kd = 0; kd < nvm; ++kd) if ((anc[i] >> dof_body[kd]) & 1ull) dm[i] |= 1ull << kd;
        for (int j = i + 1; j < nj; ++j)  // dofs of the joints strictly inside the subtree of i
          if ((sub[i] >> j) & 1ull) below[i] |= ((mj[4 * j + 1] == MPC_JOINT_FREEFLYER) ? 63ull : 1ull) << mj[4 * j + 3];
      }
      if (ext.size() & 1) ext.push_back(0);  // 8-byte alignment of the mask block
      s->L.model_mask_off = (int)ext.size();
      for (const auto* v : {&anc, &sub, &dm, &below})
        for (int i = 0; i < nj; ++i) { ext.push_back((int32_t)((*v)[i] & 0xffffffffull)); ext.push_back((int32_t)((*v)[i] >> 32)); }
    }
    // a model may be set again (a new contact frame lowered later): the old tables are released, not leaked until destroy
    for (void* old : {(void*)s->d_model_i, (void*)s->d_model_d})
      if (old) {
        for (auto it = s->allocs.begin(); it != s->allocs.end(); ++it) if (*it == old) { s->allocs.erase(it); break; }
        HIP_OK(hipFree(old));
      }
    s->d_model_i = s->alloc<int32_t>(ext.size());
    s->d_model_d = s->alloc<double>(n_d);
    spec_clear(s);  // records kept for tick reuse were evaluated on the old model
    s->reuse_this_pass = false;
    copy_sync(s, s->d_model_i, ext.data(), ext.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    copy_sync(s, s->d_model_d, dtab, n_d * sizeof(double), hipMemcpyHostToDevice);
    s->h_model_i.assign(itab, itab + n_i);
    s->L.nj = nj; s->LT.nj = nj; s->LT.model_mask_off = s->L.model_mask_off;
    if (s->dims.space == MPC_SPACE_MULTIBODY) {
      if (itab[2] * 2 != s->L.n || itab[1] + itab[2] != s->L.nx) throw std::runtime_error("model dimensions do not match the state space");
      check_multibody_model(itab, n_i);
      if (!s->d_mbwork) {
        s->mb_work_stride = multibody_work_doubles(s->L);
        s->d_mbwork = s->alloc<double>((size_t)s->L.B * (s->L.N + 2) * s->mb_work_stride + 8);  // one slot more per instance: the speculative knot
        s->d_spec_knot = s->alloc<double>((size_t)s->L.B * s->L.knot_stride);
      }
    }
    s->have_model = true;
    HIP_OK(hipStreamSynchronize(s->stream));
  })
}

int mpc_set_stage(mpc_solver* s, int32_t k, const int32_t* desc, int32_t n_desc, const double* params, int32_t n_params) {
  MPC_TRY(s, {
    spec_clear(s);
    s->leg_guess_valid = false;
    if (k < 0 || k > s->L.N) throw std::runtime_error("stage index out of range");
    upload_stage(s, slot_of(s, k), desc, n_desc, params, n_params);
  })
}

int mpc_update_stage_params_batch(mpc_solver* s, int32_t count, const int32_t* ks, const int32_t* offsets, const int32_t* lens,
                                  const double* vals) {
  MPC_TRY(s, {
    std::vector<ParamPatch> patches;
    patches.reserve(count);
    size_t pos = 0;
    for (int i = 0; i < count; ++i) {
      if (ks[i] < 0 || ks[i] > s->L.N) throw std::runtime_error("stage index out of range");
      if (offsets[i] < 0 || lens[i] < 0 || offsets[i] + lens[i] > s->L.max_stage_doubles) throw std::runtime_error("parameter update out of range");
      patches.push_back({slot_of(s, ks[i]), offsets[i], lens[i], vals + pos});
      pos += lens[i];
    }
    apply_param_patches(s, patches);  // (the caller's buffers are copied into pinned memory: not retained past the call)
  })
}

int mpc_update_stage_params(mpc_solver* s, int32_t k, int32_t offset, const double* vals, int32_t n) {
  MPC_TRY(s, {
    if (k < 0 || k > s->L.N) throw std::runtime_error("stage index out of range");
    if (offset < 0 || n < 0 || offset + n > s->L.max_stage_doubles) throw std::runtime_error("parameter update out of range");
    apply_param_patches(s, {ParamPatch{slot_of(s, k), offset, n, vals}});
  })
}

int mpc_enable_instance_params(mpc_solver* s) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    if (s->d_inst_params) return 0;
    const size_t istride = (size_t)(L.N + 1) * L.max_stage_doubles;
    s->d_inst_params = s->alloc<double>((size_t)L.B * istride);
    s->h_inst_params.assign((size_t)L.B * istride, 0.0);
    for (int sl = 0; sl <= L.N; ++sl) bcast_slot_to_instances(s, sl);
    spec_clear(s);
    s->dirty_all = true;
  })
}

int mpc_update_instance_params_batch(mpc_solver* s, int32_t count, const int32_t* insts, const int32_t* ks, const int32_t* offsets, const int32_t* lens, const double* vals) {
  MPC_TRY(s, {
    if (!s->d_inst_params) throw std::runtime_error("update_instance_params: mpc_enable_instance_params first");
    std::vector<ParamPatch> patches;
    patches.reserve(count);
    size_t pos = 0;
    for (int i = 0; i < count; ++i) {
      if (insts[i] < 0 || insts[i] >= s->L.B) throw std::runtime_error("instance index out of range");
      if (ks[i] < 0 || ks[i] > s->L.N) throw std::runtime_error("stage index out of range");
      if (offsets[i] < 0 || lens[i] < 0 || offsets[i] + lens[i] > s->L.max_stage_doubles) throw std::runtime_error("parameter update out of range");
      ParamPatch p{slot_of(s, ks[i]), offsets[i], lens[i], vals + pos};
      p.inst = insts[i];
      patches.push_back(p);
      pos += lens[i];
    }
    apply_param_patches(s, patches);
  })
}

int mpc_walk_init(mpc_solver* s, const mpc_walk_config* cfg) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    if (!cfg) throw std::runtime_error("walk_init: null configuration");
    if (!s->d_inst_params) throw std::runtime_error("walk_init: mpc_enable_instance_params first");
    if (!s->d_model_i || L.space != MPC_SPACE_MULTIBODY) throw std::runtime_error("walk_init: a whole-body model is needed (mpc_set_model)");
    const int nf_ = s->h_model_i.size() > 3 ? s->h_model_i[3] : 0;
    if (cfg->frame_lf < 0 || cfg->frame_lf >= nf_ || cfg->frame_rf < 0 || cfg->frame_rf >= nf_) throw std::runtime_error("walk_init: frame index out of range");
    for (int off : {cfg->off_lf, cfg->off_rf, cfg->toff_lf, cfg->toff_rf}) if (off >= 0 && off + 12 > L.max_stage_doubles) throw std::runtime_error("walk_init: reference offset out of range");
    if ((cfg->toff_com >= 0 && cfg->toff_com + 3 > L.max_stage_doubles) || (cfg->off_xref_z >= L.max_stage_doubles)) throw std::runtime_error("walk_init: offset out of range");
    s->walk = *cfg;
    if (!s->d_walk_state) s->d_walk_state = s->alloc<double>((size_t)L.B * 48);
    std::vector<double> st((size_t)L.B * 48);
    for (int b = 0; b < L.B; ++b) {
      double* p = st.data() + (size_t)b * 48;
      std::memcpy(p, cfg->lf0, 96); std::memcpy(p + 12, cfg->lf0, 96); std::memcpy(p + 24, cfg->rf0, 96); std::memcpy(p + 36, cfg->rf0, 96);
    }
    copy_sync(s, s->d_walk_state, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice);
    s->walk_on = true;
  })
}

int mpc_walk_update(mpc_solver* s, int32_t takeoff_RF, int32_t takeoff_LF, int32_t land_RF, int32_t land_LF, const double* forward) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    if (!s->walk_on) throw std::runtime_error("walk_update: mpc_walk_init first");
    mpc_walk_config& c = s->walk;
    if (forward) { std::memcpy(c.t_left, forward, 24); std::memcpy(c.t_right, forward + 3, 24); c.swing_apex = forward[6]; }
    const bool replanning = land_LF < 0 || land_RF < 0 || (takeoff_RF >= 0 && takeoff_RF < c.T_ds) || (takeoff_LF >= 0 && takeoff_LF < c.T_ds);
    hipLaunchKernelGGL(k_walk_refs, dim3(L.B), dim3(128), 0, s->stream, s->args(), c, s->d_walk_state, (int)takeoff_RF, (int)takeoff_LF, (int)land_RF, (int)land_LF, replanning ? 1 : 0, (replanning || s->walk_force_all) ? 1 : 0);
    HIP_OK(hipGetLastError());
    // tick reuse: the records of the knots whose references were rewritten are stale (on a replanning tick: all of them)
    if (replanning || s->walk_force_all) for (int k = 0; k < L.N; ++k) s->slot_dirty[slot_of(s, k)] = 1;
    else s->slot_dirty[slot_of(s, L.N - 1)] = 1;
    s->slot_dirty[L.N] = 1;
    // The kernel wrote the references straight into d_inst_params: the host mirror that mpc_update_instance_params_batch compares its patches with no longer
    // says what the device holds.  Poison the written ranges (NaN never memcmp-equals a caller's values), so that a host patch of the same offsets after a
    // device-generated tick always travels — switching an ensemble from the device generator to the host one, or mixing the two APIs, stays correct.
    // (Once per slot until a host patch or a stage upload rewrites its mirror: nothing per tick in steady state.)
    {
      if (s->walk_poisoned.size() != (size_t)(L.N + 1)) s->walk_poisoned.assign(L.N + 1, 0);
      const double qnan = std::numeric_limits<double>::quiet_NaN();
      const size_t istride = (size_t)(L.N + 1) * L.max_stage_doubles;
      auto poison = [&](int slot, int off, int len) {
        if (off < 0) return;
        for (int b = 0; b < L.B; ++b) std::fill_n(s->h_inst_params.data() + (size_t)b * istride + (size_t)slot * L.max_stage_doubles + off, len, qnan);
      };
      for (int k = (replanning || s->walk_force_all) ? 0 : L.N - 1; k < L.N; ++k) {
        const int sl = slot_of(s, k);
        if (s->walk_poisoned[sl]) continue;
        poison(sl, c.off_lf, 12); poison(sl, c.off_rf, 12);
        if (c.z_follow != 0.0) poison(sl, c.off_xref_z, 1);
        s->walk_poisoned[sl] = 1;
      }
      if (!s->walk_poisoned[L.N]) { poison(L.N, c.toff_com, 3); poison(L.N, c.toff_lf, 12); poison(L.N, c.toff_rf, 12); s->walk_poisoned[L.N] = 1; }
    }
    s->walk_force_all = false;
  })
}

int mpc_walk_set_state(mpc_solver* s, const double* in) {
  MPC_TRY(s, {
    if (!s->walk_on || !in) throw std::runtime_error("walk_set_state: mpc_walk_init first");
    copy_sync(s, s->d_walk_state, in, (size_t)s->L.B * 48 * sizeof(double), hipMemcpyHostToDevice);
    s->walk_force_all = true;
  })
}

int mpc_walk_get_state(mpc_solver* s, double* out) {
  MPC_TRY(s, {
    if (!s->walk_on || !out) throw std::runtime_error("walk_get_state: mpc_walk_init first");
    copy_sync(s, out, s->d_walk_state, (size_t)s->L.B * 48 * sizeof(double), hipMemcpyDeviceToHost);
  })
}

int mpc_cycle(mpc_solver* s, const int32_t* desc, int32_t n_desc, const double* params, int32_t n_params) {
  MPC_TRY(s, {
    // the slot of stage 0 is recycled for the new last stage: no data movement, only the ring head moves
    const int slot = s->head;
    {  // does the appended stage have the table of the current last stage (what the speculative evaluation assumed)?
      const Layout& L = s->L;
      const int last = (s->head + L.N - 1) % L.N;
      const int32_t* hd = s->h_desc.data() + (size_t)last * L.max_stage_ints;
      const double* hp = s->h_params.data() + (size_t)last * L.max_stage_doubles;
      const bool same = s->h_len[2 * last] == n_desc && s->h_len[2 * last + 1] == n_params && std::memcmp(hd, desc, n_desc * sizeof(int32_t)) == 0 &&
                        (n_params == 0 || std::memcmp(hp, params, n_params * sizeof(double)) == 0);
      s->cycles_since_run += 1;
      // refine_appended_knot: another dynamics kind / contact list than the stage before it (descriptor words 0 .. 3)
      s->appended_changed = n_desc >= 4 && s->h_len[2 * last] >= 4 && std::memcmp(hd, desc, 4 * sizeof(int32_t)) != 0;
      s->appended_any = true;
      s->since_change = s->appended_changed ? 0 : (s->since_change < (1 << 20) ? s->since_change + 1 : s->since_change);
      // (a last stage whose parameters were patched after the speculative evaluation: the spare record is stale)
      s->spec_next_pending = same && s->spec_rec_valid && s->cycles_since_run == 1 && !s->slot_dirty[last] && !s->dirty_all;
    }
    upload_stage(s, slot, desc, n_desc, params, n_params);
    s->slot_dirty[slot] = 0;  // the recycled slot: its record is the speculative one or a fresh evaluation (knot_reused, k = N - 1)
    s->head = (s->head + 1) % s->L.N;
  })
}

int mpc_set_x0(mpc_solver* s, const double* x0) {
  MPC_TRY(s, {
    s->perfect_feedback = (x0 == nullptr);
    if (x0) {
      HIP_OK(hipMemcpyAsync(s->d_x0, x0, (size_t)s->L.B * s->L.nx * sizeof(double), hipMemcpyHostToDevice, s->stream));
      HIP_OK(hipStreamSynchronize(s->stream));
    }
  })
}

int mpc_set_tick_reuse(mpc_solver* s, int32_t on) {
  MPC_TRY(s, {
    // whole-body problems only: the vector-space stage kernel keeps its records in knot order (no ring), so the ring head must not move
    s->tick_reuse = on != 0 && s->L.space == MPC_SPACE_MULTIBODY;
    if (!s->tick_reuse) s->khead = 0;  // records go back to knot order; they are rewritten by the next pass (spec cleared below)
    s->reuse_this_pass = false;
    spec_clear(s);
  })
}

int mpc_profile(mpc_solver* s, int32_t mode) {
  MPC_TRY(s, {
    if (mode == 2) { for (auto& p : s->prof) p.used = 0; }
    else if (mode == 3 || mode == 4) s->phase_timers = (mode == 3);  // in-kernel phase timers (developer tooling, perturbs timing)
    else if (mode >= 16) { s->profiling = true; s->prof_mask = (unsigned)mode >> 4; }
    else { s->profiling = (mode != 0); s->prof_mask = ~0u; }
  })
}

// Occupancy table of the kernels one pass of this handle launches (developer tooling: tools/occupancy_report.py).  For entry idx:
// info = {threads per workgroup, VGPRs, scratch bytes per lane, static LDS, dynamic LDS, workgroups a CU can hold (the runtime's
// occupancy calculator for this block size and LDS request), workgroups per launch, wavefronts per SIMD at that residency}.
// Returns the number of entries.
int mpc_kernel_info(mpc_solver* s, int32_t idx, char* name, int32_t name_cap, int32_t* info) {
  if (!s) return -2;
  try {
    HIP_OK(hipSetDevice(s->dims.device));
    const Layout& L = s->L;
    struct Ent { const char* name; const void* fn; int block; int dyn; long long grid; };
    std::vector<Ent> e;
    const int J = s->eff_legs();
    if (L.space == MPC_SPACE_VECTOR) {
      e.push_back({"k_eval_vector<0> (stage kernel, value + derivatives)", (const void*)k_eval_vector<0>, 64, 0, (long long)(L.N + 1) * L.B});
      e.push_back({"k_eval_vector<1> (linesearch candidate)", (const void*)k_eval_vector<1>, 64, 0, (long long)(L.N + 1) * L.B});
    } else {
      const MbLds ml = make_mb_lds(L.nj, L.n / 2, L.nx - L.n / 2, L.m, L.nz, s->contact_dyn);
      e.push_back({"k_eval_multibody<0> (stage kernel, value + derivatives)", eval_multibody_kernel(0), EVAL_THREADS, ml.total_bytes, (long long)(L.N + 1) * L.B});
      e.push_back({"k_eval_multibody<3> (alpha = 1 candidate with derivatives)", eval_multibody_kernel(3), EVAL_THREADS, ml.total_bytes, (long long)(L.N + 1) * L.B});
      e.push_back({"k_eval_multibody<1> (backtracking candidates, values only)", eval_multibody_kernel(1), EVAL_THREADS, ml.total_bytes, (long long)(L.N + 1) * L.B});
    }
    const bool small = s->ric.np == 16 && s->ric.mp == 16 && L.c <= RIC_SMALL_THREADS;
    if (J > 1) {
      if (small) e.push_back({"k_riccati_mfma<256,16,false,true> (sweep, legs)", (const void*)k_riccati_mfma<RIC_SMALL_THREADS, 16, false, true>, RIC_SMALL_THREADS, s->ric.total_bytes, (long long)L.B * J});
#define X(ID, FN, FM, GF, ST, NPV, MPV) else if (s->ric_fixed == ID) e.push_back({"k_riccati_mfma<512,80,true,true," #FN "," #FM "," #GF "," #ST "> (sweep, legs, fixed dimensions)", (const void*)k_riccati_mfma<RIC_THREADS, 80, true, true, FN, FM, GF, ST>, RIC_THREADS, s->ric.total_bytes, (long long)L.B * J});
      MPC_FIXED_MODELS(X)
#undef X
      else if (s->ric.sq) e.push_back({"k_riccati_mfma<512,80,true,true> (sweep, legs)", (const void*)k_riccati_mfma<RIC_THREADS, 80, true, true>, RIC_THREADS, s->ric.total_bytes, (long long)L.B * J});
      else e.push_back({"k_riccati_mfma<512,80,false,true> (sweep, legs)", (const void*)k_riccati_mfma<RIC_THREADS, 80, false, true>, RIC_THREADS, s->ric.total_bytes, (long long)L.B * J});
      // (the instantiations the launch sites pick: the fixed-dimension ones when the handle's layout is one of MPC_FIXED_MODELS)
      const void* lk = s->lk.mp <= 16 ? (const void*)k_leg_knot<16> : (s->lk.mp <= 32 ? (const void*)k_leg_knot<32> : (const void*)k_leg_knot<48>);
      const void* lcd = (const void*)k_leg_condense<0>;
      const char *lk_name = "k_leg_knot", *lcd_name = "k_leg_condense", *lcmp_name = "k_leg_compose (first level of the tree over the cuts)", *ltd_name = "k_leg_tree_down (last level)";
      const void *lcmp_fixed = nullptr, *ltd_fixed = nullptr;
#define X(ID, FN, FM, GF, ST, NPV, MPV) if (s->ric_fixed == ID) { lk = (const void*)k_leg_knot<MPV, FN, FM>; lcd = (const void*)k_leg_condense<FN>; lcmp_fixed = (const void*)k_leg_compose<NPV, FN, FM>; ltd_fixed = (const void*)k_leg_tree_down<NPV, FN, FM>; \
        lk_name = "k_leg_knot<" #MPV "," #FN "," #FM "> (fixed dimensions)"; lcd_name = "k_leg_condense<" #FN "> (fixed dimensions)"; \
        lcmp_name = "k_leg_compose<" #NPV "," #FN "," #FM "> (first level of the tree over the cuts, fixed dimensions)"; ltd_name = "k_leg_tree_down<" #NPV "," #FN "," #FM "> (last level, fixed dimensions)"; }
      MPC_FIXED_MODELS(X)
#undef X
      e.push_back({lk_name, lk, LK_THREADS, s->lk.total_bytes, (long long)L.N * L.B});
      e.push_back({lcd_name, lcd, LK_THREADS, s->lc.total_bytes, (long long)(J - 1) * L.B});
      if (s->use_tree()) {
        const TreeDesc T = make_tree_desc(J);
        const void* lc = s->lx.np == 16 ? (const void*)k_leg_compose<16> : s->lx.np == 32 ? (const void*)k_leg_compose<32> : s->lx.np == 48 ? (const void*)k_leg_compose<48>
                       : s->lx.np == 64 ? (const void*)k_leg_compose<64> : (const void*)k_leg_compose<80>;
        const void* ld = s->lx.np == 16 ? (const void*)k_leg_tree_down<16> : s->lx.np == 32 ? (const void*)k_leg_tree_down<32> : s->lx.np == 48 ? (const void*)k_leg_tree_down<48>
                       : s->lx.np == 64 ? (const void*)k_leg_tree_down<64> : (const void*)k_leg_tree_down<80>;
        e.push_back({lcmp_name, lcmp_fixed ? lcmp_fixed : lc, LCMP_THREADS, s->lx.total_bytes, (long long)(T.lev_cnt[0] + 1) * L.B * 2});
        e.push_back({ltd_name, ltd_fixed ? ltd_fixed : ld, LK_THREADS, s->lx.total_bytes, (long long)T.lev_cnt[0] * L.B});
      } else {
      const void* lx = s->lx.np == 16 ? (const void*)k_leg_consensus<16> : s->lx.np == 32 ? (const void*)k_leg_consensus<32> : s->lx.np == 48 ? (const void*)k_leg_consensus<48>
                     : s->lx.np == 64 ? (const void*)k_leg_consensus<64> : (const void*)k_leg_consensus<80>;
      e.push_back({"k_leg_consensus", lx, LK_THREADS, s->lx.total_bytes, (long long)L.B});
      }
      e.push_back({"k_leg_apply", (const void*)k_leg_apply, 256, 0, (long long)L.N * L.B});
      e.push_back({"k_forward_phi (forward sweeps of the legs)", L.m <= 32 ? (const void*)k_forward_phi<4, 10> : (const void*)k_forward_phi<6, 10>, 512, (int)(2 * L.n * sizeof(double)), (long long)L.B * J});
    } else if (s->use_mfma_riccati) {
      const void* fn = small ? (const void*)k_riccati_mfma<RIC_SMALL_THREADS, 16> : (s->ric.sq && s->ric.np <= 80) ? (const void*)k_riccati_mfma<RIC_THREADS, 80, true>
                     : s->ric.sq ? (const void*)k_riccati_mfma<RIC_THREADS, 96, true> : s->ric.np <= 80 ? (const void*)k_riccati_mfma<RIC_THREADS, 80> : (const void*)k_riccati_mfma<RIC_THREADS, 96>;
      e.push_back({"k_riccati_mfma (serial sweep)", fn, small ? RIC_SMALL_THREADS : RIC_THREADS, s->ric.total_bytes, (long long)L.B});
    } else e.push_back({"k_riccati_backward (serial sweep, no matrix cores)", (const void*)k_riccati_backward, 256, (int)s->riccati_lds(), (long long)L.B});
    e.push_back({"k_duals", (const void*)k_duals, 256, (int)((L.nz + 3 * L.n + 16 + 3 * L.c) * sizeof(double)), (long long)(L.N + 1) * L.B});
    e.push_back({"k_lagrangian", (const void*)k_lagrangian, 64, 0, (long long)(L.N + 1) * L.B});
    if (idx < 0 || idx >= (int)e.size()) return (int)e.size();
    const Ent& k = e[idx];
    hipFuncAttributes at;
    HIP_OK(hipFuncGetAttributes(&at, k.fn));
    int nb = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k.fn, k.block, (size_t)k.dyn));
    if (name && name_cap > 0) { strncpy(name, k.name, name_cap - 1); name[name_cap - 1] = 0; }
    info[0] = k.block; info[1] = at.numRegs; info[2] = (int)at.localSizeBytes; info[3] = (int)at.sharedSizeBytes; info[4] = k.dyn; info[5] = nb;
    info[6] = (int)k.grid; info[7] = nb * ((k.block + 63) / 64) / 4;
    return (int)e.size();
  } catch (const std::exception& ex) { s->err = ex.what(); return -1; }
}

int mpc_profile_read(mpc_solver* s, int32_t slot, char* name, int32_t name_cap, int32_t* launches, double* total_ms) {
  if (!s) return -2;
  try {
    HIP_OK(hipSetDevice(s->dims.device));
    HIP_OK(hipStreamSynchronize(s->stream));
    const int nslots = (int)s->prof.size();
    if (slot < 0 || slot >= nslots) return nslots;
    const mpc_solver::ProfSlot& p = s->prof[slot];
    double tot = 0;
    for (size_t i = 0; i < p.used; ++i) {
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, p.ev[i].first, p.ev[i].second));
      tot += ms;
    }
    if (name && name_cap > 0) { std::strncpy(name, p.name ? p.name : "", name_cap - 1); name[name_cap - 1] = 0; }
    if (launches) *launches = (int32_t)p.used;
    if (total_ms) *total_ms = tot;
    return nslots;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

static void simulate_impl(mpc_solver* s, int32_t substeps, double dt, const double* f_ext) {
  if (substeps <= 0 || !(dt > 0.0)) throw std::runtime_error("simulate: substeps and dt must be positive");
  if (s->L.space != MPC_SPACE_MULTIBODY || s->h_desc[(size_t)slot_of(s, 0) * s->L.max_stage_ints] != MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER)
    throw std::runtime_error("simulate: only contact-constrained whole-body dynamics are supported");
  const double* d_f = nullptr;
  if (f_ext) {  // (a small synchronous upload: the push is an event of a few ticks, not part of the steady loop)
    if (!s->d_fext) s->d_fext = s->alloc<double>((size_t)s->L.B * 3);
    copy_sync(s, s->d_fext, f_ext, (size_t)s->L.B * 3 * sizeof(double), hipMemcpyHostToDevice);
    d_f = s->d_fext;
  }
  launch_eval_multibody(s->stream, s->args(), s->LT, s->d_tknots, s->d_mbwork, s->mb_work_stride, true, 0, 1, substeps, dt, false, d_f);
  HIP_OK(hipGetLastError());
  s->perfect_feedback = false;
}

int mpc_simulate(mpc_solver* s, int32_t substeps, double dt) {
  MPC_TRY(s, { simulate_impl(s, substeps, dt, nullptr); })
}

int mpc_simulate_push(mpc_solver* s, int32_t substeps, double dt, const double* f_ext) {
  MPC_TRY(s, { simulate_impl(s, substeps, dt, f_ext); })
}

int mpc_simulate_torque(mpc_solver* s, const double* x, const double* tau, int32_t substeps, double dt, double* wrenches) {
  MPC_TRY(s, {
    if (substeps <= 0 || !(dt > 0.0)) throw std::runtime_error("simulate: substeps and dt must be positive");
    if (!tau) throw std::runtime_error("simulate_torque: tau must not be null");
    if (s->L.space != MPC_SPACE_MULTIBODY || s->h_desc[(size_t)slot_of(s, 0) * s->L.max_stage_ints] != MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER)
      throw std::runtime_error("simulate: only contact-constrained whole-body dynamics are supported");
    const Layout& L = s->L;
    if (!s->d_simu) { s->d_simu = s->alloc<double>((size_t)L.B * L.m); s->d_simwr = s->alloc<double>((size_t)L.B * 12); }
    if (x) copy_sync(s, s->d_x0, x, (size_t)L.B * L.nx * sizeof(double), hipMemcpyHostToDevice);
    copy_sync(s, s->d_simu, tau, (size_t)L.B * L.m * sizeof(double), hipMemcpyHostToDevice);
    launch_eval_multibody(s->stream, s->args(), s->LT, s->d_tknots, s->d_mbwork, s->mb_work_stride, true, 0, 1, substeps, dt, false, nullptr, true,
                          s->d_simu, wrenches ? s->d_simwr : nullptr);
    HIP_OK(hipGetLastError());
    if (wrenches) copy_sync(s, wrenches, s->d_simwr, (size_t)L.B * 12 * sizeof(double), hipMemcpyDeviceToHost);
    s->perfect_feedback = false;
  })
}

// include/mpc_qp_abi.h: the low-level loop of the kinodynamic pipeline with nothing but the kernels between its stages.  Everything is enqueued on
// the QP handle's stream (the plan and the simulator are idle: their streams are drained first); one synchronisation at the end.
int mpc_qp_low_level_steps(mpc_qp_solver* qp, const mpc_qp_settings* S, mpc_solver* plan, mpc_solver* sim, int32_t nk, const int32_t* frames,
                           const double* weights, const double* cone, double kd, const int32_t* contact_states, const double* tau_max,
                           const double* x, int32_t steps, double dt, double* x_prev, double* x_out, double* tau, double* forces, mpc_qp_info* info) {
  if (!qp) return -2;
  try {
    if (!S || !plan || !sim || !contact_states || !tau_max) throw std::runtime_error("qp_low_level_steps: null argument");
    if (steps <= 0 || !(dt > 0.0)) throw std::runtime_error("qp_low_level_steps: steps and dt must be positive");
    qp_id_prepare(qp, nk, frames, weights, cone);
    const QpIdBuffers q = qp_id_buffers(qp);
    const Layout& P = plan->L;
    const Layout& Z = sim->L;
    const int nx = q.nq + q.nv, nu = q.nv - 6, nf = 6 * nk;
    if (plan->dims.device != q.device || sim->dims.device != q.device) throw std::runtime_error("qp_low_level_steps: the three handles must live on one device");
    if (P.B != q.B || Z.B != q.B) throw std::runtime_error("qp_low_level_steps: the three handles must have the same batch size");
    if (P.space != MPC_SPACE_MULTIBODY || P.nx != nx || P.n != 2 * q.nv || P.m != nf + nu || P.n > PIPE_MAX_N)
      throw std::runtime_error("qp_low_level_steps: the plan must be a multibody problem with nx = nq + nv and controls (6 nk contact wrench components, nv - 6 joint accelerations)");
    if (Z.space != MPC_SPACE_MULTIBODY || Z.nx != nx || Z.m != nu ||
        sim->h_desc[(size_t)slot_of(sim, 0) * Z.max_stage_ints] != MPC_DYN_MULTIBODY_CONSTRAINT_SEMIEULER)
      throw std::runtime_error("qp_low_level_steps: the simulator handle must hold whole-body contact dynamics with nu = nv - 6 (the handle of mpc_simulate_torque)");
    if (plan->async_pending > 0) throw std::runtime_error("qp_low_level_steps: the plan has ticks in flight (mpc_wait first)");
    HIP_OK(hipStreamSynchronize(plan->stream));
    HIP_OK(hipStreamSynchronize(sim->stream));
    if (!sim->d_simu) { sim->d_simu = sim->alloc<double>((size_t)Z.B * Z.m); sim->d_simwr = sim->alloc<double>((size_t)Z.B * 12); HIP_OK(hipStreamSynchronize(sim->stream)); }
    const size_t B = q.B;
    double* scr = qp_scratch(qp, B * nx + B * nf + nu);  // x before the last period | forces + df | tau_max
    double *d_xprev = scr, *d_fnew = scr + B * nx, *d_taumax = d_fnew + B * nf;
    hipStream_t st = q.stream;
    if (x) HIP_OK(hipMemcpyAsync(sim->d_x0, x, B * nx * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(q.cs, contact_states, B * nk * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIP_OK(hipMemcpyAsync(d_taumax, tau_max, nu * sizeof(double), hipMemcpyHostToDevice, st));
    PipeArgs p;
    p.xs = plan->d_xs; p.us = plan->d_us; p.gains = plan->d_gains; p.knots = plan->d_knots;
    p.N = P.N; p.nx = nx; p.nq = q.nq; p.nv = q.nv; p.n = P.n; p.m = P.m; p.gain_stride = P.gain_stride; p.oK = P.oK; p.knot_stride = P.knot_stride; p.oXD = P.oXD;
    p.slot0 = plan->khead % P.N;
    p.x = sim->d_x0; p.xrob = q.xrob; p.acc = q.acc; p.f = q.f; p.sol = q.sol; p.nk = nk; p.qn = q.n; p.tau_max = d_taumax; p.sim_u = sim->d_simu; p.f_new = d_fnew;
    const SolverArgs za = sim->args();
    for (int step = 0; step < steps; ++step) {
      if (step == steps - 1 && x_prev) HIP_OK(hipMemcpyAsync(d_xprev, sim->d_x0, B * nx * sizeof(double), hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(k_pipe_feedback, dim3((unsigned)B), dim3(64), 0, st, p);
      qp_id_enqueue(qp, S, kd);
      qp_launch_solve(qp, S);
      hipLaunchKernelGGL(k_pipe_torque, dim3((unsigned)B), dim3(64), 0, st, p);
      launch_eval_multibody(st, za, sim->LT, sim->d_tknots, sim->d_mbwork, sim->mb_work_stride, true, 0, 1, 1, dt, false, nullptr, true, sim->d_simu, nullptr);
      HIP_OK(hipGetLastError());
    }
    if (x_prev) HIP_OK(hipMemcpyAsync(x_prev, d_xprev, B * nx * sizeof(double), hipMemcpyDeviceToHost, st));
    if (x_out) HIP_OK(hipMemcpyAsync(x_out, sim->d_x0, B * nx * sizeof(double), hipMemcpyDeviceToHost, st));
    if (tau) HIP_OK(hipMemcpyAsync(tau, sim->d_simu, B * nu * sizeof(double), hipMemcpyDeviceToHost, st));
    if (forces) HIP_OK(hipMemcpyAsync(forces, d_fnew, B * nf * sizeof(double), hipMemcpyDeviceToHost, st));
    if (info) HIP_OK(hipMemcpyAsync(info, q.info, B * sizeof(mpc_qp_info), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    sim->perfect_feedback = false;
    return 0;
  } catch (const std::exception& e) {
    qp_set_error(qp, e.what());
    return -1;
  }
}

int mpc_get_x0(mpc_solver* s, double* x0) {
  MPC_TRY(s, {
    HIP_OK(hipMemcpyAsync(x0, s->d_x0, (size_t)s->L.B * s->L.nx * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
  })
}

// ---- solver-state checkpoint (layout shared with oracle/capi.cpp) ---------------------------------------------------------------
#define MPC_STATE_MAGIC 20250304.0
#define MPC_STATE_HEADER 16
static int64_t state_doubles(const mpc_solver* s) {
  const Layout& L = s->L;
  const int64_t N1 = L.N + 1;
  return MPC_STATE_HEADER + N1 * (2 + (int64_t)L.max_stage_ints + L.max_stage_doubles) +
         (int64_t)L.B * (N1 * L.nx + (int64_t)L.N * L.m + N1 * L.c + N1 * L.n + L.nx + 4);
}

int64_t mpc_state_size(mpc_solver* s) { return s ? state_doubles(s) : -1; }

int64_t mpc_get_state(mpc_solver* s, double* buf, int64_t cap) {
  if (!s) return -2;
  try {
    HIP_OK(hipSetDevice(s->dims.device));
    const Layout& L = s->L;
    const int64_t need = state_doubles(s), N1 = L.N + 1;
    if (!buf || cap < need) throw std::runtime_error("get_state: buffer too small (mpc_state_size doubles needed)");
    HIP_OK(hipStreamSynchronize(s->stream));
    if (s->async_pending > 0) throw std::runtime_error("get_state: asynchronous ticks in flight (mpc_wait first)");
    double* o = buf;
    const double hdr[MPC_STATE_HEADER] = {MPC_STATE_MAGIC, (double)L.B, (double)L.N, (double)L.nx, (double)L.n, (double)L.m, (double)L.c, (double)L.space,
                                          s->perfect_feedback ? 1.0 : 0.0, (double)L.max_stage_ints, (double)L.max_stage_doubles, (double)(s->since_change + 1) /* 0: a state saved before the field existed */, 0, 0, 0, 0};
    std::memcpy(o, hdr, sizeof(hdr)); o += MPC_STATE_HEADER;
    for (int k = 0; k <= L.N; ++k) {  // knot order: the ring is unrolled
      const int sl = slot_of(s, k);
      *o++ = (double)s->h_len[2 * sl]; *o++ = (double)s->h_len[2 * sl + 1];
      for (int i = 0; i < L.max_stage_ints; ++i) *o++ = i < s->h_len[2 * sl] ? (double)s->h_desc[(size_t)sl * L.max_stage_ints + i] : 0.0;
      for (int i = 0; i < L.max_stage_doubles; ++i) *o++ = i < s->h_len[2 * sl + 1] ? s->h_params[(size_t)sl * L.max_stage_doubles + i] : 0.0;
    }
    copy_sync(s, o, s->d_xs, (size_t)L.B * N1 * L.nx * sizeof(double), hipMemcpyDeviceToHost); o += (size_t)L.B * N1 * L.nx;
    copy_sync(s, o, s->d_us, (size_t)L.B * L.N * L.m * sizeof(double), hipMemcpyDeviceToHost); o += (size_t)L.B * L.N * L.m;
    copy_sync(s, o, s->d_vs, (size_t)L.B * N1 * L.c * sizeof(double), hipMemcpyDeviceToHost); o += (size_t)L.B * N1 * L.c;
    copy_sync(s, o, s->d_lams, (size_t)L.B * N1 * L.n * sizeof(double), hipMemcpyDeviceToHost); o += (size_t)L.B * N1 * L.n;
    copy_sync(s, o, s->d_x0, (size_t)L.B * L.nx * sizeof(double), hipMemcpyDeviceToHost); o += (size_t)L.B * L.nx;
    std::vector<InstState> st(L.B);
    copy_sync(s, st.data(), s->d_inst, L.B * sizeof(InstState), hipMemcpyDeviceToHost);
    for (int b = 0; b < L.B; ++b) { *o++ = st[b].mu; *o++ = st[b].inner_tol; *o++ = st[b].prim_tol; *o++ = 0.0; }
    return need;
  } catch (const std::exception& e) { s->err = e.what(); return -1; }
}

int mpc_set_state(mpc_solver* s, const double* buf, int64_t len) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    const int64_t need = state_doubles(s), N1 = L.N + 1;
    if (!buf || len < need) throw std::runtime_error("set_state: truncated state");
    const double* o = buf;
    if (o[0] != MPC_STATE_MAGIC || (int)o[1] != L.B || (int)o[2] != L.N || (int)o[3] != L.nx || (int)o[4] != L.n || (int)o[5] != L.m || (int)o[6] != L.c ||
        (int)o[7] != L.space || (int)o[9] != L.max_stage_ints || (int)o[10] != L.max_stage_doubles)
      throw std::runtime_error("set_state: the state was saved by a handle of other dimensions");
    HIP_OK(hipStreamSynchronize(s->stream));
    if (s->async_pending > 0) throw std::runtime_error("set_state: asynchronous ticks in flight (mpc_wait first)");
    s->perfect_feedback = o[8] != 0.0;
    s->since_change = (int)o[11] > 0 ? (int)o[11] - 1 : 1 << 20;
    o += MPC_STATE_HEADER;
    spec_clear(s);
    s->leg_guess_valid = false; s->reuse_this_pass = false; s->spec_skip_pass = false;
    s->head = 0; s->khead = 0;
    std::vector<int32_t> desc(L.max_stage_ints);
    for (int k = 0; k <= L.N; ++k) {
      const int nd = (int)o[0], np = (int)o[1];
      o += 2;
      if (nd < 0 || nd > L.max_stage_ints || np < 0 || np > L.max_stage_doubles) throw std::runtime_error("set_state: corrupt stage table");
      for (int i = 0; i < L.max_stage_ints; ++i) desc[i] = (int32_t)o[i];
      if (nd > 0) { s->h_len[2 * k] = -1; upload_stage(s, k, desc.data(), nd, o + L.max_stage_ints, np); }  // (h_len = -1: never equal to the mirror)
      o += L.max_stage_ints + L.max_stage_doubles;
    }
    HIP_OK(hipMemcpyAsync(s->d_xs, o, (size_t)L.B * N1 * L.nx * sizeof(double), hipMemcpyHostToDevice, s->stream)); o += (size_t)L.B * N1 * L.nx;
    HIP_OK(hipMemcpyAsync(s->d_us, o, (size_t)L.B * L.N * L.m * sizeof(double), hipMemcpyHostToDevice, s->stream)); o += (size_t)L.B * L.N * L.m;
    HIP_OK(hipMemcpyAsync(s->d_vs, o, (size_t)L.B * N1 * L.c * sizeof(double), hipMemcpyHostToDevice, s->stream)); o += (size_t)L.B * N1 * L.c;
    HIP_OK(hipMemcpyAsync(s->d_lams, o, (size_t)L.B * N1 * L.n * sizeof(double), hipMemcpyHostToDevice, s->stream)); o += (size_t)L.B * N1 * L.n;
    HIP_OK(hipMemcpyAsync(s->d_x0, o, (size_t)L.B * L.nx * sizeof(double), hipMemcpyHostToDevice, s->stream)); o += (size_t)L.B * L.nx;
    std::vector<InstState> st(L.B);
    copy_sync(s, st.data(), s->d_inst, L.B * sizeof(InstState), hipMemcpyDeviceToHost);
    for (int b = 0; b < L.B; ++b) { st[b].mu = o[0]; st[b].inner_tol = o[1]; st[b].prim_tol = o[2]; o += 4; }
    copy_sync(s, s->d_inst, st.data(), L.B * sizeof(InstState), hipMemcpyHostToDevice);
    // every stage went through upload_stage above, which hands the instances the SHARED tables again: with the reference generator in the library the next
    // mpc_walk_update must rewrite the references of every knot, not only of the appended one
    s->walk_force_all = s->walk_on;
  })
}

int mpc_setup(mpc_solver* s) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    hipLaunchKernelGGL(k_setup, dim3(L.N + 2, L.B), dim3(256), 0, s->stream, s->args());  // asynchronous: the MPC loop calls it every tick
    HIP_OK(hipGetLastError());
  })
}

int mpc_run(mpc_solver* s, const double* xs, const double* us, mpc_stats* stats) {
  MPC_TRY(s, {
    s->appended_changed = false; s->appended_any = false; s->refine_now = false;  // (an uploaded warm start is the caller's: it stays as it is)
    spec_clear(s);
    s->reuse_this_pass = false;
    // A multi-iteration solve starts from an iterate the handle has never seen (cold start): the cut Hessians kept from the last pass
    // belong to another trajectory, the first pass refreshes them with its extra sweeps.  A single iteration (max_iters = 1) is an MPC
    // tick of the reference loop, whose xs / us are the previous solution shifted by one knot (fulldynamic_talos.py:532-540): the
    // guesses stay — refreshing them there would cost one sweep per tree level on every tick of the drop-in path.
    if (s->opt.max_iters > 1) s->leg_guess_valid = false;
    s->spec_skip_pass = false;
    const Layout& L = s->L;
    HIP_OK(hipMemcpyAsync(s->d_xs, xs, (size_t)L.B * (L.N + 1) * L.nx * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_us, us, (size_t)L.B * L.N * L.m * sizeof(double), hipMemcpyHostToDevice, s->stream));
    run_impl(s, stats);
  })
}

int mpc_run_shifted(mpc_solver* s, mpc_stats* stats) {
  MPC_TRY(s, {
    if (s->tick_reuse) { s->khead = (s->khead + 1) % s->L.N; s->reuse_this_pass = true; s->spec_next_now = s->spec_next_pending; }  // the records move one knot on with the iterate
    s->spec_next_pending = false; s->cycles_since_run = 0;
    begin_refine(s);
    begin_reuse_pass(s);
    launch_shift(s);
    run_impl(s, stats);
  })
}

int mpc_run_shifted_async(mpc_solver* s) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    if (s->async_pending >= mpc_solver::ASYNC_DEPTH) throw std::runtime_error("run_shifted_async: too many ticks in flight (call mpc_wait first)");
    const int slot = (s->async_head + s->async_pending) % mpc_solver::ASYNC_DEPTH;
    if (!s->h_status[slot]) {
      HIP_OK(hipHostMalloc((void**)&s->h_status[slot], L.B * sizeof(InstState), hipHostMallocDefault));
      HIP_OK(hipEventCreateWithFlags(&s->status_ev[slot], hipEventDisableTiming));
    }
    if (s->tick_reuse) { s->khead = (s->khead + 1) % L.N; s->reuse_this_pass = true; s->spec_next_now = s->spec_next_pending; }
    s->spec_next_pending = false; s->cycles_since_run = 0;
    begin_refine(s);
    begin_reuse_pass(s);
    launch_shift(s);
    hipLaunchKernelGGL(k_begin_run, dim3(L.B), dim3(64), 0, s->stream, s->args());
    launch_refine(s);
    // with max_iters = 1 one pass takes the step; a few iterations per tick (max_iters <= 4) are enqueued together — a younger tick
    // may be queued behind this one before its status is read; workgroups of instances that are done exit at once
    // (corrector_prim_tol: on the runs it applies to — corrector_window — one pass more, which instances that do not need it sit out)
    const int n_pass = (s->opt.max_iters < 1 ? 1 : (s->opt.max_iters > 4 ? 4 : s->opt.max_iters)) + (s->corrector_armed() ? 1 : 0);
    for (int p = 0; p < n_pass; ++p) { s->pass_in_run = p; s->pass_is_corrector = s->corrector_armed() && p == n_pass - 1; launch_pass(s); }
    s->pass_is_corrector = false;
    s->async_passes[slot] = n_pass;
    HIP_OK(hipMemcpyAsync(s->h_status[slot], s->d_inst, L.B * sizeof(InstState), hipMemcpyDeviceToHost, s->stream));
    // xs[1] of every instance — the state the next tick will take as its measurement under perfect-model feedback, and what a
    // reference generator needs to plan that tick (mpc_wait_state) — rides along: B rows of nx doubles out of the iterate
    if (!s->h_xnext[slot]) HIP_OK(hipHostMalloc((void**)&s->h_xnext[slot], (size_t)L.B * L.nx * sizeof(double), hipHostMallocDefault));
    HIP_OK(hipMemcpy2DAsync(s->h_xnext[slot], (size_t)L.nx * sizeof(double), s->d_xs + L.nx, (size_t)(L.N + 1) * L.nx * sizeof(double),
                            (size_t)L.nx * sizeof(double), (size_t)L.B, hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipEventRecord(s->status_ev[slot], s->stream));
    s->async_pending += 1;
  })
}

// Completes the OLDEST tick in flight.  If it is also the only one, an instance whose pass was a BCL update without a
// step gets its further passes now (as mpc_run_shifted does); with a younger tick already queued behind it the instance
// simply carries on in that tick.
static void wait_impl(mpc_solver* s, mpc_stats* stats, double* x_next) {
  const Layout& L = s->L;
  if (s->async_pending == 0) {  // nothing in flight: the status (and next states) of the last completed tick, read synchronously
    std::vector<InstState> stv(L.B);
    copy_sync(s, stv.data(), s->d_inst, L.B * sizeof(InstState), hipMemcpyDeviceToHost);
    report_status(s, L.B, stv.data(), stats);
    if (x_next) {
      HIP_OK(hipMemcpy2DAsync(x_next, (size_t)L.nx * sizeof(double), s->d_xs + L.nx, (size_t)(L.N + 1) * L.nx * sizeof(double),
                              (size_t)L.nx * sizeof(double), (size_t)L.B, hipMemcpyDeviceToHost, s->stream));
      HIP_OK(hipStreamSynchronize(s->stream));
    }
    return;
  }
  const int slot = s->async_head;
  s->async_head = (s->async_head + 1) % mpc_solver::ASYNC_DEPTH;
  s->async_pending -= 1;
  HIP_OK(hipEventSynchronize(s->status_ev[slot]));
  const InstState* st = s->h_status[slot];
  bool done = true;
  for (int b = 0; b < L.B; ++b) if (!st[b].done) done = false;
  if (!done && s->async_pending == 0) {
    run_impl(s, stats, s->async_passes[slot]);  // continues after the enqueued passes, which are complete
    if (x_next) {
      HIP_OK(hipMemcpy2DAsync(x_next, (size_t)L.nx * sizeof(double), s->d_xs + L.nx, (size_t)(L.N + 1) * L.nx * sizeof(double),
                              (size_t)L.nx * sizeof(double), (size_t)L.B, hipMemcpyDeviceToHost, s->stream));
      HIP_OK(hipStreamSynchronize(s->stream));
    }
  } else {
    report_status(s, L.B, st, stats);
    if (x_next) std::memcpy(x_next, s->h_xnext[slot], (size_t)L.B * L.nx * sizeof(double));
  }
}

int mpc_wait(mpc_solver* s, mpc_stats* stats) {
  MPC_TRY(s, { wait_impl(s, stats, nullptr); })
}

int mpc_wait_state(mpc_solver* s, mpc_stats* stats, double* x_next) {
  MPC_TRY(s, { wait_impl(s, stats, x_next); })
}

int mpc_poll(mpc_solver* s, int32_t* in_flight, int32_t* completed) {
  MPC_TRY(s, {
    int done = 0;
    for (int i = 0; i < s->async_pending; ++i) {
      const hipError_t q = hipEventQuery(s->status_ev[(s->async_head + i) % mpc_solver::ASYNC_DEPTH]);
      if (q == hipSuccess) ++done;
      else if (q != hipErrorNotReady) HIP_OK(q);
    }
    if (in_flight) *in_flight = s->async_pending;
    if (completed) *completed = done;
  })
}

int mpc_get_results(mpc_solver* s, double* xs, double* us, double* K, double* kff, double* vs, double* lams) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    const size_t B = L.B, N1 = L.N + 1, N = L.N;
    HIP_OK(hipStreamSynchronize(s->stream));
    if (xs) copy_sync(s, xs, s->d_xs, B * N1 * L.nx * sizeof(double), hipMemcpyDeviceToHost);
    if (us) copy_sync(s, us, s->d_us, B * N * L.m * sizeof(double), hipMemcpyDeviceToHost);
    if (vs) copy_sync(s, vs, s->d_vs, B * N1 * L.c * sizeof(double), hipMemcpyDeviceToHost);
    if (lams) copy_sync(s, lams, s->d_lams, B * N1 * L.n * sizeof(double), hipMemcpyDeviceToHost);
    if (K || kff) {
      // gains live inside the gain records: strided 2-D copies
      for (size_t b = 0; b < B; ++b) {
        const double* g0 = s->d_gains + b * N1 * L.gain_stride;
        if (K) HIP_OK(hipMemcpy2DAsync(K + b * N * L.m * L.n, (size_t)L.m * L.n * sizeof(double), g0 + L.oK, (size_t)L.gain_stride * sizeof(double),
                                       (size_t)L.m * L.n * sizeof(double), N, hipMemcpyDeviceToHost, s->stream));
        if (kff) HIP_OK(hipMemcpy2DAsync(kff + b * N * L.m, (size_t)L.m * sizeof(double), g0 + L.ok, (size_t)L.gain_stride * sizeof(double),
                                         (size_t)L.m * sizeof(double), N, hipMemcpyDeviceToHost, s->stream));
      }
      HIP_OK(hipStreamSynchronize(s->stream));
    }
  })
}

int mpc_get_gain(mpc_solver* s, int32_t k, double* K_k, double* kff_k) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    if (k < 0 || k >= L.N) throw std::runtime_error("get_gain: knot index out of range");
    const size_t pitch = (size_t)(L.N + 1) * L.gain_stride * sizeof(double);
    const double* g = s->d_gains + (size_t)k * L.gain_stride;
    if (K_k) HIP_OK(hipMemcpy2DAsync(K_k, (size_t)L.m * L.n * sizeof(double), g + L.oK, pitch, (size_t)L.m * L.n * sizeof(double), L.B, hipMemcpyDeviceToHost, s->stream));
    if (kff_k) HIP_OK(hipMemcpy2DAsync(kff_k, (size_t)L.m * sizeof(double), g + L.ok, pitch, (size_t)L.m * sizeof(double), L.B, hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
  })
}

int mpc_get_stage_data(mpc_solver* s, int32_t k, double* xdot, double* wrenches) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    if (k < 0 || k >= L.N) throw std::runtime_error("stage index out of range");
    HIP_OK(hipStreamSynchronize(s->stream));
    for (int b = 0; b < L.B; ++b) {
      const double* kn = s->d_knots + ((size_t)b * (L.N + 1) + (k < L.N ? (s->khead + k) % L.N : L.N)) * L.knot_stride;
      if (xdot) copy_sync(s, xdot + (size_t)b * L.n, kn + L.oXD, L.n * sizeof(double), hipMemcpyDeviceToHost);
      if (wrenches) copy_sync(s, wrenches + (size_t)b * 12, kn + L.oWR, 12 * sizeof(double), hipMemcpyDeviceToHost);
    }
  })
}

int mpc_set_failure_policy(mpc_solver* s, int32_t isolate) {
  MPC_TRY(s, { s->isolate = isolate != 0; })
}

int mpc_revive_instance(mpc_solver* s, int32_t dst, int32_t src) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    if (dst < 0 || dst >= L.B || src < 0 || src >= L.B || src == dst) throw std::runtime_error("revive_instance: instance index out of range");
    if (s->async_pending > 0) throw std::runtime_error("revive_instance: ticks in flight (call mpc_wait first)");
    auto row = [&](double* base, size_t per_inst) {
      HIP_OK(hipMemcpyAsync(base + (size_t)dst * per_inst, base + (size_t)src * per_inst, per_inst * sizeof(double), hipMemcpyDeviceToDevice, s->stream));
    };
    row(s->d_xs, (size_t)(L.N + 1) * L.nx); row(s->d_us, (size_t)L.N * L.m);
    row(s->d_vs, (size_t)(L.N + 1) * L.c); row(s->d_vs_e, (size_t)(L.N + 1) * L.c);
    // the co-state multipliers are indexed (b (N + 1) + k) n by every kernel and by mpc_get_results (the buffers merely carry B n doubles of slack)
    row(s->d_lams, (size_t)(L.N + 1) * L.n); row(s->d_lams_e, (size_t)(L.N + 1) * L.n);
    row(s->d_x0, (size_t)L.nx);
    HIP_OK(hipMemcpyAsync(s->d_inst + dst, s->d_inst + src, sizeof(InstState), hipMemcpyDeviceToDevice, s->stream));
    if (s->d_spec) HIP_OK(hipMemsetAsync(s->d_spec + dst, 0, sizeof(int), s->stream));  // its knot records belong to the old iterate
    s->leg_guess_valid = false;  // the kept cut Hessians of dst belong to the old iterate as well
    HIP_OK(hipStreamSynchronize(s->stream));
  })
}

int mpc_debug_evaluate(mpc_solver* s, const double* xs, const double* us) {
  MPC_TRY(s, {
    const Layout& L = s->L;
    HIP_OK(hipMemcpyAsync(s->d_xs, xs, (size_t)L.B * (L.N + 1) * L.nx * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_us, us, (size_t)L.B * L.N * L.m * sizeof(double), hipMemcpyHostToDevice, s->stream));
    // evaluate with every instance marked active
    std::vector<InstState> st(L.B);
    HIP_OK(hipMemcpyAsync(st.data(), s->d_inst, L.B * sizeof(InstState), hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    for (auto& i : st) { i.done = 0; i.skip_step = 0; if (i.mu <= 0) i.mu = s->opt.mu_init; }
    HIP_OK(hipMemcpyAsync(s->d_inst, st.data(), L.B * sizeof(InstState), hipMemcpyHostToDevice, s->stream));
    launch_eval(s, false);
    HIP_OK(hipStreamSynchronize(s->stream));
  })
}

int mpc_debug_get(mpc_solver* s, const char* name, int32_t b, int32_t k, double* out, int32_t cap) {
  if (!s) return -2;
  try {
    const Layout& L = s->L;
    if (b < 0 || b >= L.B || k < 0 || k > L.N) throw std::runtime_error("debug_get: index out of range");
    HIP_OK(hipSetDevice(s->dims.device));
    HIP_OK(hipStreamSynchronize(s->stream));
    std::vector<double> kn(L.knot_stride), g(L.gain_stride);
    copy_sync(s, kn.data(), s->d_knots + ((size_t)b * (L.N + 1) + (k < L.N ? (s->khead + k) % L.N : L.N)) * L.knot_stride, L.knot_stride * sizeof(double), hipMemcpyDeviceToHost);
    copy_sync(s, g.data(), s->d_gains + ((size_t)b * (L.N + 1) + k) * L.gain_stride, L.gain_stride * sizeof(double), hipMemcpyDeviceToHost);
    const int n = L.n, nz = L.nz, c = (int)kn[L.oMISC + MISC_NC], m = (int)kn[L.oMISC + MISC_M], nzk = n + m;
    const std::string nm(name);
    std::vector<double> v;
    auto mat = [&](const double* src, int rows, int cols, int ld) { v.resize((size_t)rows * cols); for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) v[(size_t)i * cols + j] = src[(size_t)i * ld + j]; };
    auto dev_vec = [&](const double* d_ptr, int len) { v.resize(len); copy_sync(s, v.data(), d_ptr, len * sizeof(double), hipMemcpyDeviceToHost); };
    if (nm == "ls") {  // linesearch of the last pass: phi0, dphi0, alpha, backtracking steps, then the merit of every candidate alpha_i = 2^-i (sum over the knots; candidates > 0 only if the pass backtracked)
      InstState st;
      copy_sync(s, &st, s->d_inst + b, sizeof(InstState), hipMemcpyDeviceToHost);
      std::vector<double> tp((size_t)L.n_alpha * (L.N + 1));
      copy_sync(s, tp.data(), s->d_trial_phi + (size_t)b * L.n_alpha * (L.N + 1), tp.size() * sizeof(double), hipMemcpyDeviceToHost);
      v = {st.phi0, st.dphi0, st.alpha, (double)st.ls_step};
      for (int i = 0; i < L.n_alpha; ++i) { double t = 0; for (int kk = 0; kk <= L.N; ++kk) t += tp[(size_t)i * (L.N + 1) + kk]; v.push_back(t); }
    }
    else if (nm == "inst_params") {  // the parameter table instance b uses at knot k (its own copy after mpc_enable_instance_params, else the shared one)
      const int slot = slot_of(s, k);
      const double* src = s->d_inst_params ? s->d_inst_params + ((size_t)b * (L.N + 1) + slot) * L.max_stage_doubles : s->d_stage_params + (size_t)slot * L.max_stage_doubles;
      dev_vec(src, s->h_len[2 * slot + 1]);
    }
    else if (nm == "H") mat(kn.data() + L.oH, nzk, nzk, nz);
    else if (nm == "grad") mat(kn.data() + L.oG, 1, nzk, nz);
    else if (nm == "AB") mat(kn.data() + L.oAB, k < L.N ? n : 0, nzk, nz);
    else if (nm == "f") mat(kn.data() + L.oF, 1, k < L.N ? n : 0, n);
    else if (nm == "E6") mat(kn.data() + L.oE6, 6, 6, 6);
    else if (nm == "D12") mat(kn.data() + L.oD12, 1, 74, 74);  // D1_b (36) | Dd_b (36) | dt | valid (layout.h)
    else if (nm == "cval") mat(kn.data() + L.oCV, 1, c, c);
    else if (nm == "act") mat(kn.data() + L.oACT, 1, c, c);  // active flags of the constraint rows (developer probes)
    else if (nm == "lo") mat(kn.data() + L.oLO, 1, c, c);
    else if (nm == "hi") mat(kn.data() + L.oHI, 1, c, c);
    else if (nm == "ctype") mat(kn.data() + L.oCT, 1, c, c);
    else if (nm == "CD") mat(kn.data() + L.oCD, c, nzk, nz);
    else if (nm == "cost") mat(kn.data() + L.oMISC + MISC_COST, 1, 1, 1);
    else if (nm == "xnext") mat(kn.data() + L.oXN, 1, L.nx, L.nx);
    else if (nm == "xdot") mat(kn.data() + L.oXD, 1, n, n);
    else if (nm == "wrench") mat(kn.data() + L.oWR, 1, 12, 12);
    else if (nm == "P") mat(g.data() + L.oP, n, n, n);
    else if (nm == "p") mat(g.data() + L.op, 1, n, n);
    else if (nm == "K") mat(g.data() + L.oK, m, n, n);
    else if (nm == "kff") mat(g.data() + L.ok, 1, m, m);
    else if (nm == "Knu") { for (int i = 0; i < c; ++i) if (kn[L.oACT + i] == 0.0) for (int z = 0; z < n; ++z) g[L.oKnu + i * n + z] = 0.0; mat(g.data() + L.oKnu, c, n, n); }  // inactive rows are not written by the sweep
    else if (nm == "knu") mat(g.data() + L.oknu, 1, c, c);
    else if (nm == "Mx") mat(g.data() + L.oMx, k < L.N ? n : 0, n, n);
    else if (nm == "mx") mat(g.data() + L.omx, 1, k < L.N ? n : 0, n);
    else if (nm == "Phi") mat(g.data() + L.oPhi, k < L.N ? n : 0, n, n);
    else if (nm == "phi") mat(g.data() + L.ophi, 1, k < L.N ? n : 0, n);
    else if (nm == "Gam") mat(g.data() + L.oGam, n, n, n);
    else if (nm == "Ku") mat(g.data() + L.oKu, m, n, n);
    else if (nm == "Lm") mat(g.data() + L.oLm, n, n, n);
    else if (nm == "Mu") mat(g.data() + L.oMu, m, m, L.mpad);
    else if (nm == "Znu" || nm == "Knup") {  // rows of the ACTIVE constraints, scattered to their row numbers like the oracle's (inactive rows zero)
      const bool zn = nm == "Znu";
      const int cols = zn ? m : n, ld = zn ? L.mpad : n;
      v.assign((size_t)c * cols, 0.0);
      int ai = 0;
      for (int i = 0; i < c; ++i) if (kn[L.oACT + i] != 0.0) { for (int j = 0; j < cols; ++j) v[(size_t)i * cols + j] = g[(zn ? L.oZnu : L.oKnup) + ai * ld + j]; ++ai; }
    }
    else if (nm == "Sg" || nm == "sg" || nm == "Zx" || nm == "zc" || nm == "calP" || nm == "calp" || nm == "theta") {
      // records of parametric leg k (k = leg index here)
      if (!s->d_legbuf || k >= s->leg_cap - 1) throw std::runtime_error("debug_get: no such leg record");
      const double* lr = s->d_legbuf + ((size_t)b * (s->leg_cap - 1) + k) * L.leg_stride;
      if (nm == "Sg") dev_vec(lr + L.lSg, n * n); else if (nm == "sg") dev_vec(lr + L.lsg, n); else if (nm == "Zx") dev_vec(lr + L.lZx, n * n);
      else if (nm == "zc") dev_vec(lr + L.lzc, n); else if (nm == "calP") dev_vec(lr + L.ldP, n * n);  // what the consensus worked on (the oracle's "calP")
      else if (nm == "calp") dev_vec(lr + L.lcp, n);
      else dev_vec(lr + L.lth, n);
    }
    else if (nm == "ls_knot") {  // knot k of the last pass: merit of every linesearch candidate alpha_i = 2^-i at this knot, then the knot's cost and penalty at the current point (developer probes)
      std::vector<double> tp((size_t)L.n_alpha * (L.N + 1));
      copy_sync(s, tp.data(), s->d_trial_phi + (size_t)b * L.n_alpha * (L.N + 1), tp.size() * sizeof(double), hipMemcpyDeviceToHost);
      for (int i = 0; i < L.n_alpha; ++i) v.push_back(tp[(size_t)i * (L.N + 1) + k]);
      v.push_back(kn[L.oMISC + MISC_COST]); v.push_back(kn[L.oMISC + MISC_PEN]);
    }
    else if (nm == "fixed_dims") v = {(double)s->ric_fixed};  // which fixed-dimension instantiations serve this handle (0: the generic kernels)
    else if (nm == "ric_prof") { dev_vec(s->d_prof + (size_t)b * 64, 64); HIP_OK(hipMemsetAsync(s->d_prof + (size_t)b * 64, 0, 64 * sizeof(double), s->stream)); HIP_OK(hipStreamSynchronize(s->stream)); }
    else if (nm == "dx") dev_vec(s->d_dxs + ((size_t)b * (L.N + 1) + k) * n, n);
    else if (nm == "du") { if (k >= L.N) throw std::runtime_error("no du at the terminal knot"); dev_vec(s->d_dus + ((size_t)b * L.N + k) * L.m, L.m); }
    else if (nm == "dvs") dev_vec(s->d_dvs + ((size_t)b * (L.N + 1) + k) * L.c, L.c);
    else if (nm == "dlams") dev_vec(s->d_dlams + ((size_t)b * (L.N + 1) + k) * n, n);
    else throw std::runtime_error("debug_get: unknown quantity " + nm);
    if ((int)v.size() > cap) throw std::runtime_error("debug_get: output buffer too small");
    if (!v.empty()) std::memcpy(out, v.data(), v.size() * sizeof(double));
    return (int)v.size();
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

}  // extern "C"

