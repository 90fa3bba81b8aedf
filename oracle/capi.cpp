// oracle/capi.cpp — TEST INFRASTRUCTURE (CPU oracle). Not part of the product path.
// Exports the C-ABI of include/mpc_abi.h on top of the CPU restatement (oracle/solver.hpp).  Only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg load this library; the product
// (mpc_benchmark_amd) never does.  PARITY UNPINNED: the reference's arithmetic lives in un-vendored
// Aligator/Pinocchio (README.md:10-16) and no golden vectors exist.
#include <cstring>
#include <string>
#include "solver.hpp"
#include "../include/mpc_qp_abi.h"

using namespace orc;

// the swing-foot reference generator (mpc_walk_*): the functions both libraries compile (see the header)
#ifdef ORC_BACKEND_NAME   // built through oracle/cpu_port/capi_port.cpp: se3_math.h is there already, inside namespace cpu_port
#define WALKGEN_NS cpu_port
namespace cpu_port {
#define DEV static inline
#include "../mpc_benchmark_amd/csrc/walk_generator.h"
#undef DEV
}
#else
#define WALKGEN_NS walkgen
namespace walkgen {
#define DEV static inline
#include "../mpc_benchmark_amd/csrc/se3_math.h"
#include "../mpc_benchmark_amd/csrc/walk_generator.h"
#undef DEV
}
#endif

struct mpc_solver {
  Solver s;
  std::string err;
  bool perfect_feedback = false;
  bool appended_changed = false;  // the stage of the last mpc_cycle has another contact pattern than its predecessor (refine_appended_knot)
  int since_change = 1 << 20;     // cycles since the appended stage last changed its contact pattern
  bool appended_any = false;      // a stage was appended since the last run (refine_appended_knot < 0: refine after every cycle)
  bool isolate = false;        // mpc_set_failure_policy
  std::vector<int> failed;     // per instance: 0 or the failure code reported as mpc_stats.converged = -code
  // mpc_walk_*: raw model tables (the generator's forward kinematics reads them as the device kernel does), configuration, per-instance plan
  std::vector<int32_t> model_itab;
  std::vector<double> model_dtab;
  bool walk_on = false, walk_force_all = false;
  mpc_walk_config walk{};
  std::vector<double> walk_state;  // [B][48]: start_L | final_L | start_R | final_R
};

#define MPC_TRY(h, ...)                  \
  try {                                  \
    __VA_ARGS__;                         \
    return 0;                            \
  } catch (const std::exception& e) {    \
    if (h) (h)->err = e.what();          \
    return -1;                           \
  }

extern "C" {

int mpc_abi_version(void) { return MPC_ABI_VERSION; }
#ifndef ORC_BACKEND_NAME
#define ORC_BACKEND_NAME "oracle-cpu"
#endif
const char* mpc_backend_name(void) { return ORC_BACKEND_NAME; }

int mpc_create(const mpc_dims* dims, mpc_solver** out) {
  if (!dims || !out) return -2;
  mpc_solver* h = nullptr;
  try {
    h = new mpc_solver();
    h->s.init(*dims);
    *out = h;
    return 0;
  } catch (const std::exception&) {
    delete h;
    return -1;
  }
}
void mpc_destroy(mpc_solver* h) { delete h; }
const char* mpc_last_error(mpc_solver* h) { return h ? h->err.c_str() : "null handle"; }

int mpc_set_options(mpc_solver* h, const mpc_options* opt) { MPC_TRY(h, h->s.opt = *opt) }

int mpc_set_model(mpc_solver* h, const int32_t* itab, int32_t n_i, const double* dtab, int32_t n_d) {
  MPC_TRY(h, { h->s.model.parse(itab, n_i, dtab, n_d); h->s.have_model = true; h->model_itab.assign(itab, itab + n_i); h->model_dtab.assign(dtab, dtab + n_d); })
}

int mpc_set_stage(mpc_solver* h, int32_t k, const int32_t* desc, int32_t n_desc, const double* params, int32_t n_params) {
  MPC_TRY(h, {
    if (k < 0 || k > h->s.N()) throw std::runtime_error("stage index out of range");
    h->s.stages[k].parse(desc, n_desc, params, n_params);
    for (auto& ip : h->s.inst_params) ip[k] = h->s.stages[k].params;
  })
}

int mpc_update_stage_params(mpc_solver* h, int32_t k, int32_t offset, const double* vals, int32_t n) {
  MPC_TRY(h, {
    if (k < 0 || k > h->s.N()) throw std::runtime_error("stage index out of range");
    auto& p = h->s.stages[k].params;
    if (offset < 0 || offset + n > (int)p.size()) throw std::runtime_error("parameter update out of range");
    std::memcpy(p.data() + offset, vals, n * sizeof(double));
    for (auto& ip : h->s.inst_params) std::memcpy(ip[k].data() + offset, vals, n * sizeof(double));
  })
}

int mpc_update_stage_params_batch(mpc_solver* h, int32_t count, const int32_t* ks, const int32_t* offsets, const int32_t* lens,
                                  const double* vals) {
  MPC_TRY(h, {
    size_t pos = 0;
    for (int i = 0; i < count; ++i) {
      if (ks[i] < 0 || ks[i] > h->s.N()) throw std::runtime_error("stage index out of range");
      auto& p = h->s.stages[ks[i]].params;
      if (offsets[i] < 0 || lens[i] < 0 || offsets[i] + lens[i] > (int)p.size()) throw std::runtime_error("parameter update out of range");
      std::memcpy(p.data() + offsets[i], vals + pos, lens[i] * sizeof(double));
      for (auto& ip : h->s.inst_params) std::memcpy(ip[ks[i]].data() + offsets[i], vals + pos, lens[i] * sizeof(double));
      pos += lens[i];
    }
  })
}

int mpc_enable_instance_params(mpc_solver* h) {
  MPC_TRY(h, {
    Solver& s = h->s;
    if (!s.inst_params.empty()) return 0;
    s.inst_params.assign(s.dims.batch, std::vector<std::vector<double>>(s.N() + 1));
    for (auto& ip : s.inst_params) for (int k = 0; k <= s.N(); ++k) ip[k] = s.stages[k].params;
  })
}

int mpc_update_instance_params_batch(mpc_solver* h, int32_t count, const int32_t* insts, const int32_t* ks, const int32_t* offsets, const int32_t* lens, const double* vals) {
  MPC_TRY(h, {
    Solver& s = h->s;
    if (s.inst_params.empty()) throw std::runtime_error("update_instance_params: mpc_enable_instance_params first");
    size_t pos = 0;
    for (int i = 0; i < count; ++i) {
      if (insts[i] < 0 || insts[i] >= s.dims.batch) throw std::runtime_error("instance index out of range");
      if (ks[i] < 0 || ks[i] > s.N()) throw std::runtime_error("stage index out of range");
      auto& p = s.inst_params[insts[i]][ks[i]];
      if (offsets[i] < 0 || lens[i] < 0 || offsets[i] + lens[i] > (int)p.size()) throw std::runtime_error("parameter update out of range");
      std::memcpy(p.data() + offsets[i], vals + pos, lens[i] * sizeof(double));
      pos += lens[i];
    }
  })
}

int mpc_walk_init(mpc_solver* h, const mpc_walk_config* cfg) {
  MPC_TRY(h, {
    Solver& s = h->s;
    if (!cfg) throw std::runtime_error("walk_init: null configuration");
    if (s.inst_params.empty()) throw std::runtime_error("walk_init: mpc_enable_instance_params first");
    if (h->model_itab.empty() || s.dims.space != MPC_SPACE_MULTIBODY) throw std::runtime_error("walk_init: a whole-body model is needed (mpc_set_model)");
    const int nf = h->model_itab[3];
    if (cfg->frame_lf < 0 || cfg->frame_lf >= nf || cfg->frame_rf < 0 || cfg->frame_rf >= nf) throw std::runtime_error("walk_init: frame index out of range");
    // (the same range checks and messages as the HIP library: the offsets are used for 96-byte copies into the instance tables)
    for (int off : {cfg->off_lf, cfg->off_rf, cfg->toff_lf, cfg->toff_rf}) if (off >= 0 && off + 12 > s.dims.max_stage_doubles) throw std::runtime_error("walk_init: reference offset out of range");
    if ((cfg->toff_com >= 0 && cfg->toff_com + 3 > s.dims.max_stage_doubles) || (cfg->off_xref_z >= s.dims.max_stage_doubles)) throw std::runtime_error("walk_init: offset out of range");
    h->walk = *cfg;
    h->walk_state.assign((size_t)s.dims.batch * 48, 0.0);
    for (int b = 0; b < s.dims.batch; ++b) {
      double* st = h->walk_state.data() + (size_t)b * 48;
      std::memcpy(st, cfg->lf0, 96); std::memcpy(st + 12, cfg->lf0, 96); std::memcpy(st + 24, cfg->rf0, 96); std::memcpy(st + 36, cfg->rf0, 96);
    }
    h->walk_on = true;
  })
}

int mpc_walk_update(mpc_solver* h, int32_t takeoff_RF, int32_t takeoff_LF, int32_t land_RF, int32_t land_LF, const double* forward) {
  MPC_TRY(h, {
    Solver& s = h->s;
    if (!h->walk_on) throw std::runtime_error("walk_update: mpc_walk_init first");
    mpc_walk_config& c = h->walk;
    if (forward) { std::memcpy(c.t_left, forward, 24); std::memcpy(c.t_right, forward + 3, 24); c.swing_apex = forward[6]; }
    const int N = s.N();
    const bool replanning = land_LF < 0 || land_RF < 0 || (takeoff_RF >= 0 && takeoff_RF < c.T_ds) || (takeoff_LF >= 0 && takeoff_LF < c.T_ds);
    for (int b = 0; b < s.dims.batch; ++b) {
      double* st = h->walk_state.data() + (size_t)b * 48;
      if (replanning) {  // from the instance's own predicted next state xs[1]: what perfect-model feedback hands the next solve
        const double* q = s.inst[b].xs[1].data();
        WALKGEN_NS::M3 R; WALKGEN_NS::V3 p;
        double LF[12], RF[12];
        WALKGEN_NS::walk_frame_placement(h->model_itab.data(), h->model_dtab.data(), q, c.frame_lf, R, p); WALKGEN_NS::walk_pose_store(LF, R, p);
        WALKGEN_NS::walk_frame_placement(h->model_itab.data(), h->model_dtab.data(), q, c.frame_rf, R, p); WALKGEN_NS::walk_pose_store(RF, R, p);
        WALKGEN_NS::walk_plan(st, LF, RF, takeoff_RF, takeoff_LF, land_RF, land_LF, c.T_ds, c.t_left, c.t_right, c.rot_diff, c.floor_z);
      }
      double L[12], Rr[12];
      for (int j = (replanning || h->walk_force_all) ? 0 : N - 1; j < N; ++j) {
        WALKGEN_NS::walk_ref(L, st, st + 12, land_LF, j, c.T_ss, c.swing_apex);
        WALKGEN_NS::walk_ref(Rr, st + 24, st + 36, land_RF, j, c.T_ss, c.swing_apex);
        std::vector<double>& tab = s.inst_params[b][j];
        if (c.off_lf >= 0) std::memcpy(tab.data() + c.off_lf, L, 96);
        if (c.off_rf >= 0) std::memcpy(tab.data() + c.off_rf, Rr, 96);
        if (c.off_xref_z >= 0 && c.z_follow != 0.0) tab[c.off_xref_z] = c.xref_z0 + 0.5 * (L[11] + Rr[11]) - c.feet_z0;
      }
      // terminal node: the last foot references and the CoM target between them (L, Rr hold knot N - 1)
      std::vector<double>& tt = s.inst_params[b][N];
      if (c.toff_com >= 0) {
        tt[c.toff_com] = 0.5 * (L[9] + Rr[9]); tt[c.toff_com + 1] = 0.5 * (L[10] + Rr[10]);
        tt[c.toff_com + 2] = c.com0[2] + (c.z_follow != 0.0 ? 0.5 * (L[11] + Rr[11]) - c.feet_z0 : 0.0);
      }
      if (c.toff_lf >= 0) std::memcpy(tt.data() + c.toff_lf, L, 96);
      if (c.toff_rf >= 0) std::memcpy(tt.data() + c.toff_rf, Rr, 96);
    }
    h->walk_force_all = false;
  })
}

int mpc_walk_set_state(mpc_solver* h, const double* in) {
  MPC_TRY(h, {
    if (!h->walk_on || !in) throw std::runtime_error("walk_set_state: mpc_walk_init first");
    std::memcpy(h->walk_state.data(), in, h->walk_state.size() * sizeof(double));
    h->walk_force_all = true;
  })
}

int mpc_walk_get_state(mpc_solver* h, double* out) {
  MPC_TRY(h, {
    if (!h->walk_on || !out) throw std::runtime_error("walk_get_state: mpc_walk_init first");
    std::memcpy(out, h->walk_state.data(), h->walk_state.size() * sizeof(double));
  })
}

int mpc_cycle(mpc_solver* h, const int32_t* desc, int32_t n_desc, const double* params, int32_t n_params) {
  MPC_TRY(h, {
    const int N = h->s.N();
    StageDesc sd;
    sd.parse(desc, n_desc, params, n_params);
    // (refine_appended_knot: does the appended stage differ in its contact pattern — dynamics kind, contact list — from the one before it?)
    const StageDesc& last = h->s.stages[N - 1];
    h->appended_changed = sd.dyn != last.dyn || sd.ncontact != last.ncontact || sd.cid[0] != last.cid[0] || sd.cid[1] != last.cid[1];
    h->appended_any = true;
    h->since_change = h->appended_changed ? 0 : (h->since_change < (1 << 20) ? h->since_change + 1 : h->since_change);
    for (int k = 0; k + 1 < N; ++k) h->s.stages[k] = std::move(h->s.stages[k + 1]);
    h->s.stages[N - 1] = std::move(sd);
    for (auto& ip : h->s.inst_params) {  // the appended stage starts from the shared table in every instance
      for (int k = 0; k + 1 < N; ++k) ip[k] = std::move(ip[k + 1]);
      ip[N - 1] = h->s.stages[N - 1].params;
    }
  })
}

int mpc_set_x0(mpc_solver* h, const double* x0) {
  MPC_TRY(h, {
    const int nx = h->s.dims.nx;
    h->perfect_feedback = (x0 == nullptr);
    if (x0) for (int b = 0; b < h->s.dims.batch; ++b) h->s.inst[b].x0.assign(x0 + b * nx, x0 + (b + 1) * nx);
  })
}

int mpc_simulate_push(mpc_solver* h, int32_t substeps, double dt, const double* f_ext) {
  MPC_TRY(h, {
    if (substeps <= 0 || !(dt > 0.0)) throw std::runtime_error("simulate: substeps and dt must be positive");
    for (int b = 0; b < h->s.dims.batch; ++b) h->s.simulate(h->s.inst[b], substeps, dt, f_ext ? f_ext + 3 * b : nullptr);
    h->perfect_feedback = false;
  })
}

int mpc_simulate(mpc_solver* h, int32_t substeps, double dt) {
  MPC_TRY(h, {
    if (substeps <= 0 || !(dt > 0.0)) throw std::runtime_error("simulate: substeps and dt must be positive");
    for (int b = 0; b < h->s.dims.batch; ++b) h->s.simulate(h->s.inst[b], substeps, dt);
    h->perfect_feedback = false;
  })
}
int mpc_simulate_torque(mpc_solver* h, const double* x, const double* tau, int32_t substeps, double dt, double* wrenches) {
  MPC_TRY(h, {
    if (substeps <= 0 || !(dt > 0.0)) throw std::runtime_error("simulate: substeps and dt must be positive");
    if (!tau) throw std::runtime_error("simulate_torque: tau must not be null");
    const int nx = h->s.dims.nx, nu = h->s.dims.nu;
    for (int b = 0; b < h->s.dims.batch; ++b)
      h->s.simulate_torque(h->s.inst[b], x ? x + (size_t)b * nx : nullptr, tau + (size_t)b * nu, substeps, dt, wrenches ? wrenches + 12 * b : nullptr);
    h->perfect_feedback = false;
  })
}
// include/mpc_qp_abi.h mpc_qp_low_level_steps: the low-level loop of kinodynamic_talos.py:411-462 as a plain host loop over this library's own
// entry points (the HIP library does the same with two small kernels between the stages, csrc/pipeline_glue.h)
extern "C" void orc_qp_set_error(mpc_qp_solver* s, const char* what);  // qp_capi.cpp
int mpc_qp_low_level_steps(mpc_qp_solver* qp, const mpc_qp_settings* S, mpc_solver* plan, mpc_solver* sim, int32_t nk, const int32_t* frames,
                           const double* weights, const double* cone, double kd, const int32_t* contact_states, const double* tau_max,
                           const double* x, int32_t steps, double dt, double* x_prev, double* x_out, double* tau, double* forces, mpc_qp_info* info) {
  if (!qp) return -2;
  try {
    if (!S || !plan || !sim || !contact_states || !tau_max || !frames || !weights || !cone) throw std::runtime_error("qp_low_level_steps: null argument");
    if (steps <= 0 || !(dt > 0.0)) throw std::runtime_error("qp_low_level_steps: steps and dt must be positive");
    const Solver& ps = plan->s;
    Solver& zs = sim->s;
    const Model& m = ps.model;
    const int B = ps.dims.batch, nq = m.nq, nv = m.nv, nx = nq + nv, n = 2 * nv, nu = nv - 6, nf = 6 * nk, mu = ps.dims.nu, qn = 2 * nv - 6 + nf;
    if (zs.dims.batch != B) throw std::runtime_error("qp_low_level_steps: the three handles must have the same batch size");
    if (ps.dims.space != MPC_SPACE_MULTIBODY || ps.dims.nx != nx || ps.dims.ndx != n || mu != nf + nu)
      throw std::runtime_error("qp_low_level_steps: the plan must be a multibody problem with nx = nq + nv and controls (6 nk contact wrench components, nv - 6 joint accelerations)");
    if (zs.dims.space != MPC_SPACE_MULTIBODY || zs.dims.nx != nx || zs.dims.nu != nu)
      throw std::runtime_error("qp_low_level_steps: the simulator handle must hold whole-body contact dynamics with nu = nv - 6 (the handle of mpc_simulate_torque)");
    std::vector<double> xm((size_t)B * nx), a0((size_t)B * nv), f0((size_t)B * nf), sol((size_t)B * qn), tq((size_t)B * nu), fn((size_t)B * nf), d(n);
    std::vector<mpc_qp_info> inf(B);
    if (x) std::memcpy(xm.data(), x, xm.size() * sizeof(double));
    else for (int b = 0; b < B; ++b) std::memcpy(xm.data() + (size_t)b * nx, zs.inst[b].x0.data(), nx * sizeof(double));
    for (int step = 0; step < steps; ++step) {
      if (step == steps - 1 && x_prev) std::memcpy(x_prev, xm.data(), xm.size() * sizeof(double));
      for (int b = 0; b < B; ++b) {
        const Instance& in = ps.inst[b];
        mb_difference(m, xm.data() + (size_t)b * nx, in.xs[0].data(), d.data());  // difference(x_measured, xs[0])
        const std::vector<double>& K0 = in.gains[0].K;
        for (int i = 0; i < mu; ++i) {
          double su = 0.0;
          for (int j = 0; j < n; ++j) su += K0[(size_t)i * n + j] * d[j];
          su = in.us[0][i] - su;
          if (i < nf) f0[(size_t)b * nf + i] = su; else a0[(size_t)b * nv + 6 + (i - nf)] = su;
        }
        for (int i = 0; i < 6; ++i) a0[(size_t)b * nv + i] = in.knots[0].xdot[nv + i];
      }
      if (mpc_qp_solve_id(qp, S, nk, frames, weights, cone, kd, xm.data(), a0.data(), f0.data(), contact_states, sol.data(), nullptr, nullptr, inf.data(),
                          nullptr, nullptr, nullptr, nullptr) != 0)
        return -1;  // (the QP handle holds the message)
      for (int b = 0; b < B; ++b) {
        for (int i = 0; i < nu; ++i) tq[(size_t)b * nu + i] = std::fmin(std::fmax(sol[(size_t)b * qn + nv + nf + i], -tau_max[i]), tau_max[i]);
        for (int i = 0; i < nf; ++i) fn[(size_t)b * nf + i] = f0[(size_t)b * nf + i] + sol[(size_t)b * qn + nv + i];
        zs.simulate_torque(zs.inst[b], xm.data() + (size_t)b * nx, tq.data() + (size_t)b * nu, 1, dt, nullptr);
        std::memcpy(xm.data() + (size_t)b * nx, zs.inst[b].x0.data(), nx * sizeof(double));
      }
    }
    sim->perfect_feedback = false;
    if (x_out) std::memcpy(x_out, xm.data(), xm.size() * sizeof(double));
    if (tau) std::memcpy(tau, tq.data(), tq.size() * sizeof(double));
    if (forces) std::memcpy(forces, fn.data(), fn.size() * sizeof(double));
    if (info) std::memcpy(info, inf.data(), inf.size() * sizeof(mpc_qp_info));
    return 0;
  } catch (const std::exception& e) {
    orc_qp_set_error(qp, e.what());
    return -1;
  }
}

int mpc_get_x0(mpc_solver* h, double* x0) {
  MPC_TRY(h, {
    const int nx = h->s.dims.nx;
    for (int b = 0; b < h->s.dims.batch; ++b) std::memcpy(x0 + (size_t)b * nx, h->s.inst[b].x0.data(), nx * sizeof(double));
  })
}

int mpc_profile(mpc_solver*, int32_t) { return 0; }
int mpc_profile_read(mpc_solver*, int32_t, char*, int32_t, int32_t*, double*) { return 0; }
int mpc_kernel_info(mpc_solver*, int32_t, char*, int32_t, int32_t*) { return 0; }

int mpc_setup(mpc_solver* h) { MPC_TRY(h, h->s.setup()) }

static void run_all(mpc_solver* h, mpc_stats* stats) {
  Solver& s = h->s;
  // mpc_options.corrector_window (csrc/mpc_hip.hip corrector_armed: the same rule)
  s.corrector_armed = s.opt.corrector_prim_tol > 0.0 && (s.opt.corrector_window <= 0 || h->since_change < s.opt.corrector_window);
  h->failed.resize(s.dims.batch, 0);
  for (int b = 0; b < s.dims.batch; ++b) {
    if (h->isolate && h->failed[b]) {  // sits this run out until mpc_revive_instance
      s.inst[b].stats.num_iters = 0; s.inst[b].stats.converged = -h->failed[b];
    } else if (!h->isolate) {
      s.run_instance(s.inst[b]);
    } else {
      try {
        s.run_instance(s.inst[b]);
      } catch (const std::exception& e) {  // the codes of the HIP library: 2 / 3 / 4 the blocks of the Riccati step, 5 anything else
        const std::string what = e.what();
        h->failed[b] = what.find("mu_dyn P") != std::string::npos ? 2 : what.find("control Hessian") != std::string::npos ? 3 : what.find("Schur") != std::string::npos ? 4 : 5;
        s.inst[b].stats.converged = -h->failed[b];
      }
    }
    if (stats) stats[b] = s.inst[b].stats;
  }
}

int mpc_run(mpc_solver* h, const double* xs, const double* us, mpc_stats* stats) {
  MPC_TRY(h, {
    h->appended_changed = false; h->appended_any = false;  // (an uploaded warm start is the caller's: it stays as it is)
    Solver& s = h->s;
    const int N = s.N(), nx = s.dims.nx, nu = s.dims.nu;
    for (int b = 0; b < s.dims.batch; ++b) {
      Instance& in = s.inst[b];
      for (int k = 0; k <= N; ++k) in.xs[k].assign(xs + ((size_t)b * (N + 1) + k) * nx, xs + ((size_t)b * (N + 1) + k + 1) * nx);
      for (int k = 0; k < N; ++k) in.us[k].assign(us + ((size_t)b * N + k) * nu, us + ((size_t)b * N + k + 1) * nu);
    }
    run_all(h, stats);
  })
}

int mpc_run_shifted(mpc_solver* h, mpc_stats* stats) {
  MPC_TRY(h, {
    Solver& s = h->s;
    const int N = s.N();
    for (int b = 0; b < s.dims.batch; ++b) {
      Instance& in = s.inst[b];
      for (int k = 0; k < N; ++k) in.xs[k] = in.xs[k + 1];
      for (int k = 0; k + 1 < N; ++k) in.us[k] = in.us[k + 1];
      if (h->perfect_feedback) in.x0 = in.xs[0];  // predicted next state becomes the measurement
      in.xs[0] = in.x0;
      // (R > 0: when the appended stage changed its contact pattern ; R < 0: after every cycle, |R| steps)
      if (((s.opt.refine_appended_knot > 0 && h->appended_changed) || (s.opt.refine_appended_knot < 0 && h->appended_any)) && in.stats.converged >= 0) s.refine_appended_knot(in);
    }
    h->appended_changed = false; h->appended_any = false;
    run_all(h, stats);
  })
}

int mpc_run_shifted_async(mpc_solver* h) { return mpc_run_shifted(h, nullptr); }
int mpc_set_tick_reuse(mpc_solver*, int32_t) { return 0; }
int mpc_poll(mpc_solver*, int32_t* in_flight, int32_t* completed) { if (in_flight) *in_flight = 0; if (completed) *completed = 0; return 0; }  // ticks run inside the call  // an optimisation of the HIP library only
int mpc_wait(mpc_solver* h, mpc_stats* stats) {
  MPC_TRY(h, { if (stats) for (int b = 0; b < h->s.dims.batch; ++b) stats[b] = h->s.inst[b].stats; })
}

int mpc_get_gain(mpc_solver* h, int32_t k, double* K_k, double* kff_k) {
  MPC_TRY(h, {
    Solver& s = h->s;
    const int nu = s.dims.nu, n = s.dims.ndx;
    if (k < 0 || k >= s.N()) throw std::runtime_error("get_gain: knot index out of range");
    for (int b = 0; b < s.dims.batch; ++b) {
      const Gains& g = s.inst[b].gains[k];
      if (K_k) { std::fill(K_k + (size_t)b * nu * n, K_k + (size_t)(b + 1) * nu * n, 0.0); std::memcpy(K_k + (size_t)b * nu * n, g.K.data(), std::min(g.K.size(), (size_t)nu * n) * sizeof(double)); }
      if (kff_k) { std::fill(kff_k + (size_t)b * nu, kff_k + (size_t)(b + 1) * nu, 0.0); std::memcpy(kff_k + (size_t)b * nu, g.kff.data(), std::min(g.kff.size(), (size_t)nu) * sizeof(double)); }
    }
  })
}

int mpc_wait_state(mpc_solver* h, mpc_stats* stats, double* x_next) {
  MPC_TRY(h, {
    Solver& s = h->s;
    if (stats) for (int b = 0; b < s.dims.batch; ++b) stats[b] = s.inst[b].stats;
    if (x_next) for (int b = 0; b < s.dims.batch; ++b) std::memcpy(x_next + (size_t)b * s.dims.nx, s.inst[b].xs[1].data(), s.dims.nx * sizeof(double));
  })
}

// ---- solver-state checkpoint (layout shared with mpc_benchmark_amd/csrc/mpc_hip.hip) --------------------------------------------------
#define MPC_STATE_MAGIC 20250304.0
#define MPC_STATE_HEADER 16
static int64_t state_doubles(const Solver& s) {
  const mpc_dims& d = s.dims;
  const int64_t N1 = d.horizon + 1;
  return MPC_STATE_HEADER + N1 * (2 + (int64_t)d.max_stage_ints + d.max_stage_doubles) +
         (int64_t)d.batch * (N1 * d.nx + (int64_t)d.horizon * d.nu + N1 * d.nc_max + N1 * d.ndx + d.nx + 4);
}
int64_t mpc_state_size(mpc_solver* h) { return h ? state_doubles(h->s) : -1; }

int64_t mpc_get_state(mpc_solver* h, double* buf, int64_t cap) {
  if (!h) return -2;
  try {
    Solver& s = h->s;
    const mpc_dims& d = s.dims;
    const int N = s.N();
    const int64_t need = state_doubles(s);
    if (!buf || cap < need) throw std::runtime_error("get_state: buffer too small (mpc_state_size doubles needed)");
    std::fill(buf, buf + need, 0.0);
    double* o = buf;
    const double hdr[MPC_STATE_HEADER] = {MPC_STATE_MAGIC, (double)d.batch, (double)N, (double)d.nx, (double)d.ndx, (double)d.nu, (double)d.nc_max, (double)d.space,
                                          h->perfect_feedback ? 1.0 : 0.0, (double)d.max_stage_ints, (double)d.max_stage_doubles, (double)(h->since_change + 1) /* 0: a state saved before the field existed */, 0, 0, 0, 0};
    std::memcpy(o, hdr, sizeof(hdr)); o += MPC_STATE_HEADER;
    for (int k = 0; k <= N; ++k) {
      const StageDesc& sd = s.stages[k];
      const int nd = MPC_STAGE_HEADER_WORDS + MPC_TERM_WORDS * (int)sd.terms.size();
      if (nd > d.max_stage_ints || (int)sd.params.size() > d.max_stage_doubles) throw std::runtime_error("get_state: stage table exceeds the capacity of mpc_create");
      o[0] = nd; o[1] = (double)sd.params.size();
      double* di = o + 2;
      di[0] = sd.dyn; di[1] = sd.ncontact; di[2] = sd.cid[0]; di[3] = sd.cid[1]; di[4] = sd.dyn_poff; di[5] = (double)sd.terms.size(); di[6] = sd.nc; di[7] = 0;
      for (size_t t = 0; t < sd.terms.size(); ++t) {
        const Term& tr = sd.terms[t];
        const double w[8] = {(double)tr.type, (double)tr.role, (double)tr.dim, (double)tr.i0, (double)tr.i1, (double)tr.poff, (double)tr.woff, (double)tr.flags};
        std::memcpy(di + MPC_STAGE_HEADER_WORDS + MPC_TERM_WORDS * t, w, sizeof(w));
      }
      std::memcpy(o + 2 + d.max_stage_ints, sd.params.data(), sd.params.size() * sizeof(double));
      o += 2 + d.max_stage_ints + d.max_stage_doubles;
    }
    const int B = d.batch;
    for (int b = 0; b < B; ++b) for (int k = 0; k <= N; ++k) { std::memcpy(o, s.inst[b].xs[k].data(), d.nx * sizeof(double)); o += d.nx; }
    for (int b = 0; b < B; ++b) for (int k = 0; k < N; ++k) { std::memcpy(o, s.inst[b].us[k].data(), d.nu * sizeof(double)); o += d.nu; }
    for (int b = 0; b < B; ++b) for (int k = 0; k <= N; ++k) { std::memcpy(o, s.inst[b].vs[k].data(), std::min<size_t>(d.nc_max, s.inst[b].vs[k].size()) * sizeof(double)); o += d.nc_max; }
    for (int b = 0; b < B; ++b) for (int k = 0; k <= N; ++k) { std::memcpy(o, s.inst[b].lams[k].data(), d.ndx * sizeof(double)); o += d.ndx; }
    for (int b = 0; b < B; ++b) { std::memcpy(o, s.inst[b].x0.data(), d.nx * sizeof(double)); o += d.nx; }
    for (int b = 0; b < B; ++b) { *o++ = s.inst[b].mu; *o++ = s.inst[b].inner_tol; *o++ = s.inst[b].prim_tol; *o++ = 0.0; }
    return need;
  } catch (const std::exception& e) { h->err = e.what(); return -1; }
}

int mpc_set_state(mpc_solver* h, const double* buf, int64_t len) {
  MPC_TRY(h, {
    Solver& s = h->s;
    const mpc_dims& d = s.dims;
    const int N = s.N(), B = d.batch;
    if (!buf || len < state_doubles(s)) throw std::runtime_error("set_state: truncated state");
    const double* o = buf;
    if (o[0] != MPC_STATE_MAGIC || (int)o[1] != B || (int)o[2] != N || (int)o[3] != d.nx || (int)o[4] != d.ndx || (int)o[5] != d.nu || (int)o[6] != d.nc_max ||
        (int)o[7] != d.space || (int)o[9] != d.max_stage_ints || (int)o[10] != d.max_stage_doubles)
      throw std::runtime_error("set_state: the state was saved by a handle of other dimensions");
    h->perfect_feedback = o[8] != 0.0;
    h->since_change = (int)o[11] > 0 ? (int)o[11] - 1 : 1 << 20;
    o += MPC_STATE_HEADER;
    std::vector<int32_t> desc(d.max_stage_ints);
    for (int k = 0; k <= N; ++k) {
      const int nd = (int)o[0], np = (int)o[1];
      for (int i = 0; i < d.max_stage_ints; ++i) desc[i] = (int32_t)o[2 + i];
      if (nd > 0) {
        s.stages[k].parse(desc.data(), nd, o + 2 + d.max_stage_ints, np);
        for (auto& ip : s.inst_params) ip[k] = s.stages[k].params;  // (as mpc_set_stage: the instances start from the shared table again)
      }
      o += 2 + d.max_stage_ints + d.max_stage_doubles;
    }
    h->walk_force_all = h->walk_on;  // the next mpc_walk_update rewrites the references of every knot
    for (int b = 0; b < B; ++b) for (int k = 0; k <= N; ++k) { s.inst[b].xs[k].assign(o, o + d.nx); o += d.nx; }
    for (int b = 0; b < B; ++b) for (int k = 0; k < N; ++k) { s.inst[b].us[k].assign(o, o + d.nu); o += d.nu; }
    for (int b = 0; b < B; ++b) for (int k = 0; k <= N; ++k) { std::copy(o, o + std::min<size_t>(d.nc_max, s.inst[b].vs[k].size()), s.inst[b].vs[k].begin()); o += d.nc_max; }
    for (int b = 0; b < B; ++b) for (int k = 0; k <= N; ++k) { s.inst[b].lams[k].assign(o, o + d.ndx); o += d.ndx; }
    for (int b = 0; b < B; ++b) { s.inst[b].x0.assign(o, o + d.nx); o += d.nx; }
    for (int b = 0; b < B; ++b) { s.inst[b].mu = o[0]; s.inst[b].inner_tol = o[1]; s.inst[b].prim_tol = o[2]; o += 4; s.inst[b].tree_guess_valid = false; }
  })
}

int mpc_get_results(mpc_solver* h, double* xs, double* us, double* K, double* kff, double* vs, double* lams) {
  MPC_TRY(h, {
    Solver& s = h->s;
    const int N = s.N(), nx = s.dims.nx, nu = s.dims.nu, n = s.dims.ndx, nc = s.dims.nc_max;
    for (int b = 0; b < s.dims.batch; ++b) {
      const Instance& in = s.inst[b];
      for (int k = 0; k <= N; ++k) {
        if (xs) std::memcpy(xs + ((size_t)b * (N + 1) + k) * nx, in.xs[k].data(), nx * sizeof(double));
        if (vs) std::memcpy(vs + ((size_t)b * (N + 1) + k) * nc, in.vs[k].data(), nc * sizeof(double));
        if (lams) std::memcpy(lams + ((size_t)b * (N + 1) + k) * n, in.lams[k].data(), n * sizeof(double));
      }
      for (int k = 0; k < N; ++k) {
        if (us) std::memcpy(us + ((size_t)b * N + k) * nu, in.us[k].data(), nu * sizeof(double));
        if (K) {
          double* dst = K + ((size_t)b * N + k) * nu * n;
          if ((int)in.gains[k].K.size() == nu * n) std::memcpy(dst, in.gains[k].K.data(), nu * n * sizeof(double));
          else std::memset(dst, 0, nu * n * sizeof(double));
        }
        if (kff) {
          double* dst = kff + ((size_t)b * N + k) * nu;
          if ((int)in.gains[k].kff.size() == nu) std::memcpy(dst, in.gains[k].kff.data(), nu * sizeof(double));
          else std::memset(dst, 0, nu * sizeof(double));
        }
      }
    }
  })
}

int mpc_get_stage_data(mpc_solver* h, int32_t k, double* xdot, double* wrenches) {
  MPC_TRY(h, {
    Solver& s = h->s;
    if (k < 0 || k >= s.N()) throw std::runtime_error("stage index out of range");
    for (int b = 0; b < s.dims.batch; ++b) {
      const Knot& kn = s.inst[b].knots[k];
      if (xdot) for (int i = 0; i < s.dims.ndx; ++i) xdot[b * s.dims.ndx + i] = i < (int)kn.xdot.size() ? kn.xdot[i] : 0.0;
      if (wrenches) std::memcpy(wrenches + b * 12, kn.wrench, 12 * sizeof(double));
    }
  })
}

int mpc_set_failure_policy(mpc_solver* h, int32_t isolate) {
  MPC_TRY(h, { h->isolate = isolate != 0; })
}

int mpc_revive_instance(mpc_solver* h, int32_t dst, int32_t src) {
  MPC_TRY(h, {
    Solver& s = h->s;
    if (dst < 0 || dst >= s.dims.batch || src < 0 || src >= s.dims.batch || src == dst) throw std::runtime_error("revive_instance: instance index out of range");
    s.inst[dst] = s.inst[src];
    h->failed.resize(s.dims.batch, 0);
    h->failed[dst] = 0;
  })
}

int mpc_debug_evaluate(mpc_solver* h, const double* xs, const double* us) {
  MPC_TRY(h, {
    Solver& s = h->s;
    const int N = s.N(), nx = s.dims.nx, nu = s.dims.nu;
    for (int b = 0; b < s.dims.batch; ++b) {
      Instance& in = s.inst[b];
      for (int k = 0; k <= N; ++k) in.xs[k].assign(xs + ((size_t)b * (N + 1) + k) * nx, xs + ((size_t)b * (N + 1) + k + 1) * nx);
      for (int k = 0; k < N; ++k) in.us[k].assign(us + ((size_t)b * N + k) * nu, us + ((size_t)b * N + k + 1) * nu);
      s.evaluate(in, in.xs, in.us, in.knots, true);
    }
  })
}

int mpc_debug_get(mpc_solver* h, const char* name, int32_t b, int32_t k, double* out, int32_t cap) {
  if (!h) return -2;
  try {
    Solver& s = h->s;
    if (b < 0 || b >= s.dims.batch || k < 0 || k > s.N()) throw std::runtime_error("debug_get: index out of range");
    const Instance& in = s.inst[b];
    const Knot& kn = in.knots[k];
    const Gains& g = in.gains[k];
    const std::string nm(name);
    const std::vector<double>* v = nullptr;
    std::vector<double> tmp;
    if (nm == "inst_params") {  // the parameter table instance b uses at knot k (its own copy after mpc_enable_instance_params, else the shared one)
      tmp = s.inst_params.empty() ? s.stages[k].params : s.inst_params[b][k]; v = &tmp;
    }
    else if (nm == "H") v = &kn.H;
    else if (nm == "grad") v = &kn.grad;
    else if (nm == "AB") v = &kn.AB;
    else if (nm == "f") v = &kn.f;
    else if (nm == "E6") { tmp.assign(kn.E6, kn.E6 + 36); v = &tmp; }
    else if (nm == "cval") v = &kn.cval;
    else if (nm == "CD") v = &kn.CD;
    else if (nm == "cost") { tmp.assign(1, kn.cost); v = &tmp; }
    else if (nm == "xnext") v = &kn.xnext;
    else if (nm == "xdot") v = &kn.xdot;
    else if (nm == "wrench") { tmp.assign(kn.wrench, kn.wrench + 12); v = &tmp; }
    else if (nm == "P") v = &g.P;
    else if (nm == "p") v = &g.p;
    else if (nm == "K") v = &g.K;
    else if (nm == "kff") v = &g.kff;
    else if (nm == "Knu") v = &g.Knu;
    else if (nm == "knu") v = &g.knu;
    else if (nm == "Kexact") v = &g.Kexact;
    else if (nm == "Lm") v = &g.Lm;
    else if (nm == "Sg") v = &g.Sg;
    else if (nm == "sg") v = &g.sg;
    else if (nm == "Kth") v = &g.Kth;
    else if (nm == "Knuth") v = &g.Knuth;
    else if (nm == "Mth") v = &g.Mth;
    else if (nm == "Pt") v = &g.Pt;
    else if (nm == "mx0") v = &g.mx0;
    else if (nm == "p0") v = &g.p0;
    else if (nm == "kff0") v = &g.kff0;
    else if (nm == "Mu") v = &g.Mu;
    else if (nm == "Znu") v = &g.Znu;
    else if (nm == "T6") { tmp.assign(g.T6, g.T6 + 36); v = &tmp; }
    else if (nm == "Mx") v = &g.Mx;
    else if (nm == "mx") v = &g.mx;
    else if (nm == "Zx" || nm == "zc" || nm == "calP" || nm == "calp" || nm == "theta") {  // k = index of the parametric leg
      const auto& src = nm == "Zx" ? in.leg_Zx : nm == "zc" ? in.leg_zc : nm == "calP" ? in.leg_calP : nm == "calp" ? in.leg_calp : in.leg_theta;
      if (k >= (int)src.size()) throw std::runtime_error("debug_get: no such leg record");
      v = &src[k];
    }
    else if (nm == "dx") v = &in.dxs[k];
    else if (nm == "du") { if (k >= s.N()) throw std::runtime_error("no du at the terminal knot"); v = &in.dus[k]; }
    else if (nm == "dvs") v = &in.dvs[k];
    else if (nm == "dlams") v = &in.dlams[k];
    else throw std::runtime_error("debug_get: unknown quantity " + nm);
    const int cnt = (int)v->size();
    if (cnt > cap) throw std::runtime_error("debug_get: output buffer too small");
    if (cnt > 0) std::memcpy(out, v->data(), cnt * sizeof(double));  // (an empty quantity: data() may be null)
    return cnt;
  } catch (const std::exception& e) {
    h->err = e.what();
    return -1;
  }
}

}  // extern "C"
