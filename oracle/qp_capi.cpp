// oracle/qp_capi.cpp — TEST INFRASTRUCTURE: the mpc_qp_* entry points of include/mpc_qp_abi.h on the CPU (oracle/qp.hpp).
#include <cstring>
#include <stdexcept>
#include <string>
#include "qp.hpp"

struct mpc_qp_solver {
  mpc_qp_dims d;
  std::vector<double> x, y, z;  // previous solution (warm start)
  std::string err;
};

extern "C" {

int mpc_qp_create(const mpc_qp_dims* dims, mpc_qp_solver** out) {
  if (!dims || !out) return -2;
  if (dims->batch <= 0 || dims->n <= 0 || dims->neq < 0 || dims->nin < 0) return -2;
  mpc_qp_solver* s = new mpc_qp_solver();
  s->d = *dims;
  const size_t B = dims->batch, m = dims->nin + (dims->box ? dims->n : 0);
  s->x.assign(B * dims->n, 0.0); s->y.assign(B * dims->neq, 0.0); s->z.assign(B * m, 0.0);
  *out = s;
  return 0;
}
void mpc_qp_destroy(mpc_qp_solver* s) { delete s; }
const char* mpc_qp_last_error(mpc_qp_solver* s) { return s ? s->err.c_str() : "null handle"; }
void mpc_qp_default_settings(mpc_qp_settings* o) {
  o->eps_abs = 1e-5; o->rho = 1e-6; o->mu_eq = 1e-3; o->mu_in = 1e-1; o->mu_min_eq = 1e-9; o->mu_min_in = 1e-8;
  o->mu_update_factor = 0.1; o->alpha_bcl = 0.1; o->beta_bcl = 0.9; o->max_iter = 10000; o->max_iter_in = 1500; o->warm_start = 0; o->reserved = 0;
}

int mpc_qp_solve(mpc_qp_solver* s, const mpc_qp_settings* S, const double* H, const double* g, const double* A, const double* b,
                 const double* C, const double* l, const double* u, const double* l_box, const double* u_box,
                 double* x, double* y, double* z, double* z_box, mpc_qp_info* info) {
  if (!s) return -2;
  try {
    if (!S || !H || !g || !x || !info) throw std::runtime_error("qp_solve: null argument");
    const mpc_qp_dims& d = s->d;
    if (d.box && (!l_box || !u_box)) throw std::runtime_error("qp_solve: box bounds missing");
    const size_t n = d.n, neq = d.neq, nin = d.nin, m = nin + (d.box ? n : 0);
    for (int bi = 0; bi < d.batch; ++bi) {
      qp::Problem P;
      P.n = d.n; P.neq = d.neq; P.nin = d.nin; P.box = d.box;
      P.H = H + bi * n * n; P.g = g + bi * n; P.A = A ? A + bi * neq * n : nullptr; P.b = b ? b + bi * neq : nullptr;
      P.C = C ? C + bi * nin * n : nullptr; P.l = l ? l + bi * nin : nullptr; P.u = u ? u + bi * nin : nullptr;
      P.lb = d.box ? l_box + bi * n : nullptr; P.ub = d.box ? u_box + bi * n : nullptr;
      double* xs = s->x.data() + bi * n; double* ys = s->y.data() + bi * neq; double* zs = s->z.data() + bi * m;
      if (!S->warm_start) { std::fill(xs, xs + n, 0.0); std::fill(ys, ys + neq, 0.0); std::fill(zs, zs + m, 0.0); }
      qp::solve_one(P, *S, xs, ys, zs, info[bi]);
      std::memcpy(x + bi * n, xs, n * sizeof(double));
      if (y) std::memcpy(y + bi * neq, ys, neq * sizeof(double));
      if (z) std::memcpy(z + bi * nin, zs, nin * sizeof(double));
      if (z_box && d.box) std::memcpy(z_box + bi * n, zs + nin, n * sizeof(double));
    }
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

}  // extern "C"
