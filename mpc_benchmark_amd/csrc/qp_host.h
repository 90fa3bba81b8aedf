// qp_host.h — host side of the batched QP solver (include/mpc_qp_abi.h) for the HIP library; included by mpc_hip.hip.
#pragma once
#include <atomic>
#include "qp_kernel.h"
#include "qp_assemble.h"
#include "qp_device_api.h"

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device, per-kernel setting shared by all handles: keep the LARGEST request so
// far (a second handle with a smaller problem must not lower the limit under a handle that is still in use)
#include <map>
#include <mutex>
static void qp_lds_attr_max(const void* fn, int bytes) {
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

struct mpc_qp_solver {
  mpc_qp_dims d{};
  hipStream_t stream = nullptr;
  QpLds lds{};
  double *dH = nullptr, *dg = nullptr, *dA = nullptr, *db = nullptr, *dC = nullptr, *dl = nullptr, *du = nullptr, *dlb = nullptr, *dub = nullptr;
  double *dx = nullptr, *dy = nullptr, *dz = nullptr;
  mpc_qp_info* dinfo = nullptr;
  // on-device assembly of the inverse-dynamics QP (mpc_qp_set_model / mpc_qp_solve_id)
  int32_t* d_mi = nullptr; double* d_md = nullptr;
  int m_nj = 0, m_nq = 0, m_nv = 0, m_nframes = 0;
  double *d_xrob = nullptr, *d_acc = nullptr, *d_f = nullptr, *d_cone = nullptr;
  int32_t *d_cs = nullptr, *d_frames = nullptr;
  int id_nk = 0;
  bool id_const_uploaded = false;
  double *d_ik = nullptr, *d_gains = nullptr, *d_w = nullptr;  // mpc_qp_solve_ikid
  int ikid_nk = 0;
  std::vector<double> ikid_const;
  std::vector<double> id_const;  // weights[2], cone[54], frames[nk] as last uploaded
  double* d_glue = nullptr; size_t glue_cap = 0;  // scratch of mpc_qp_low_level_steps (pipeline_glue.h)
  std::vector<void*> allocs;
  std::string err;
  template <class T> T* alloc(size_t count) {
    void* p = nullptr;
    HIP_OK(hipMalloc(&p, (count ? count : 1) * sizeof(T)));
    HIP_OK(hipMemsetAsync(p, 0, (count ? count : 1) * sizeof(T), stream));
    allocs.push_back(p);
    return (T*)p;
  }
};

// launch k_qp_solve on the handle's device buffers
static void qp_launch(mpc_qp_solver* s, const mpc_qp_settings* S) {
  const mpc_qp_dims& d = s->d;
  QpArgs a;
  a.d = d; a.S = *S;
  a.H = s->dH; a.g = s->dg; a.A = s->dA; a.b = s->db; a.C = s->dC; a.l = s->dl; a.u = s->du; a.lb = s->dlb; a.ub = s->dub;
  a.x = s->dx; a.y = s->dy; a.z = s->dz; a.info = s->dinfo; a.lds = s->lds;
  const dim3 grid(d.batch), blk(QP_THREADS);
  if (s->lds.mf && s->lds.mats == 1) hipLaunchKernelGGL((k_qp_solve<1, true>), grid, blk, s->lds.total_bytes, s->stream, a);
  else if (s->lds.mf && s->lds.mats == 2) hipLaunchKernelGGL((k_qp_solve<2, true>), grid, blk, s->lds.total_bytes, s->stream, a);
  else if (s->lds.mf) hipLaunchKernelGGL((k_qp_solve<0, true>), grid, blk, s->lds.total_bytes, s->stream, a);
  else if (s->lds.mats) hipLaunchKernelGGL((k_qp_solve<1, false>), grid, blk, s->lds.total_bytes, s->stream, a);
  else hipLaunchKernelGGL((k_qp_solve<0, false>), grid, blk, s->lds.total_bytes, s->stream, a);
  HIP_OK(hipGetLastError());
}

// ... and bring the solution back
static void qp_launch_and_fetch(mpc_qp_solver* s, const mpc_qp_settings* S, double* x, double* y, double* z, double* z_box, mpc_qp_info* info) {
  const mpc_qp_dims& d = s->d;
  const size_t B = d.batch, n = d.n, neq = d.neq, nin = d.nin, m = nin + (d.box ? n : 0);
  qp_launch(s, S);
  HIP_OK(hipMemcpyAsync(x, s->dx, B * n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  if (y && neq) HIP_OK(hipMemcpyAsync(y, s->dy, B * neq * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  std::vector<double> zh(B * m);
  if (m) HIP_OK(hipMemcpyAsync(zh.data(), s->dz, B * m * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIP_OK(hipMemcpyAsync(info, s->dinfo, B * sizeof(mpc_qp_info), hipMemcpyDeviceToHost, s->stream));
  HIP_OK(hipStreamSynchronize(s->stream));
  for (size_t bi = 0; bi < B; ++bi) {
    if (z && nin) std::memcpy(z + bi * nin, zh.data() + bi * m, nin * sizeof(double));
    if (z_box && d.box) std::memcpy(z_box + bi * n, zh.data() + bi * m + nin, n * sizeof(double));
  }
}

extern "C" {

int mpc_qp_create(const mpc_qp_dims* dims, mpc_qp_solver** out) {
  if (!dims || !out) return -2;
  if (dims->batch <= 0 || dims->n <= 0 || dims->neq < 0 || dims->nin < 0) return -2;
  mpc_qp_solver* s = new mpc_qp_solver();
  try {
    s->d = *dims;
    HIP_OK(hipSetDevice(dims->device));
    HIP_OK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    const size_t B = dims->batch, n = dims->n, neq = dims->neq, nin = dims->nin, m = nin + (dims->box ? n : 0);
    // matrices in LDS shorten one QP's dependent chain but halve the workgroups a CU can hold: only while every QP has a CU to itself
    s->lds = make_qp_lds(dims->n, dims->neq, dims->nin, (int)m, dims->batch <= 256, getenv("MPC_QP_NO_MFMA") == nullptr);  // (MPC_QP_NO_MFMA: the column-by-column form, developer switch)
    if (dims->n > 128) throw std::runtime_error("QP too large: more than 128 unknowns (the H mat-vec of the kernel takes two columns per lane)");
    if (s->lds.total_bytes > 160 * 1024) throw std::runtime_error("QP too large for the LDS of one workgroup (n (n + 1) + n neq + neq (neq + 1) doubles)");
    s->dH = s->alloc<double>(B * n * n); s->dg = s->alloc<double>(B * n); s->dA = s->alloc<double>(B * neq * n); s->db = s->alloc<double>(B * neq);
    s->dC = s->alloc<double>(B * nin * n); s->dl = s->alloc<double>(B * nin); s->du = s->alloc<double>(B * nin);
    s->dlb = s->alloc<double>(B * n); s->dub = s->alloc<double>(B * n);
    s->dx = s->alloc<double>(B * n); s->dy = s->alloc<double>(B * neq); s->dz = s->alloc<double>(B * m);
    s->dinfo = s->alloc<mpc_qp_info>(B);
    for (const void* fn : {(const void*)k_qp_solve<0, false>, (const void*)k_qp_solve<1, false>, (const void*)k_qp_solve<0, true>, (const void*)k_qp_solve<1, true>,
                           (const void*)k_qp_solve<2, true>})
      qp_lds_attr_max(fn, s->lds.total_bytes);  // (the kernels keep a static int beside the dynamic carve-out)
    HIP_OK(hipStreamSynchronize(s->stream));
  } catch (const std::exception& e) {
    fprintf(stderr, "mpc_qp_create: %s\n", e.what());
    for (void* p : s->allocs) (void)hipFree(p);
    delete s;
    return -1;
  }
  *out = s;
  return 0;
}

void mpc_qp_destroy(mpc_qp_solver* s) {
  if (!s) return;
  (void)hipStreamSynchronize(s->stream);
  for (void* p : s->allocs) (void)hipFree(p);
  (void)hipStreamDestroy(s->stream);
  delete s;
}
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
    if ((d.neq && (!A || !b)) || (d.nin && (!C || !l || !u))) throw std::runtime_error("qp_solve: constraint data missing");
    const size_t B = d.batch, n = d.n, neq = d.neq, nin = d.nin, m = nin + (d.box ? n : 0);
    auto up = [&](double* dst, const double* src, size_t cnt) { if (cnt) HIP_OK(hipMemcpyAsync(dst, src, cnt * sizeof(double), hipMemcpyHostToDevice, s->stream)); };
    s->id_const_uploaded = false;  // H, g, u of a later mpc_qp_solve_id are uploaded again
    s->ikid_const.clear();         // ... and u, l_box, u_box of a later mpc_qp_solve_ikid
    up(s->dH, H, B * n * n); up(s->dg, g, B * n); up(s->dA, A, B * neq * n); up(s->db, b, B * neq);
    up(s->dC, C, B * nin * n); up(s->dl, l, B * nin); up(s->du, u, B * nin);
    if (d.box) { up(s->dlb, l_box, B * n); up(s->dub, u_box, B * n); }
    if (!S->warm_start) {
      HIP_OK(hipMemsetAsync(s->dx, 0, B * n * sizeof(double), s->stream));
      if (neq) HIP_OK(hipMemsetAsync(s->dy, 0, B * neq * sizeof(double), s->stream));
      if (m) HIP_OK(hipMemsetAsync(s->dz, 0, B * m * sizeof(double), s->stream));
    }
    qp_launch_and_fetch(s, S, x, y, z, z_box, info);
    return 0;
  } catch (const std::exception& e) {
    s->err = e.what();
    return -1;
  }
}

int mpc_qp_set_model(mpc_qp_solver* s, const int32_t* itab, int32_t n_i, const double* dtab, int32_t n_d) {
  if (!s) return -2;
  try {
    HIP_OK(hipSetDevice(s->d.device));
    if (!itab || !dtab || n_i < MPC_MODEL_HEADER_WORDS) throw std::runtime_error("qp_set_model: model table too short");
    const int nj = itab[0], nf = itab[3], ncn = itab[4];
    if (n_i < MPC_MODEL_HEADER_WORDS + MPC_MODEL_JOINT_WORDS * nj + nf + ncn ||
        n_d < MPC_MODEL_HEADER_DOUBLES + MPC_MODEL_JOINT_DOUBLES * nj + MPC_MODEL_FRAME_DOUBLES * nf + MPC_MODEL_CONTACT_DOUBLES * ncn)
      throw std::runtime_error("qp_set_model: model table size mismatch");
    for (int i = 0; i < nj; ++i) {
      const int32_t* ip = itab + MPC_MODEL_HEADER_WORDS + MPC_MODEL_JOINT_WORDS * i;
      if (ip[0] >= i) throw std::runtime_error("qp_set_model: joints must be topologically ordered");
      if ((ip[1] == MPC_JOINT_FREEFLYER) != (i == 0)) throw std::runtime_error("qp_set_model: free-flyer root followed by revolute joints expected");
    }
    s->d_mi = s->alloc<int32_t>(n_i); s->d_md = s->alloc<double>(n_d);
    HIP_OK(hipMemcpyAsync(s->d_mi, itab, n_i * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_md, dtab, n_d * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    s->m_nj = nj; s->m_nq = itab[1]; s->m_nv = itab[2]; s->m_nframes = nf;
    s->id_const_uploaded = false;
    return 0;
  } catch (const std::exception& e) { s->err = e.what(); return -1; }
}

}  // extern "C"

// buffers and constants of the inverse-dynamics QP (H = diag(w0 I_nv, w1 I_6nk, 0), g = 0, u = 1e5 as the reference, frames, cone rows: uploaded when they change)
void qp_id_prepare(mpc_qp_solver* s, int32_t nk, const int32_t* frames, const double* weights, const double* cone) {
  HIP_OK(hipSetDevice(s->d.device));
  if (!frames || !weights || !cone) throw std::runtime_error("qp_solve_id: null argument");
  if (!s->d_mi) throw std::runtime_error("qp_solve_id: mpc_qp_set_model first");
  const mpc_qp_dims& d = s->d;
  const int nv = s->m_nv, nq = s->m_nq;
  if (nk <= 0 || d.n != 2 * nv - 6 + 6 * nk || d.neq != nv + 6 * nk || d.nin != 9 * nk || d.box)
    throw std::runtime_error("qp_solve_id: the handle's dimensions are not those of the inverse-dynamics QP (n = 2 nv - 6 + 6 nk, neq = nv + 6 nk, nin = 9 nk, no box)");
  for (int c = 0; c < nk; ++c) if (frames[c] < 0 || frames[c] >= s->m_nframes) throw std::runtime_error("qp_solve_id: contact frame index out of range");
  const size_t B = d.batch, n = d.n, nin = d.nin;
  if (!s->d_xrob || s->id_nk != nk) {
    s->d_xrob = s->alloc<double>(B * (nq + nv)); s->d_acc = s->alloc<double>(B * nv); s->d_f = s->alloc<double>(B * 6 * nk);
    s->d_cs = s->alloc<int32_t>(B * nk); s->d_frames = s->alloc<int32_t>(nk); s->d_cone = s->alloc<double>(108);
    s->id_nk = nk; s->id_const_uploaded = false;
    qp_lds_attr_max((const void*)k_qp_assemble<false>, (int)qp_assemble_lds_bytes(s->m_nj, nv, nq, nk));
  }
  std::vector<double> key(110 + nk);
  key[0] = weights[0]; key[1] = weights[1];
  for (int i = 0; i < 108; ++i) key[2 + i] = cone[i];
  for (int c = 0; c < nk; ++c) key[110 + c] = frames[c];
  if (!s->id_const_uploaded || key != s->id_const) {
    std::vector<double> H(n * n, 0.0), g(n, 0.0), u(nin, 1e5);
    for (int i = 0; i < nv; ++i) H[(size_t)i * n + i] = weights[0];
    for (int i = 0; i < 6 * nk; ++i) H[(size_t)(nv + i) * n + nv + i] = weights[1];
    for (size_t bi = 0; bi < B; ++bi) {
      HIP_OK(hipMemcpyAsync(s->dH + bi * n * n, H.data(), n * n * sizeof(double), hipMemcpyHostToDevice, s->stream));
      HIP_OK(hipMemcpyAsync(s->dg + bi * n, g.data(), n * sizeof(double), hipMemcpyHostToDevice, s->stream));
      HIP_OK(hipMemcpyAsync(s->du + bi * nin, u.data(), nin * sizeof(double), hipMemcpyHostToDevice, s->stream));
    }
    HIP_OK(hipMemcpyAsync(s->d_frames, frames, nk * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_cone, cone, 108 * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));  // (host vectors go out of scope)
    s->id_const = key; s->id_const_uploaded = true;
  }
}

// assembly + solve of the inverse-dynamics QP from d_xrob / d_acc / d_f / d_cs, enqueued on the handle's stream
void qp_id_enqueue(mpc_qp_solver* s, const mpc_qp_settings* S, double kd) {
  const mpc_qp_dims& d = s->d;
  const int nv = s->m_nv, nq = s->m_nq, nk = s->id_nk;
  const size_t B = d.batch, n = d.n, neq = d.neq, nin = d.nin;
  QpAssembleArgs qa = {};
  qa.mi = s->d_mi; qa.md = s->d_md; qa.x = s->d_xrob; qa.acc = s->d_acc; qa.f = s->d_f; qa.cs = s->d_cs; qa.frames = s->d_frames; qa.cone = s->d_cone;
  qa.kd = kd; qa.nk = nk; qa.n = (int)n; qa.neq = (int)neq; qa.nin = (int)nin;
  qa.A = s->dA; qa.b = s->db; qa.C = s->dC; qa.l = s->dl;
  hipLaunchKernelGGL(k_qp_assemble<false>, dim3(d.batch), dim3(QPA_THREADS), qp_assemble_lds_bytes(s->m_nj, nv, nq, nk), s->stream, qa);
  HIP_OK(hipGetLastError());
  if (!S->warm_start) {
    HIP_OK(hipMemsetAsync(s->dx, 0, B * n * sizeof(double), s->stream));
    HIP_OK(hipMemsetAsync(s->dy, 0, B * neq * sizeof(double), s->stream));
    HIP_OK(hipMemsetAsync(s->dz, 0, B * nin * sizeof(double), s->stream));
  }
}

QpIdBuffers qp_id_buffers(mpc_qp_solver* s) {
  QpIdBuffers o;
  o.stream = s->stream; o.xrob = s->d_xrob; o.acc = s->d_acc; o.f = s->d_f; o.cs = s->d_cs; o.sol = s->dx; o.info = s->dinfo;
  o.B = s->d.batch; o.n = s->d.n; o.nq = s->m_nq; o.nv = s->m_nv; o.nk = s->id_nk; o.device = s->d.device;
  return o;
}
void qp_launch_solve(mpc_qp_solver* s, const mpc_qp_settings* S) { qp_launch(s, S); }
double* qp_scratch(mpc_qp_solver* s, size_t doubles) {
  if (s->glue_cap < doubles) { s->d_glue = s->alloc<double>(doubles); s->glue_cap = doubles; }
  return s->d_glue;
}
void qp_set_error(mpc_qp_solver* s, const char* what) { s->err = what; }

extern "C" {

int mpc_qp_solve_id(mpc_qp_solver* s, const mpc_qp_settings* S, int32_t nk, const int32_t* frames, const double* weights, const double* cone, double kd,
                    const double* xrob, const double* acc, const double* forces, const int32_t* contact_states,
                    double* x, double* y, double* z, mpc_qp_info* info, double* A_out, double* b_out, double* C_out, double* l_out) {
  if (!s) return -2;
  try {
    if (!S || !xrob || !acc || !forces || !contact_states || !x || !info) throw std::runtime_error("qp_solve_id: null argument");
    qp_id_prepare(s, nk, frames, weights, cone);
    const mpc_qp_dims& d = s->d;
    const int nv = s->m_nv, nq = s->m_nq;
    const size_t B = d.batch, n = d.n, neq = d.neq, nin = d.nin;
    HIP_OK(hipMemcpyAsync(s->d_xrob, xrob, B * (nq + nv) * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_acc, acc, B * nv * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_f, forces, B * 6 * nk * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_cs, contact_states, B * nk * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    qp_id_enqueue(s, S, kd);
    if (A_out) HIP_OK(hipMemcpyAsync(A_out, s->dA, B * neq * n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (b_out) HIP_OK(hipMemcpyAsync(b_out, s->db, B * neq * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (C_out) HIP_OK(hipMemcpyAsync(C_out, s->dC, B * nin * n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (l_out) HIP_OK(hipMemcpyAsync(l_out, s->dl, B * nin * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    qp_launch_and_fetch(s, S, x, y, z, nullptr, info);
    return 0;
  } catch (const std::exception& e) { s->err = e.what(); return -1; }
}

int mpc_qp_solve_ikid(mpc_qp_solver* s, const mpc_qp_settings* S, int32_t nk, const int32_t* frames, int32_t base_frame, int32_t torso_frame,
                      const double* weights, const double* gains, const double* cone, const double* l_box, const double* u_box,
                      const double* xrob, const double* ik, const double* forces, const int32_t* contact_states,
                      double* x, double* y, double* z, double* z_box, mpc_qp_info* info,
                      double* H_out, double* g_out, double* A_out, double* b_out, double* C_out, double* l_out) {
  if (!s) return -2;
  try {
    HIP_OK(hipSetDevice(s->d.device));
    if (!S || !frames || !weights || !gains || !cone || !l_box || !u_box || !xrob || !ik || !forces || !contact_states || !x || !info) throw std::runtime_error("qp_solve_ikid: null argument");
    if (!s->d_mi) throw std::runtime_error("qp_solve_ikid: mpc_qp_set_model first");
    const mpc_qp_dims& d = s->d;
    const int nv = s->m_nv, nq = s->m_nq;
    if (nk != 2 || d.n != 2 * nv - 6 + 6 * nk || d.neq != nv + 6 * nk || d.nin != 9 * nk || !d.box)
      throw std::runtime_error("qp_solve_ikid: two contacts and the handle's dimensions n = 2 nv - 6 + 6 nk, neq = nv + 6 nk, nin = 9 nk, box = 1 expected");
    for (int c = 0; c < nk + 2; ++c) {
      const int fi = c < nk ? frames[c] : (c == nk ? base_frame : torso_frame);
      if (fi < 0 || fi >= s->m_nframes) throw std::runtime_error("qp_solve_ikid: frame index out of range");
    }
    const size_t B = d.batch, n = d.n, neq = d.neq, nin = d.nin, m = nin + n, ngain = (size_t)2 * nv * nv + 90, nik = QPA_IK_DOUBLES(nv);
    if (!s->d_ik || s->ikid_nk != nk) {
      if (!s->d_xrob || s->id_nk != nk) {
        s->d_xrob = s->alloc<double>(B * (nq + nv)); s->d_acc = s->alloc<double>(B * nv); s->d_f = s->alloc<double>(B * 6 * nk);
        s->d_cs = s->alloc<int32_t>(B * nk); s->d_frames = s->alloc<int32_t>(nk); s->d_cone = s->alloc<double>(108);
        s->id_nk = nk;
      }
      s->d_ik = s->alloc<double>(B * nik); s->d_gains = s->alloc<double>(ngain); s->d_w = s->alloc<double>(8);
      s->ikid_nk = nk; s->ikid_const.clear();
      qp_lds_attr_max((const void*)k_qp_assemble<true>, (int)qp_assemble_lds_bytes(s->m_nj, nv, nq, nk, true));
    }
    s->id_const_uploaded = false;  // (H, g, u of a later mpc_qp_solve_id are uploaded again)
    // constants: weights, gains, cone rows, frames, the torque box and u = 1e5 — uploaded when they change
    std::vector<double> key;
    key.reserve(5 + ngain + 54 + 2 * n + nk + 2);
    key.insert(key.end(), weights, weights + 5); key.insert(key.end(), gains, gains + ngain); key.insert(key.end(), cone, cone + 108);
    key.insert(key.end(), l_box, l_box + n); key.insert(key.end(), u_box, u_box + n);
    for (int c = 0; c < nk; ++c) key.push_back(frames[c]);
    key.push_back(base_frame); key.push_back(torso_frame);
    if (key != s->ikid_const) {
      std::vector<double> u(nin, 1e5);
      HIP_OK(hipMemcpyAsync(s->d_w, weights, 5 * sizeof(double), hipMemcpyHostToDevice, s->stream));
      HIP_OK(hipMemcpyAsync(s->d_gains, gains, ngain * sizeof(double), hipMemcpyHostToDevice, s->stream));
      HIP_OK(hipMemcpyAsync(s->d_cone, cone, 108 * sizeof(double), hipMemcpyHostToDevice, s->stream));
      HIP_OK(hipMemcpyAsync(s->d_frames, frames, nk * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
      for (size_t bi = 0; bi < B; ++bi) {
        HIP_OK(hipMemcpyAsync(s->du + bi * nin, u.data(), nin * sizeof(double), hipMemcpyHostToDevice, s->stream));
        HIP_OK(hipMemcpyAsync(s->dlb + bi * n, l_box, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
        HIP_OK(hipMemcpyAsync(s->dub + bi * n, u_box, n * sizeof(double), hipMemcpyHostToDevice, s->stream));
      }
      HIP_OK(hipStreamSynchronize(s->stream));
      s->ikid_const = key;
    }
    HIP_OK(hipMemcpyAsync(s->d_xrob, xrob, B * (nq + nv) * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_ik, ik, B * nik * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_f, forces, B * 6 * nk * sizeof(double), hipMemcpyHostToDevice, s->stream));
    HIP_OK(hipMemcpyAsync(s->d_cs, contact_states, B * nk * sizeof(int32_t), hipMemcpyHostToDevice, s->stream));
    QpAssembleArgs qa = {};
    qa.mi = s->d_mi; qa.md = s->d_md; qa.x = s->d_xrob; qa.f = s->d_f; qa.cs = s->d_cs; qa.frames = s->d_frames; qa.cone = s->d_cone;
    qa.nk = nk; qa.n = (int)n; qa.neq = (int)neq; qa.nin = (int)nin;
    qa.A = s->dA; qa.b = s->db; qa.C = s->dC; qa.l = s->dl; qa.H = s->dH; qa.g = s->dg;
    qa.base_frame = base_frame; qa.torso_frame = torso_frame; qa.w = s->d_w; qa.gains = s->d_gains; qa.ik = s->d_ik;
    hipLaunchKernelGGL(k_qp_assemble<true>, dim3(d.batch), dim3(QPA_THREADS), qp_assemble_lds_bytes(s->m_nj, nv, nq, nk, true), s->stream, qa);
    HIP_OK(hipGetLastError());
    if (!S->warm_start) {
      HIP_OK(hipMemsetAsync(s->dx, 0, B * n * sizeof(double), s->stream));
      HIP_OK(hipMemsetAsync(s->dy, 0, B * neq * sizeof(double), s->stream));
      HIP_OK(hipMemsetAsync(s->dz, 0, B * m * sizeof(double), s->stream));
    }
    if (H_out) HIP_OK(hipMemcpyAsync(H_out, s->dH, B * n * n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (g_out) HIP_OK(hipMemcpyAsync(g_out, s->dg, B * n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (A_out) HIP_OK(hipMemcpyAsync(A_out, s->dA, B * neq * n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (b_out) HIP_OK(hipMemcpyAsync(b_out, s->db, B * neq * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (C_out) HIP_OK(hipMemcpyAsync(C_out, s->dC, B * nin * n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (l_out) HIP_OK(hipMemcpyAsync(l_out, s->dl, B * nin * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    qp_launch_and_fetch(s, S, x, y, z, z_box, info);
    return 0;
  } catch (const std::exception& e) { s->err = e.what(); return -1; }
}

}  // extern "C"
