// eval_multibody.hip — translation unit of the whole-body stage kernel (K1-K5, K7 of SURVEY.md §8a-2) and its launcher.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdlib>
#include <stdexcept>
#include <string>
#define CHOL16_RIGHT_LOOKING  // (mfma_blocks.h: the 16 x 16 inverse with independent updates — pays in this kernel's one-wavefront chain only)
#include "eval_multibody.h"

void launch_eval_multibody(hipStream_t stream, const SolverArgs& a, const Layout& LT, double* records, double* scratch, size_t scratch_stride,
                           bool trial, int cand0, int ncand, int sim_substeps, double sim_dt, bool with_derivs, const double* f_ext, bool contact_dyn,
                           const double* sim_u, double* sim_wrench) {
  const Layout& L = a.L;
  MbArgs mb;
  mb.lds = make_mb_lds(L.nj, L.n / 2, L.nx - L.n / 2, L.m, L.nz, contact_dyn);
  mb.scratch = scratch;
  mb.scratch_stride = scratch_stride;
  mb.sim_substeps = sim_substeps; mb.sim_dt = sim_dt; mb.f_ext = f_ext; mb.sim_u = sim_u; mb.sim_wrench = sim_wrench;
  mb.ncand_loop = (trial && !with_derivs && ncand > 1) ? ncand : 0;
  if (mb.lds.total_bytes > 160 * 1024) throw std::runtime_error("multibody model too large for the LDS budget of the stage kernel");
  // hipFuncSetAttribute is a per-device setting: remember what was requested on every device (a process may hold handles on several
  // devices, driven from different threads)
  static std::atomic<int> attr_bytes_dev[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::atomic<int>& attr_bytes = attr_bytes_dev[dev & 63];
  if (attr_bytes.load() < mb.lds.total_bytes + 1) {  // (the largest request so far stays: the two carve-outs of a process differ)
    // the kernel also owns a few bytes of static LDS, so request exactly what the carve-out needs
    hipError_t e1 = hipFuncSetAttribute((const void*)k_eval_multibody<0>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    hipError_t e2 = hipFuncSetAttribute((const void*)k_eval_multibody<1>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)k_eval_multibody<2>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    if (e2 == hipSuccess) e2 = hipFuncSetAttribute((const void*)k_eval_multibody<3>, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
    if (e1 != hipSuccess || e2 != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(LDS) failed: ") + hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    attr_bytes.store(mb.lds.total_bytes + 1);  // + 1: the zero-initialised slots mean "not set"
  }
  // fixed-dimension instantiations (eval_multibody.h):  X(id, bodies, velocity dofs, controls, contact-constrained dynamics)
  //   1, 2: the complete Talos (33 bodies, 38 dofs), full-dynamics OCP (32 joint torques) and kinodynamic OCP (12 wrench components + 32 joint
  //   accelerations, no contact-dynamics blocks) ; 3, 4: the Talos with the upper body locked as the scripts lock it (23 bodies, 28 dofs)
#define MB_FIXED_MODELS(X) X(1, 33, 38, 32, true) X(2, 33, 38, 44, false) X(3, 23, 28, 22, true) X(4, 23, 28, 34, false)
  int fixed = 0;
  if (!getenv("MPC_HIP_GENERIC_DIMS") && sim_substeps <= 0) {
#define X(ID, FJ, FV, FU, FCD) if (L.nj == FJ && L.n == 2 * FV && L.nx == 2 * FV + 1 && L.m == FU && L.nz == 2 * FV + FU && contact_dyn == FCD) fixed = ID;
    MB_FIXED_MODELS(X)
#undef X
  }
  if (fixed) {
    static std::atomic<int> attr_fixed_dev[4][64];
    std::atomic<int>& done = attr_fixed_dev[fixed - 1][dev & 63];
    const void *fn0 = nullptr, *fn1 = nullptr, *fn3 = nullptr;
#define X(ID, FJ, FV, FU, FCD) if (fixed == ID) { fn0 = (const void*)k_eval_multibody<0, FJ, FV, FU, FCD>; fn1 = (const void*)k_eval_multibody<1, FJ, FV, FU, FCD>; fn3 = (const void*)k_eval_multibody<3, FJ, FV, FU, FCD>; }
    MB_FIXED_MODELS(X)
#undef X
    if (done.load() < mb.lds.total_bytes + 1) {
      hipError_t e1 = hipFuncSetAttribute(fn0, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
      if (e1 == hipSuccess) e1 = hipFuncSetAttribute(fn1, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
      if (e1 == hipSuccess) e1 = hipFuncSetAttribute(fn3, hipFuncAttributeMaxDynamicSharedMemorySize, mb.lds.total_bytes);
      if (e1 != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(LDS) failed: ") + hipGetErrorString(e1));
      done.store(mb.lds.total_bytes + 1);
    }
#define X3(ID, FJ, FV, FU, FCD) else if (fixed == ID) hipLaunchKernelGGL((k_eval_multibody<3, FJ, FV, FU, FCD>), dim3(L.N + 1 + (a.spec_knot ? 1 : 0), L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, L, records, mb, 0);
#define X0(ID, FJ, FV, FU, FCD) else if (fixed == ID) hipLaunchKernelGGL((k_eval_multibody<0, FJ, FV, FU, FCD>), dim3(a.only_knot >= 0 ? 1 : L.N + 1, L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, L, records, mb, 0);
#define X1(ID, FJ, FV, FU, FCD) else if (fixed == ID) hipLaunchKernelGGL((k_eval_multibody<1, FJ, FV, FU, FCD>), dim3(L.N + 1, L.B, mb.ncand_loop > 0 ? 1 : ncand), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, LT, records, mb, cand0);
    if (trial && with_derivs) { if (false) {} MB_FIXED_MODELS(X3) }
    else if (!trial) { if (false) {} MB_FIXED_MODELS(X0) }
    else { if (false) {} MB_FIXED_MODELS(X1) }
#undef X3
#undef X0
#undef X1
    return;
  }
  if (sim_substeps > 0) hipLaunchKernelGGL(k_eval_multibody<2>, dim3(1, L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, LT, records, mb, 0);
  else if (trial && with_derivs) hipLaunchKernelGGL(k_eval_multibody<3>, dim3(L.N + 1 + (a.spec_knot ? 1 : 0), L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, L, records, mb, 0);  // records = the knot records (+ the speculative knot)
  else if (!trial) hipLaunchKernelGGL(k_eval_multibody<0>, dim3(a.only_knot >= 0 ? 1 : L.N + 1, L.B, 1), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, L, records, mb, 0);
  else hipLaunchKernelGGL(k_eval_multibody<1>, dim3(L.N + 1, L.B, mb.ncand_loop > 0 ? 1 : ncand), dim3(EVAL_THREADS), mb.lds.total_bytes, stream, a, LT, records, mb, cand0);
}

const void* eval_multibody_kernel(int trial) {
  switch (trial) {
    case 0: return (const void*)k_eval_multibody<0>;
    case 1: return (const void*)k_eval_multibody<1>;
    case 2: return (const void*)k_eval_multibody<2>;
    default: return (const void*)k_eval_multibody<3>;
  }
}
