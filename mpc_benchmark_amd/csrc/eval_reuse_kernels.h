// eval_reuse_kernels.h — the two small kernels of tick reuse that go with the stage kernels (mpc_set_tick_reuse, include/mpc_abi.h).
#pragma once
#include "eval_common.h"

// Tick reuse, appended knot: its record is the speculative one of the previous tick — from the spare slot to its place in the ring
// (the slot knot 0 has just left).  grid (32, B), block 256: 32 workgroups share the 220 KB of a record
__global__ void __launch_bounds__(256) k_copy_spec(SolverArgs a) {
  const Layout& L = a.L;
  const int b = blockIdx.y;
  if (a.inst[b].done || !knot_reused(a, b, L.N - 1)) return;
  const double* sp = a.spec_knot + (size_t)b * L.knot_stride;
  double* kn = knot_ptr(a, b, L.N - 1);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < L.knot_stride; i += 32 * 256) kn[i] = sp[i];
}

// Tick reuse: the record of a reused knot holds the evaluation of the current point already (written by the full-step
// candidate of the previous tick); only what depends on the multipliers — projections, active flags, penalty,
// infeasibility — is refreshed here, exactly as the tail of the stage kernel does.  grid (N+1, B), block 256
__global__ void __launch_bounds__(256) k_reproject(SolverArgs a) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  const InstState& st = a.inst[b];
  if (st.done || !knot_reused(a, b, k)) return;
  __shared__ double red[2 * 256 + 8];
  double* kn = knot_ptr(a, b, k);
  const int c = (int)kn[L.oMISC + MISC_NC], N = L.N, n = L.n;
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const size_t vo = ((size_t)b * (N + 1) + k) * L.c, lo = ((size_t)b * (N + 1) + k + 1) * n;
  double pen = 0, prim = 0;
  knot_merit(L, kn, c, (k < N) ? kn + L.oF : nullptr, a.vs + vo, nullptr, a.vs_e + vo, a.lams + lo, nullptr, a.lams_e + lo, 0.0, mu, mud, true, red,
             pen, prim, tid, nthr);
  if (tid == 0) { kn[L.oMISC + MISC_PEN] = pen; kn[L.oMISC + MISC_PRIM] = prim; }
}

