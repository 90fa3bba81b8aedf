// solver_kernels.h — stage-kind independent kernels of the on-device ProxDDP iteration:
//   k_lagrangian      (P4)  gradient of the Lagrangian per knot -> dual infeasibility / inner criterion
//   k_decide          (P4)  per-instance reductions + BCL outer-loop bookkeeping (no host round trip)
//   k_riccati_backward(P6)  proximal Riccati sweep, one workgroup per MPC instance, sequential over knots
//   k_forward         (P7)  state/control steps along the horizon
//   k_duals           (P7)  multiplier steps + merit directional derivative, parallel over knots
//   k_linesearch      (P8)  deterministic reductions of the candidate merits, Armijo choice
//   k_accept          (P9)  apply the accepted step
// Phase names follow SURVEY.md §3.3; the algebra follows SURVEY.md App. B.3/B.4 as restated in DESIGN.md.
#pragma once
#include "solver_args.h"


// ------------------------------------------------------------------------------------------------
// P4: L_z = grad + CD^T v + AB^T lam' + [E_{k-1}^T lam_k; 0]   (current multipliers), inner criterion
// grid (N+1, B), block 64
// ------------------------------------------------------------------------------------------------
__global__ void k_lagrangian(SolverArgs a) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  const InstState& st = a.inst[b];
  if (st.done) return;
  double* kn = knot_ptr(a, b, k);
  const int n = L.n, nz = L.nz, N = L.N;
  const int c = (int)kn[L.oMISC + MISC_NC], m = (int)kn[L.oMISC + MISC_M], nzk = n + m;
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const double* v = a.vs + ((size_t)b * (N + 1) + k) * L.c;
  const double* lamn = a.lams + ((size_t)b * (N + 1) + k + 1) * n;
  const double* lamk = a.lams + ((size_t)b * (N + 1) + k) * n;
  __shared__ double red[64];
  double dual = 0.0, crit = 0.0;
  for (int z = tid; z < nzk; z += nthr) {
    double s = kn[L.oG + z];
    for (int i = 0; i < c; ++i) if (v[i] != 0.0) s += kn[L.oCD + i * nz + z] * v[i];  // rows with a zero multiplier are not read (all of them right after setup)
    if (k < N) for (int i = 0; i < n; ++i) if (lamn[i] != 0.0) s += kn[L.oAB + i * nz + z] * lamn[i];
    if (k > 0 && z < n) {
      if (L.space == MPC_SPACE_MULTIBODY && z < 6) {
        const double* E6 = knot_ptr(a, b, k - 1) + L.oE6;
        for (int i = 0; i < 6; ++i) s += E6[i * 6 + z] * lamk[i];
      } else {
        s -= lamk[z];
      }
    }
    const bool fixed = (k == 0 && a.opt.force_initial_condition && z < n);
    if (!fixed) dual = fmax(dual, fabs(s));
  }
  for (int i = tid; i < c; i += nthr) crit = fmax(crit, fabs(kn[L.oDT + i] - mu * v[i]));
  if (k < N) {
    const double* le = a.lams_e + ((size_t)b * (N + 1) + k + 1) * n;
    for (int i = tid; i < n; i += nthr) crit = fmax(crit, fabs(kn[L.oF + i] + mud * (le[i] - lamn[i])));
  }
  red[tid] = dual;
  __syncthreads();
  if (tid == 0) { double r = 0; for (int i = 0; i < nthr; ++i) r = fmax(r, red[i]); kn[L.oMISC + MISC_DUAL] = r; }
  __syncthreads();
  red[tid] = crit;
  __syncthreads();
  if (tid == 0) { double r = 0; for (int i = 0; i < nthr; ++i) r = fmax(r, red[i]); kn[L.oMISC + MISC_CRIT] = r; }
}

// ------------------------------------------------------------------------------------------------
// P4 (cont.): per-instance reductions in fixed order + the BCL bookkeeping of SolverProxDDP::run.
// grid B, block 128: the knot partials are gathered in parallel, summed in knot order by one thread
// (deterministic), and the multiplier-estimate refresh of an accepted BCL step is a block-wide copy.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(128) k_decide(SolverArgs a) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  InstState& st = a.inst[b];
  if (st.done) return;
  const mpc_options& o = a.opt;
  __shared__ double part[5][128];
  __shared__ int refresh;
  double cost = 0, pen = 0, prim = 0, dual = 0, crit = 0;
  for (int k0 = 0; k0 <= L.N; k0 += nthr) {  // chunks of nthr knots, each summed in knot order
    const int k = k0 + tid;
    if (k <= L.N) {
      const double* ms = knot_ptr(a, b, k) + L.oMISC;
      part[0][tid] = ms[MISC_COST]; part[1][tid] = ms[MISC_PEN]; part[2][tid] = ms[MISC_PRIM];
      part[3][tid] = ms[MISC_DUAL]; part[4][tid] = ms[MISC_CRIT];
    }
    __syncthreads();
    if (tid == 0) {
      const int cnt = min(nthr, L.N + 1 - k0);
      for (int i = 0; i < cnt; ++i) {
        cost += part[0][i]; pen += part[1][i];
        prim = fmax(prim, part[2][i]); dual = fmax(dual, part[3][i]); crit = fmax(crit, part[4][i]);
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    refresh = 0;
    crit = fmax(crit, dual);
    st.cost = cost; st.phi0 = cost + pen; st.prim = prim; st.dual = dual; st.crit = crit;
    st.skip_step = 0;
    const bool via_stall = (st.stalled & 1) != 0;  // bit 0: the previous pass found no descent left; bits 8..: consecutive stalls
    if (crit <= st.inner_tol || via_stall) {
      // inner problem solved: outer (BCL) update, no step this pass
      st.stalled &= ~1;
      st.skip_step = 1;
      if (prim <= st.prim_tol) {
        st.prim_tol *= pow(st.mu, o.bcl_prim_beta);
        st.inner_tol *= pow(st.mu, o.bcl_dual_beta);
        refresh = 1;
        if (fmax(prim, dual) <= o.tol) { st.converged = 1; st.done = 1; }
      } else {
        st.mu = fmax(st.mu * o.bcl_mu_update_factor, o.bcl_mu_lower_bound);
        st.prim_tol = o.prim_tol0 * pow(st.mu, o.bcl_prim_alpha);
        st.inner_tol = o.inner_tol0 * pow(st.mu, o.bcl_dual_alpha);
      }
      st.inner_tol = fmax(st.inner_tol, o.tol);
      st.prim_tol = fmax(st.prim_tol, o.tol);
      st.al_iters += 1;
      if (st.al_iters >= o.max_al_iters) st.done = 1;
      if (via_stall && (st.stalled >> 8) >= 4) st.done = 1;  // four stalls with no step in between (two full BCL cycles): nothing left to gain
      // the run of this instance ends here (k_after_step returns early for it): its records stay the evaluation of its iterate, but
      // whatever speculative record of an appended knot exists belongs to an earlier tick
      if (st.done && a.spec) a.spec[b] = a.spec_on ? 2 : 0;
    }
  }
  __syncthreads();
  if (refresh) {
    const size_t nv = (size_t)(L.N + 1) * L.c, nl = (size_t)(L.N + 1) * L.n;
    for (size_t i = tid; i < nv; i += nthr) a.vs_e[b * nv + i] = a.vs[b * nv + i];
    for (size_t i = tid; i < nl; i += nthr) a.lams_e[b * nl + i] = a.lams[b * nl + i];
  }
}

// ------------------------------------------------------------------------------------------------
// P6: proximal Riccati backward sweep.  grid B, block 256, dynamic LDS: Lp[n*n] Lr[m*m] Sc[c*c] + small.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_riccati_backward(SolverArgs a) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, nz = L.nz, N = L.N, nr = n + 1;
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const bool ff = L.space == MPC_SPACE_MULTIBODY;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Lp = lds;                 // n*n
  double* Lr = Lp + n * n;          // m*m
  double* ScL = Lr + L.m * L.m;     // sc_cap^2: Schur complement of the ACTIVE constraint rows when it fits
  int* act_idx = (int*)(ScL + L.sc_cap * L.sc_cap);  // c
  int* iflag = act_idx + L.c;       // 2
  double* wk = a.work + (size_t)b * L.work_stride;
  double *Ph = wk + L.wPh, *Pt = wk + L.wPt, *G = wk + L.wG, *Hh = wk + L.wHh, *gh = wk + L.wgh, *Ct = wk + L.wCt;
  double *W = wk + L.wW, *Y = wk + L.wY, *V = wk + L.wV, *Acl = wk + L.wAcl, *vec = wk + L.wvec;
  double *ph = vec, *ft = vec + n, *w = vec + 2 * n;

  // ---- terminal node ----
  {
    const double* kn = knot_ptr(a, b, N);
    double* g = gain_ptr(a, b, N);
    const int c = (int)kn[L.oMISC + MISC_NC];
    for (int i = tid; i < c; i += nthr) g[L.oknu + i] = kn[L.oDT + i] / mu;
    for (int idx = tid; idx < c * n; idx += nthr) {
      const int i = idx / n, z = idx % n;
      g[L.oKnu + idx] = (kn[L.oACT + i] != 0.0) ? kn[L.oCD + i * nz + z] / mu : 0.0;
    }
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int r = idx / n, s = idx % n;
      double t = kn[L.oH + r * nz + s];
      for (int i = 0; i < c; ++i) if (kn[L.oACT + i] != 0.0) t += kn[L.oCD + i * nz + r] * g[L.oKnu + i * n + s];
      g[L.oP + idx] = t;
    }
    for (int r = tid; r < n; r += nthr) {
      double t = kn[L.oG + r];
      for (int i = 0; i < c; ++i) t += kn[L.oCD + i * nz + r] * g[L.oknu + i];
      g[L.op + r] = t;
    }
    __syncthreads();
  }

  for (int k = N - 1; k >= 0; --k) {
    const double* kn = knot_ptr(a, b, k);
    const double* gn = gain_ptr(a, b, k + 1);
    double* g = gain_ptr(a, b, k);
    const int m = (int)kn[L.oMISC + MISC_M], c = (int)kn[L.oMISC + MISC_NC];
    const double* AB = kn + L.oAB;
    const double* le = a.lams_e + ((size_t)b * (N + 1) + k + 1) * n;
    // 1. y = Ebar x' : T6 = (-E6)^-1 ; Ph = T^T P' T ; ph = T^T p'
    if (tid == 0) {
      if (ff) {
        double Eb[36];
        for (int i = 0; i < 36; ++i) Eb[i] = -kn[L.oE6 + i];
        inv6_serial(Eb, g + L.oT6);
      } else {
        for (int i = 0; i < 36; ++i) g[L.oT6 + i] = (i % 7 == 0) ? 1.0 : 0.0;
      }
      int ca = 0;
      for (int i = 0; i < c; ++i) if (kn[L.oACT + i] != 0.0) act_idx[ca++] = i;
      iflag[1] = ca;
    }
    __syncthreads();
    const int ca = iflag[1];
    double* Sc = (ca <= L.sc_cap) ? ScL : (wk + L.wSc);
    const double* T6 = g + L.oT6;
    if (ff) {
      // Pt used as temporary: tmp = P' T
      for (int idx = tid; idx < n * n; idx += nthr) {
        const int i = idx / n, j = idx % n;
        double s;
        if (j < 6) { s = 0; for (int l = 0; l < 6; ++l) s += gn[L.oP + i * n + l] * T6[l * 6 + j]; }
        else s = gn[L.oP + idx];
        Pt[idx] = s;
      }
      __syncthreads();
      for (int idx = tid; idx < n * n; idx += nthr) {
        const int i = idx / n, j = idx % n;
        double s;
        if (i < 6) { s = 0; for (int l = 0; l < 6; ++l) s += T6[l * 6 + i] * Pt[l * n + j]; }
        else s = Pt[idx];
        Ph[idx] = s;
      }
      for (int i = tid; i < n; i += nthr) {
        double s;
        if (i < 6) { s = 0; for (int l = 0; l < 6; ++l) s += T6[l * 6 + i] * gn[L.op + l]; }
        else s = gn[L.op + i];
        ph[i] = s;
      }
    } else {
      for (int idx = tid; idx < n * n; idx += nthr) Ph[idx] = gn[L.oP + idx];
      for (int i = tid; i < n; i += nthr) ph[i] = gn[L.op + i];
    }
    for (int i = tid; i < n; i += nthr) ft[i] = kn[L.oF + i] + mud * le[i];
    __syncthreads();
    // 2. Lp = chol(I + mud sym(Ph)) ; Pt = Lam Ph ; w = Lam (Ph ft + ph)
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int i = idx / n, j = idx % n;
      Lp[idx] = mud * 0.5 * (Ph[idx] + Ph[j * n + i]) + (i == j ? 1.0 : 0.0);
      Pt[idx] = Ph[idx];
    }
    for (int i = tid; i < n; i += nthr) { double s = ph[i]; for (int j = 0; j < n; ++j) s += Ph[i * n + j] * ft[j]; w[i] = s; }
    __syncthreads();
    if (!chol_block(Lp, n, n, tid, nthr, iflag)) { if (tid == 0) a.inst[b].done = 2; return; }
    potrs_block(Lp, n, n, Pt, n, n, tid, nthr);
    potrs_block(Lp, n, n, w, 1, 1, tid, nthr);
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int i = idx / n, j = idx % n;
      if (j > i) { const double s = 0.5 * (Pt[idx] + Pt[j * n + i]); Pt[idx] = s; Pt[j * n + i] = s; }
    }
    __syncthreads();
    // 3. G = Pt AB ; Hh = H + AB^T G ; gh = grad + AB^T w
    const int nzk = n + m;
    for (int idx = tid; idx < n * nzk; idx += nthr) {
      const int i = idx / nzk, z = idx % nzk;
      double s = 0;
      for (int l = 0; l < n; ++l) s += Pt[i * n + l] * AB[l * nz + z];
      G[i * nz + z] = s;
    }
    __syncthreads();
    for (int idx = tid; idx < nzk * nzk; idx += nthr) {
      const int r = idx / nzk, z = idx % nzk;
      double s = kn[L.oH + r * nz + z];
      for (int i = 0; i < n; ++i) s += AB[i * nz + r] * G[i * nz + z];
      Hh[r * nz + z] = s;
    }
    for (int z = tid; z < nzk; z += nthr) {
      double s = kn[L.oG + z];
      for (int i = 0; i < n; ++i) s += AB[i * nz + z] * w[i];
      gh[z] = s;
    }
    __syncthreads();
    // 4. stage KKT, controls first then the ACTIVE constraint rows (compacted)
    for (int idx = tid; idx < m * m; idx += nthr) {
      const int i = idx / m, j = idx % m;
      Lr[idx] = 0.5 * (Hh[(n + i) * nz + n + j] + Hh[(n + j) * nz + n + i]);
    }
    for (int idx = tid; idx < ca * nzk; idx += nthr) {
      const int i = idx / nzk, z = idx % nzk;
      Ct[i * nz + z] = kn[L.oCD + act_idx[i] * nz + z];
    }
    for (int idx = tid; idx < m * nr; idx += nthr) {
      const int i = idx / nr, z = idx % nr;
      W[idx] = (z < n) ? -Hh[(n + i) * nz + z] : -gh[n + i];
    }
    for (int idx = tid; idx < m * ca; idx += nthr) {
      const int i = idx / ca, j = idx % ca;
      Y[i * ca + j] = kn[L.oCD + act_idx[j] * nz + n + i];
    }
    __syncthreads();
    if (!chol_block(Lr, m, m, tid, nthr, iflag)) { if (tid == 0) a.inst[b].done = 3; return; }
    trsm_lower_block(Lr, m, m, W, nr, nr, tid, nthr);
    if (ca > 0) {
      trsm_lower_block(Lr, m, m, Y, ca, ca, tid, nthr);
      for (int idx = tid; idx < ca * ca; idx += nthr) {
        const int i = idx / ca, j = idx % ca;
        double s = (i == j) ? mu : 0.0;
        for (int l = 0; l < m; ++l) s += Y[l * ca + i] * Y[l * ca + j];
        Sc[idx] = s;
      }
      for (int idx = tid; idx < ca * nr; idx += nthr) {
        const int i = idx / nr, z = idx % nr;
        double s = (z < n) ? Ct[i * nz + z] : kn[L.oDT + act_idx[i]];
        for (int l = 0; l < m; ++l) s += Y[l * ca + i] * W[l * nr + z];
        V[idx] = s;
      }
      __syncthreads();
      if (!chol_block(Sc, ca, ca, tid, nthr, iflag)) { if (tid == 0) a.inst[b].done = 4; return; }
      potrs_block(Sc, ca, ca, V, nr, nr, tid, nthr);
      for (int idx = tid; idx < m * nr; idx += nthr) {
        const int l = idx / nr, z = idx % nr;
        double s = 0;
        for (int i = 0; i < ca; ++i) s += Y[l * ca + i] * V[i * nr + z];
        W[idx] -= s;
      }
      __syncthreads();
    }
    trsm_lower_t_block(Lr, m, m, W, nr, nr, tid, nthr);
    for (int idx = tid; idx < m * n; idx += nthr) g[L.oK + idx] = W[(idx / n) * nr + idx % n];
    for (int i = tid; i < m; i += nthr) g[L.ok + i] = W[i * nr + n];
    for (int idx = tid; idx < c * n; idx += nthr) g[L.oKnu + idx] = 0.0;
    for (int i = tid; i < c; i += nthr) g[L.oknu + i] = 0.0;
    __syncthreads();
    for (int idx = tid; idx < ca * n; idx += nthr) g[L.oKnu + act_idx[idx / n] * n + idx % n] = V[(idx / n) * nr + idx % n];
    for (int i = tid; i < ca; i += nthr) g[L.oknu + act_idx[i]] = V[i * nr + n];
    __syncthreads();
    // 5. value function
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int r = idx / n, s = idx % n;
      double t = Hh[r * nz + s];
      for (int i = 0; i < m; ++i) t += Hh[r * nz + n + i] * W[i * nr + s];
      for (int i = 0; i < ca; ++i) t += Ct[i * nz + r] * V[i * nr + s];
      g[L.oP + idx] = t;
    }
    for (int r = tid; r < n; r += nthr) {
      double t = gh[r];
      for (int i = 0; i < m; ++i) t += Hh[r * nz + n + i] * W[i * nr + n];
      for (int i = 0; i < ca; ++i) t += Ct[i * nz + r] * V[i * nr + n];
      g[L.op + r] = t;
    }
    // 6. what the forward sweep needs: x' = T Lam (A x + B u + ft - mud ph) with Lam = I - mud Pt
    for (int idx = tid; idx < n * n; idx += nthr) g[L.oMx + idx] = Pt[idx];
    for (int i = tid; i < n; i += nthr) g[L.omx + i] = ft[i] - mud * ph[i];
    __syncthreads();
    for (int idx = tid; idx < n * n; idx += nthr) {
      const int r = idx / n, s = idx % n;
      if (s > r) { const double t = 0.5 * (g[L.oP + idx] + g[L.oP + s * n + r]); g[L.oP + idx] = t; g[L.oP + s * n + r] = t; }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// P7: du_k = K dx_k + k ; y = A dx_k + B du_k + yv ; dx_{k+1} = T (y - mud Pt y).  grid B, block 256.
// Every product is a wave-per-row dot product with coalesced row reads and a shuffle reduction.
// ------------------------------------------------------------------------------------------------
// 16 wavefronts per instance: the sweep is a chain of latency-bound mat-vecs, more waves = more HBM loads in flight
__global__ void __launch_bounds__(1024) k_forward(SolverArgs a) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, N = L.N, m = L.m, nz = L.nz;
  const double mud = st.mu * a.opt.dyn_al_scale;
  const bool ff = L.space == MPC_SPACE_MULTIBODY;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* dz = lds;           // [dx; du]  (nz)
  double* y = lds + nz;       // n
  double* z = y + n;          // n
  for (int i = tid; i < n; i += blockDim.x) { dz[i] = 0.0; a.dxs[(size_t)b * (N + 1) * n + i] = 0.0; }
  __syncthreads();
  for (int k = 0; k < N; ++k) {
    const double* g = gain_ptr(a, b, k);
    const double* kn = knot_ptr(a, b, k);
    for (int r = wv; r < m; r += nw) {
      double s = 0;
      for (int c0 = lane; c0 < n; c0 += 64) s += g[L.oK + r * n + c0] * dz[c0];
      s = wave_sum(s);
      if (lane == 0) { s += g[L.ok + r]; dz[n + r] = s; a.dus[((size_t)b * N + k) * m + r] = s; }
    }
    __syncthreads();
    for (int r = wv; r < n; r += nw) {
      double s = 0;
      for (int c0 = lane; c0 < n + m; c0 += 64) s += kn[L.oAB + r * nz + c0] * dz[c0];
      s = wave_sum(s);
      if (lane == 0) y[r] = s + g[L.omx + r];
    }
    __syncthreads();
    for (int r = wv; r < n; r += nw) {
      double s = 0;
      for (int c0 = lane; c0 < n; c0 += 64) s += g[L.oMx + r * n + c0] * y[c0];
      s = wave_sum(s);
      if (lane == 0) z[r] = y[r] - mud * s;
    }
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x) {
      double s = z[i];
      if (ff && i < 6) { s = 0; for (int l = 0; l < 6; ++l) s += g[L.oT6 + i * 6 + l] * z[l]; }
      dz[i] = s;
      a.dxs[((size_t)b * (N + 1) + k + 1) * n + i] = s;
    }
    __syncthreads();
  }
}

// Same sweep with the operands of knot k+1 (rows of K, [A B], Pt and the small vectors) requested from HBM while
// knot k computes: the matrices do not depend on the recursion, only the vectors do, so every wavefront keeps "its"
// rows of the next knot in registers (<= 3 rows of K, 6 of [A B] and Pt, two columns per lane) and the chain of
// mat-vecs never waits on memory.  The barrier is a bare s_barrier behind the LDS counter, so that the loads in
// flight are not drained.  FW_KR / FW_NR = rows of K / of [A B] and Pt per wavefront (8 wavefronts, up to 256 registers each): m <= 8 FW_KR,
// n <= 8 FW_NR, n + m <= 128.
#define FW_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int FW_KR, int FW_NR>
__global__ void __launch_bounds__(512) k_forward_prefetch(SolverArgs a) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: row addresses are SGPR base + lane offset
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, N = L.N, m = L.m, nz = L.nz;
  const double mud = st.mu * a.opt.dyn_al_scale;
  const bool ff = L.space == MPC_SPACE_MULTIBODY;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* dz = lds;           // [dx; du]  (nz)
  double* y = lds + nz;       // n
  double* z = y + n;          // n
  double Kr[FW_KR][2], kf[FW_KR], ABr[FW_NR][2], mxv[FW_NR], Mr[FW_NR][2], t6[6];
  const int c0 = lane, c1 = lane + 64;
  const int cn0 = c0 < n ? c0 : n - 1, cn1 = c1 < n ? c1 : n - 1, cz0 = c0 < n + m ? c0 : n + m - 1, cz1 = c1 < n + m ? c1 : n + m - 1;
  auto load_K = [&](int k) {
    const double* g = gain_ptr(a, b, k);
#pragma unroll
    for (int q = 0; q < FW_KR; ++q) {
      const int r = wv + nw * q, rr = r < m ? r : 0;
      Kr[q][0] = g[L.oK + rr * n + cn0]; Kr[q][1] = g[L.oK + rr * n + cn1]; kf[q] = g[L.ok + rr];
    }
  };
  auto load_AB = [&](int k) {
    const double* g = gain_ptr(a, b, k);
    const double* kn = knot_ptr(a, b, k);
#pragma unroll
    for (int q = 0; q < FW_NR; ++q) {
      const int r = wv + nw * q, rr = r < n ? r : 0;
      ABr[q][0] = kn[L.oAB + rr * nz + cz0]; ABr[q][1] = kn[L.oAB + rr * nz + cz1]; mxv[q] = g[L.omx + rr];
    }
  };
  auto load_M = [&](int k) {
    const double* g = gain_ptr(a, b, k);
#pragma unroll
    for (int q = 0; q < FW_NR; ++q) {
      const int r = wv + nw * q, rr = r < n ? r : 0;
      Mr[q][0] = g[L.oMx + rr * n + cn0]; Mr[q][1] = g[L.oMx + rr * n + cn1];
    }
  };
  auto load_T6 = [&](int k) {
    const double* g = gain_ptr(a, b, k);
#pragma unroll
    for (int l = 0; l < 6; ++l) t6[l] = g[L.oT6 + (tid < 6 ? tid : 0) * 6 + l];
  };
  load_K(0); load_AB(0); load_M(0); load_T6(0);
  for (int i = tid; i < n; i += blockDim.x) { dz[i] = 0.0; a.dxs[(size_t)b * (N + 1) * n + i] = 0.0; }
  FW_BARRIER();
  for (int k = 0; k < N; ++k) {
    const bool more = k + 1 < N;
#pragma unroll
    for (int q = 0; q < FW_KR; ++q) {
      const int r = wv + nw * q;
      if (r < m) {
        double s = (c0 < n ? Kr[q][0] * dz[c0] : 0.0) + (c1 < n ? Kr[q][1] * dz[c1] : 0.0);
        s = wave_sum(s);
        if (lane == 0) { s += kf[q]; dz[n + r] = s; a.dus[((size_t)b * N + k) * m + r] = s; }
      }
    }
    if (more) load_K(k + 1);
    FW_BARRIER();
#pragma unroll
    for (int q = 0; q < FW_NR; ++q) {
      const int r = wv + nw * q;
      if (r < n) {
        double s = (c0 < n + m ? ABr[q][0] * dz[c0] : 0.0) + (c1 < n + m ? ABr[q][1] * dz[c1] : 0.0);
        s = wave_sum(s);
        if (lane == 0) { y[r] = s + mxv[q]; if (a.abdz) a.abdz[((size_t)b * N + k) * n + r] = s; }
      }
    }
    if (more) load_AB(k + 1);
    FW_BARRIER();
#pragma unroll
    for (int q = 0; q < FW_NR; ++q) {
      const int r = wv + nw * q;
      if (r < n) {
        double s = (c0 < n ? Mr[q][0] * y[c0] : 0.0) + (c1 < n ? Mr[q][1] * y[c1] : 0.0);
        s = wave_sum(s);
        if (lane == 0) z[r] = y[r] - mud * s;
      }
    }
    if (more) load_M(k + 1);
    FW_BARRIER();
    if (tid < n) {
      double s = z[tid];
      if (ff && tid < 6) { s = 0; for (int l = 0; l < 6; ++l) s += t6[l] * z[l]; }
      dz[tid] = s;
      a.dxs[((size_t)b * (N + 1) + k + 1) * n + tid] = s;
    }
    if (more) load_T6(k + 1);
    FW_BARRIER();
  }
}

// ------------------------------------------------------------------------------------------------
// P7 (cont.): multiplier steps and the merit directional derivative, parallel over knots.
// grid (N+1, B), block 256, LDS (nz + 3 n + 16 + 3 c) doubles
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_duals(SolverArgs a) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wv = tid >> 6, nw = nthr >> 6;
  const InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  const int n = L.n, nz = L.nz, N = L.N;
  double* kn = knot_ptr(a, b, k);
  const double* g = gain_ptr(a, b, k);
  const int c = (int)kn[L.oMISC + MISC_NC], m = (int)kn[L.oMISC + MISC_M];
  const double mu = st.mu, mud = mu * a.opt.dyn_al_scale;
  const bool ff = L.space == MPC_SPACE_MULTIBODY;
  const double* v = a.vs + ((size_t)b * (N + 1) + k) * L.c;
  double* dv = a.dvs + ((size_t)b * (N + 1) + k) * L.c;
  // one matrix row per wavefront, lanes across the columns (coalesced), DPP reductions; the step [dx; du] in LDS
  extern __shared__ __attribute__((aligned(16))) double sh[];
  double* dz = sh;            // nz
  double* dxn = dz + nz;      // n
  double* lnew = dxn + n;     // n
  double* jdl = lnew + n;     // n
  double* part = jdl + n;     // nw + 1 partial sums of the directional derivative
  for (int z = tid; z < n + m; z += nthr) dz[z] = (z < n) ? a.dxs[((size_t)b * (N + 1) + k) * n + z] : a.dus[((size_t)b * N + k) * L.m + z - n];
  if (k < N) for (int z = tid; z < n; z += nthr) dxn[z] = a.dxs[((size_t)b * (N + 1) + k + 1) * n + z];
  __syncthreads();
  // per-row scalars of the constraint block (active flag, shifted value / mu, multiplier): one coalesced load per thread up front — read
  // row by row from global memory they are three dependent round trips per row, and most rows are inactive (nothing else to do)
  double* rsc = part + nw + 2;  // [3][L.c]
  for (int i = tid; i < L.c; i += nthr) {
    const bool in = i < c;
    rsc[i] = in ? kn[L.oACT + i] : 0.0; rsc[L.c + i] = in ? kn[L.oDT + i] / mu : 0.0; rsc[2 * L.c + i] = in ? v[i] : 0.0;
  }
  __syncthreads();
  double acc = 0.0;  // lane 0 of every wavefront accumulates the rows of that wavefront
  if (wv == 0) {     // cost gradient part
    double s = 0;
    for (int z = lane; z < n + m; z += 64) s += kn[L.oG + z] * dz[z];
    acc += wave_sum(s);
  }
  // constraints
  for (int i = wv; i < L.c; i += nw) {
    if (i >= c) { if (lane == 0) dv[i] = 0.0; continue; }
    double s = 0, jd = 0;
    const bool act = rsc[i] != 0.0;
    const double vp = rsc[L.c + i], vi = rsc[2 * L.c + i];
    const double wj = vp + (act ? (vp - vi) : 0.0);  // weight of the row's directional derivative in the merit slope
    if (act)  // the sweep writes the dual gains of active rows only (inactive: dv = -v)
      for (int z = lane; z < n; z += 64) s += g[L.oKnu + i * n + z] * dz[z];
    if (wj != 0.0)  // an inactive row with a zero projection does not enter the slope: its Jacobian row is not read
      for (int z = lane; z < n + m; z += 64) jd += kn[L.oCD + i * nz + z] * dz[z];
    s = wave_sum(s) + g[L.oknu + i];
    jd = wave_sum(jd);
    const double dvi = s - vi;
    if (lane == 0) dv[i] = dvi;
    acc += wj * jd - mu * (vp - vi) * dvi;
  }
  if (k == 0) for (int i = tid; i < n; i += nthr) a.dlams[(size_t)b * (N + 1) * n + i] = 0.0;
  if (k < N) {
    const double* gn = gain_ptr(a, b, k + 1);
    // [A B] [dx; du]: left by the three-mat-vec forward sweep (abdz) ; without it (closed-loop / leg sweeps) the linearised
    // dynamics residual comes from the multiplier update the LQ step satisfies, A dx + B du + E dx' = mu_d (lambda' - lambda_e) - f
    // (below): the 66 KB of [A B] per knot are not read again
    const bool have = a.abdz != nullptr;
    // lambda' = P' dx' + p': a wavefront takes four rows of P' at a time, all their loads in flight before the reductions (n <= 128)
    const int c0 = lane < n ? lane : 0, c1 = lane + 64 < n ? lane + 64 : 0;
    const double d0 = lane < n ? dxn[c0] : 0.0, d1 = lane + 64 < n ? dxn[c1] : 0.0;
    if (n > 128) {  // general form (no kernel of the whole-body / centroidal problems gets here)
      for (int i = wv; i < n; i += nw) {
        double s = 0;
        for (int z = lane; z < n; z += 64) s += gn[L.oP + i * n + z] * dxn[z];
        s = wave_sum(s);
        if (lane == 0) { lnew[i] = s + gn[L.op + i]; jdl[i] = have ? a.abdz[((size_t)b * N + k) * n + i] : 0.0; }
      }
    } else
    for (int r0 = 4 * wv; r0 < n; r0 += 4 * nw) {
      double pv[4][2];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int r = r0 + q < n ? r0 + q : 0; pv[q][0] = gn[L.oP + r * n + c0]; pv[q][1] = gn[L.oP + r * n + c1]; }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = r0 + q;
        const double s = wave_sum(pv[q][0] * d0 + pv[q][1] * d1);
        if (lane == 0 && i < n) { lnew[i] = s + gn[L.op + i]; jdl[i] = have ? a.abdz[((size_t)b * N + k) * n + i] : 0.0; }
      }
    }
  }
  if (lane == 0) part[wv] = acc;
  __syncthreads();
  if (k < N && wv == 0) {
    const double* lam = a.lams + ((size_t)b * (N + 1) + k + 1) * n;
    const double* le = a.lams_e + ((size_t)b * (N + 1) + k + 1) * n;
    double* dl = a.dlams + ((size_t)b * (N + 1) + k + 1) * n;
    double s = 0;
    for (int i = lane; i < n; i += 64) {
      double ln = lnew[i];
      if (ff && i < 6) { ln = 0; for (int l = 0; l < 6; ++l) ln += g[L.oT6 + l * 6 + i] * lnew[l]; }
      const double dli = ln - lam[i];
      dl[i] = dli;
      const double lp = le[i] + kn[L.oF + i] / mud;
      double jd = jdl[i];
      if (a.abdz == nullptr) jd = mud * (ln - le[i]) - kn[L.oF + i];
      else if (ff && i < 6) { for (int l = 0; l < 6; ++l) jd += kn[L.oE6 + i * 6 + l] * dxn[l]; }
      else jd -= dxn[i];
      s += (2.0 * lp - lam[i]) * jd - mud * (lp - lam[i]) * dli;
    }
    s = wave_sum(s);
    if (lane == 0) part[nw] = s;
  }
  __syncthreads();
  if (tid == 0) { double r = 0; for (int i = 0; i < nw; ++i) r += part[i]; if (k < N) r += part[nw]; kn[L.oMISC + MISC_DMERIT] = r; }
}

// ------------------------------------------------------------------------------------------------
// P8: Armijo backtracking over the pre-evaluated candidates alpha_i = 2^-i.  grid B, block 64
// ------------------------------------------------------------------------------------------------
// first == 1: only the full step (candidate 0) has been evaluated; accept it if it passes Armijo, otherwise
// raise ls_more so that the remaining candidates get evaluated.  first == 0: backtrack over the candidates 0 .. upto evaluated so far
// (two launches: alpha = 1/2, 1/4 first — most backtracking ends there — then the rest for the instances still undecided).
// one wavefront per instance: the sums over the knots are taken by the lanes (two knots per lane, then the fixed DPP tree of wave_sum:
// deterministic), not by one thread walking 101 records
__global__ void __launch_bounds__(64) k_linesearch(SolverArgs a, int first, int upto) {
  const Layout& L = a.L;
  const int b = blockIdx.x, lane = threadIdx.x;
  InstState& st = a.inst[b];
  if (st.done || st.skip_step) return;
  if (!first && !st.ls_more) return;
  auto knot_sum = [&](auto&& value_of) {  // sum over k = 0 .. N of value_of(k)
    double s = 0.0;
    for (int k = lane; k <= L.N; k += 64) s += value_of(k);
    return wave_sum(s);
  };
  const double d = knot_sum([&](int k) { return knot_ptr(a, b, k)[L.oMISC + MISC_DMERIT]; });
  double alpha = 1.0;
  int step = 0;
  if (first && fabs(d) <= MPC_STALL_TOL * (1.0 + fabs(st.phi0))) {
    // the Newton step cannot decrease the merit any further (round-off floor of the 1/mu-conditioned system): the inner
    // problem counts as solved — no step, no iteration counted, the next pass takes the BCL branch
    if (lane == 0) { st.dphi0 = d; st.stalled = (((st.stalled >> 8) + 1) << 8) | 1; st.ls_more = 0; st.alpha = 0.0; st.ls_step = 0; }
    return;
  }
  if (first) {
    const double* tp = a.trial_phi + ((size_t)b * L.n_alpha) * (L.N + 1);
    const double phi = knot_sum([&](int k) { return tp[k]; });
    const bool ok = phi <= st.phi0 + a.opt.ls_armijo_c1 * d;
    const bool last = a.opt.ls_max_steps <= 1 || L.n_alpha <= 1 || 0.5 < a.opt.ls_alpha_min;
    if (lane == 0) { st.dphi0 = d; st.ls_more = (ok || last) ? 0 : 1; st.alpha = 1.0; st.ls_step = 0; }
    return;
  }
  for (;; ++step) {
    const double* tp = a.trial_phi + ((size_t)b * L.n_alpha + step) * (L.N + 1);
    const double phi = knot_sum([&](int k) { return tp[k]; });
    if (phi <= st.phi0 + a.opt.ls_armijo_c1 * alpha * d) break;
    if (step + 1 >= a.opt.ls_max_steps || step + 1 >= L.n_alpha || 0.5 * alpha < a.opt.ls_alpha_min) { if (a.reject_failed) alpha = 0.0; break; }
    if (step >= upto) return;  // the next candidate is evaluated by the second backtracking launch: undecided, ls_more stays set
    alpha *= 0.5;
  }
  if (lane == 0) { st.dphi0 = d; st.alpha = alpha; st.ls_step = step; st.ls_more = 0; }
}

// ------------------------------------------------------------------------------------------------
// mpc_options.refine_appended_knot (include/mpc_abi.h): warm start of the knot mpc_cycle appended, made consistent with its own stage.
// grid B, block 256.  mode 0: one Newton step on u_{N-1} alone from the record the stage kernel has just written for knot N - 1
// (x_{N-1} fixed), the stage KKT system of that knot with the controls eliminated first (as the Riccati sweep does):
//     Huu = L L^T ;  Y = L^-1 D_a^T ;  S = Y^T Y + rho I ;  nu = S^-1 (Pi_N(z)_a - Y^T L^-1 g_u) ;  du = -L^-T (L^-1 g_u + Y nu)
// rho = max(mu, 1e-8 max diag(Y^T Y)): see oracle/solver.hpp refine_appended_knot (same arithmetic) for why not mu itself.
// An indefinite block or more than MPC_REFINE_MP active rows leave u as it is.  mode 1: x_N = phi(x_{N-1}, u_{N-1}) from the record.
// ------------------------------------------------------------------------------------------------
#define MPC_REFINE_MP 48
// Cholesky of the n x n matrix A (leading dimension ld, lower triangle in place) by the whole workgroup; *ok = 0 if it is not positive definite
DEV void refine_chol(double* A, int ld, int n, int tid, int nthr, int* ok) {
  for (int j = 0; j < n; ++j) {
    if (tid == 0) { const double d = A[j * ld + j]; if (d > 0.0) A[j * ld + j] = sqrt(d); else *ok = 0; }
    __syncthreads();
    if (!*ok) return;
    const double dj = A[j * ld + j];
    for (int i = j + 1 + tid; i < n; i += nthr) A[i * ld + j] /= dj;
    __syncthreads();
    const int r = n - j - 1;
    for (int idx = tid; idx < r * r; idx += nthr) {
      const int i = j + 1 + idx / r, k2 = j + 1 + idx % r;
      if (k2 <= i) A[i * ld + k2] -= A[i * ld + j] * A[k2 * ld + j];
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) k_refine_knot(SolverArgs a, int mode) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
  const InstState& st = a.inst[b];
  if (st.done || st.converged < 0) return;
  const int N = L.N, n = L.n, nz = L.nz;
  const double* kn = knot_ptr(a, b, N - 1);
  if (mode == 1) {
    double* xN = a.xs + ((size_t)b * (N + 1) + N) * L.nx;
    for (int i = tid; i < L.nx; i += nthr) xN[i] = kn[L.oXN + i];
    return;
  }
  const int m = (int)kn[L.oMISC + MISC_M], c = (int)kn[L.oMISC + MISC_NC];
  if (m <= 0 || m > MPC_REFINE_MP) return;
  constexpr int ld = MPC_REFINE_MP + 1;
  __shared__ double A[MPC_REFINE_MP * ld], Y[MPC_REFINE_MP * ld], S[MPC_REFINE_MP * ld], w[MPC_REFINE_MP], nu[MPC_REFINE_MP], v[MPC_REFINE_MP];
  __shared__ int act[MPC_REFINE_MP], ca_s, ok;
  if (tid == 0) {
    int ca = 0;
    for (int q = 0; q < c; ++q) if (kn[L.oACT + q] != 0.0) { if (ca < MPC_REFINE_MP) act[ca] = q; ++ca; }
    ca_s = ca; ok = 1;
  }
  for (int idx = tid; idx < m * m; idx += nthr) {
    const int i = idx / m, j = idx % m;
    A[i * ld + j] = 0.5 * (kn[L.oH + (n + i) * nz + n + j] + kn[L.oH + (n + j) * nz + n + i]);
  }
  for (int i = tid; i < m; i += nthr) w[i] = kn[L.oG + n + i];
  __syncthreads();
  const int ca = ca_s;
  if (ca > MPC_REFINE_MP) return;
  for (int idx = tid; idx < m * ca; idx += nthr) { const int i = idx / ca, q = idx % ca; Y[i * ld + q] = kn[L.oCD + act[q] * nz + n + i]; }
  for (int q = tid; q < ca; q += nthr) v[q] = kn[L.oDT + act[q]];
  refine_chol(A, ld, m, tid, nthr, &ok);
  if (!ok) return;
  // forward substitutions L [Y | w] = [D_a^T | g_u]: a thread per right-hand side
  if (tid <= ca) {
    const int q = tid;
    for (int i = 0; i < m; ++i) {
      double t = (q < ca) ? Y[i * ld + q] : w[i];
      for (int k2 = 0; k2 < i; ++k2) t -= A[i * ld + k2] * ((q < ca) ? Y[k2 * ld + q] : w[k2]);
      t /= A[i * ld + i];
      if (q < ca) Y[i * ld + q] = t; else w[i] = t;
    }
  }
  __syncthreads();
  if (ca > 0) {
    for (int idx = tid; idx < ca * ca; idx += nthr) {
      const int p = idx / ca, q = idx % ca;
      double t = 0;
      for (int i = 0; i < m; ++i) t += Y[i * ld + p] * Y[i * ld + q];
      S[p * ld + q] = t;
    }
    __syncthreads();
    if (tid == 0) {
      double dmax = 0.0;
      for (int q = 0; q < ca; ++q) dmax = fmax(dmax, S[q * ld + q]);
      const double rho = fmax(st.mu, 1e-8 * dmax);
      for (int q = 0; q < ca; ++q) S[q * ld + q] += rho;
    }
    for (int q = tid; q < ca; q += nthr) { double t = v[q]; for (int i = 0; i < m; ++i) t -= Y[i * ld + q] * w[i]; nu[q] = t; }
    __syncthreads();
    refine_chol(S, ld, ca, tid, nthr, &ok);
    if (!ok) return;
    if (tid == 0) {
      for (int i = 0; i < ca; ++i) { double t = nu[i]; for (int k2 = 0; k2 < i; ++k2) t -= S[i * ld + k2] * nu[k2]; nu[i] = t / S[i * ld + i]; }
      for (int i = ca - 1; i >= 0; --i) { double t = nu[i]; for (int k2 = i + 1; k2 < ca; ++k2) t -= S[k2 * ld + i] * nu[k2]; nu[i] = t / S[i * ld + i]; }
    }
    __syncthreads();
    for (int i = tid; i < m; i += nthr) { double t = w[i]; for (int q = 0; q < ca; ++q) t += Y[i * ld + q] * nu[q]; w[i] = t; }
    __syncthreads();
  }
  if (tid == 0) {
    for (int i = m - 1; i >= 0; --i) { double t = w[i]; for (int k2 = i + 1; k2 < m; ++k2) t -= A[k2 * ld + i] * w[k2]; w[i] = t / A[i * ld + i]; }
    bool fin = true;
    for (int i = 0; i < m; ++i) fin = fin && isfinite(w[i]);
    ok = fin ? 1 : 0;
  }
  __syncthreads();
  if (!ok) return;
  double* u = a.us + ((size_t)b * N + N - 1) * L.m;
  for (int i = tid; i < m; i += nthr) u[i] -= w[i];
}

// ------------------------------------------------------------------------------------------------
// P9: accept the step.  grid (N+1, B), block 64
// ------------------------------------------------------------------------------------------------
__global__ void k_accept(SolverArgs a) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  InstState& st = a.inst[b];
  if (st.done || st.skip_step || (st.stalled & 1)) return;
  const int n = L.n, N = L.N, nx = L.nx, m = L.m;
  const double alpha = st.alpha;
  if (alpha == 0.0) return;  // (a rejected step, MPC_HIP_REJECT_FAILED: the iterate stays bit for bit)
  double* x = a.xs + ((size_t)b * (N + 1) + k) * nx;
  const double* dx = a.dxs + ((size_t)b * (N + 1) + k) * n;
  __shared__ double xn[160];
  state_integrate_group(L.space, nx, n, x, dx, alpha, xn, tid, nthr);
  __syncthreads();
  for (int i = tid; i < nx; i += nthr) x[i] = xn[i];
  double* v = a.vs + ((size_t)b * (N + 1) + k) * L.c;
  const double* dv = a.dvs + ((size_t)b * (N + 1) + k) * L.c;
  for (int i = tid; i < L.c; i += nthr) v[i] += alpha * dv[i];
  double* lam = a.lams + ((size_t)b * (N + 1) + k) * n;
  const double* dl = a.dlams + ((size_t)b * (N + 1) + k) * n;
  for (int i = tid; i < n; i += nthr) lam[i] += alpha * dl[i];
  if (k < N) {
    double* u = a.us + ((size_t)b * N + k) * m;
    const double* du = a.dus + ((size_t)b * N + k) * m;
    for (int i = tid; i < m; i += nthr) u[i] += alpha * du[i];
  }
}

// bookkeeping after a step.  grid B, block 1
__global__ void k_after_step(SolverArgs a) {
  InstState& st = a.inst[blockIdx.x];
  if (a.isolate && st.done >= 2 && st.converged >= 0) st.converged = -st.done;  // a failed factorisation: remembered across runs (mpc_set_failure_policy)
  if (st.done) return;
  const bool stepped = !st.skip_step && !(st.stalled & 1);
  if (stepped) {
    st.stalled = 0;
    st.num_iters += 1;
    // mpc_options.corrector_prim_tol: the budget ends with an iteration that started from an iterate infeasible by more than the tolerance
    // (st.prim: what k_decide of this pass measured), or whose step was shortened — one more iteration, once per run (oracle/solver.hpp run_instance: the same rule)
    if (a.corrector_on && !st.corrector && st.num_iters >= a.opt.max_iters && (st.prim > a.opt.corrector_prim_tol || st.alpha < 1.0)) st.corrector = 1;
    if (st.num_iters >= a.opt.max_iters + st.corrector) st.done = 1;
  }
  // the records now hold the evaluation of the accepted iterate iff the full step (the one evaluated with derivatives) was taken (1) ;
  // after a BCL update without a step (skip_step: no candidate was evaluated) they still hold the evaluation of the unchanged iterate
  // (2: as 1, but there is no speculative record of the knot the next tick appends) — the next pass only re-projects them
  if (a.spec) a.spec[blockIdx.x] = !a.spec_on ? 0 : (stepped && st.alpha == 1.0) ? 1 : (!stepped && st.skip_step) ? 2 : 0;
}

// setup(): multipliers and their estimates to zero, fresh per-instance solver state (mu = mu_init).  One launch instead
// of four fills, an upload and a stream sync: the MPC loop calls setup every tick.  grid (N + 2, B), block 256
__global__ void k_setup(SolverArgs a) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t lrow = ((size_t)b * (L.N + 2) + k) * L.n;
  for (int i = tid; i < L.n; i += blockDim.x) { a.lams[lrow + i] = 0.0; a.lams_e[lrow + i] = 0.0; }
  if (k <= L.N) {
    const size_t vrow = ((size_t)b * (L.N + 1) + k) * L.c;
    for (int i = tid; i < L.c; i += blockDim.x) { a.vs[vrow + i] = 0.0; a.vs_e[vrow + i] = 0.0; }
  }
  if (k == 0 && tid == 0) {
    const int failed = (a.isolate && a.inst[b].converged < 0) ? a.inst[b].converged : 0;  // (stays out until mpc_revive_instance)
    InstState z;
    memset(&z, 0, sizeof(z));
    z.mu = a.opt.mu_init;
    z.converged = failed; z.done = -failed;
    a.inst[b] = z;
  }
}

// start of run(): xs[0] = x0, tolerances of the BCL loop.  grid B, block 64
__global__ void k_begin_run(SolverArgs a) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x;
  InstState& st = a.inst[b];
  if (a.opt.force_initial_condition)
    for (int i = tid; i < L.nx; i += blockDim.x) a.xs[(size_t)b * (L.N + 1) * L.nx + i] = a.x0[(size_t)b * L.nx + i];
  if (tid == 0 && a.isolate && st.converged < 0) { st.done = -st.converged; st.num_iters = 0; st.skip_step = 0; }  // a failed instance sits this run out
  else if (tid == 0) {
    st.num_iters = 0; st.al_iters = 0; st.converged = 0; st.done = 0; st.skip_step = 0; st.ls_step = 0; st.alpha = 0;
    st.stalled = 0; st.ls_more = 0; st.corrector = 0;
    st.prim_tol = fmax(a.opt.prim_tol0 * pow(st.mu, a.opt.bcl_prim_alpha), a.opt.tol);
    st.inner_tol = fmax(a.opt.inner_tol0 * pow(st.mu, a.opt.bcl_dual_alpha), a.opt.tol);
  }
}

// warm-start shift on the device (fulldynamic_talos.py:532-534): xs <- [xs[1:], xs[-1]], us <- [us[1:], us[-1]], out of place (the
// host swaps the two buffer pairs), one workgroup per knot.  grid (N + 1, B), block 64
__global__ void k_shift(SolverArgs a, const double* xs_in, const double* us_in, int perfect_feedback) {
  const Layout& L = a.L;
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, nthr = blockDim.x;
  const double* xi = xs_in + ((size_t)b * (L.N + 1) + (k < L.N ? k + 1 : L.N)) * L.nx;
  double* xo = a.xs + ((size_t)b * (L.N + 1) + k) * L.nx;
  for (int i = tid; i < L.nx; i += nthr) {
    const double v = xi[i];
    xo[i] = v;
    if (k == 0 && perfect_feedback) a.x0[(size_t)b * L.nx + i] = v;  // the predicted next state is the new measurement
  }
  if (k < L.N) {
    const double* ui = us_in + ((size_t)b * L.N + (k + 1 < L.N ? k + 1 : L.N - 1)) * L.m;
    double* uo = a.us + ((size_t)b * L.N + k) * L.m;
    for (int i = tid; i < L.m; i += nthr) uo[i] = ui[i];
  }
}


// ------------------------------------------------------------------------------------------------
// mpc_walk_update (include/mpc_abi.h): the swing-foot reference generator of the walking loops for every instance of an ensemble with per-instance
// parameter tables.  grid B, block 128.  On a replanning tick two threads run the forward kinematics of the two sole frames at the instance's
// predicted next state xs[1] and one applies the foothold rules (walk_generator.h) to the instance's plan; then a thread per knot forms the two
// placement references (Bezier swing curve, geodesic between the rotations) and writes them into the instance's table of that knot — ring-indexed
// as the stage tables are, BEFORE the rotation of this tick — and the owner of knot N - 1 the terminal node's targets.
// ------------------------------------------------------------------------------------------------
#include "walk_generator.h"
__global__ void __launch_bounds__(128) k_walk_refs(SolverArgs a, mpc_walk_config c, double* state, int takeoff_RF, int takeoff_LF, int land_RF, int land_LF, int replanning, int write_all) {
  const Layout& L = a.L;
  const int b = blockIdx.x, tid = threadIdx.x, N = L.N;
  __shared__ double st[48], meas[24];
  double* gst = state + (size_t)b * 48;
  if (replanning) {
    if (tid < 2) {
      M3 R; V3 p;
      const double* q = a.xs + ((size_t)b * (N + 1) + 1) * L.nx;
      walk_frame_placement(a.model_i, a.model_d, q, tid == 0 ? c.frame_lf : c.frame_rf, R, p);
      walk_pose_store(meas + 12 * tid, R, p);
    }
    if (tid >= 64 && tid < 112) st[tid - 64] = gst[tid - 64];
    __syncthreads();
    if (tid == 0) walk_plan(st, meas, meas + 12, takeoff_RF, takeoff_LF, land_RF, land_LF, c.T_ds, c.t_left, c.t_right, c.rot_diff, c.floor_z);
    __syncthreads();
    if (tid < 48) gst[tid] = st[tid];
  } else {
    if (tid < 48) st[tid] = gst[tid];
    __syncthreads();
  }
  double* tables = const_cast<double*>(a.inst_params) + (size_t)b * (N + 1) * L.max_stage_doubles;
  for (int j = (write_all ? 0 : N - 1) + tid; j < N; j += blockDim.x) {
    double Lr[12], Rr[12];
    walk_ref(Lr, st, st + 12, land_LF, j, c.T_ss, c.swing_apex);
    walk_ref(Rr, st + 24, st + 36, land_RF, j, c.T_ss, c.swing_apex);
    double* tab = tables + (size_t)stage_slot(a, j) * L.max_stage_doubles;
    if (c.off_lf >= 0) for (int e = 0; e < 12; ++e) tab[c.off_lf + e] = Lr[e];
    if (c.off_rf >= 0) for (int e = 0; e < 12; ++e) tab[c.off_rf + e] = Rr[e];
    if (c.off_xref_z >= 0 && c.z_follow != 0.0) tab[c.off_xref_z] = c.xref_z0 + 0.5 * (Lr[11] + Rr[11]) - c.feet_z0;
    if (j == N - 1) {  // terminal node: the last foot references and the CoM target between them
      double* tt = tables + (size_t)N * L.max_stage_doubles;
      if (c.toff_com >= 0) {
        tt[c.toff_com] = 0.5 * (Lr[9] + Rr[9]); tt[c.toff_com + 1] = 0.5 * (Lr[10] + Rr[10]);
        tt[c.toff_com + 2] = c.com0[2] + (c.z_follow != 0.0 ? 0.5 * (Lr[11] + Rr[11]) - c.feet_z0 : 0.0);
      }
      if (c.toff_lf >= 0) for (int e = 0; e < 12; ++e) tt[c.toff_lf + e] = Lr[e];
      if (c.toff_rf >= 0) for (int e = 0; e < 12; ++e) tt[c.toff_rf + e] = Rr[e];
    }
  }
}
