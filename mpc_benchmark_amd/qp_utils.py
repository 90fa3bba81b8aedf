"""Whole-body inverse-dynamics QP of the reference's 1 kHz loop on the batched QP solver ("next" row N3).

``IDSolver_ulim`` keeps the constructor and ``solve`` signature of QP_utils.py:437-575 (used at kinodynamic_talos.py:351,
421-446): unknowns ``x = (da, df, tau)`` around the MPC's acceleration ``a`` and contact forces,

    M (a + da) + nle = S tau + Jc^T (f + df)          dynamics
    Jc (a + da) + gamma = 0                           contacts do not accelerate (with velocity damping)
    Cmin (f + df) >= 0                                friction cone, unilaterality, CoP inside the sole
    min  w0 |da|^2 + w1 |df|^2

solved with eps_abs = 1e-3, max_iter = max_iter_in = 10 as there.  The rigid-body terms come from
``mpc_benchmark_amd.robot.dynamics`` (numpy) instead of Pinocchio; ``data`` must have gone through
``dynamics.compute_all_terms(model, data, q, v)``.  The QP itself runs on the GPU (HIP library; pass ``library=`` to use the
oracle in tests).  ``solve_batch`` solves the QPs of several robots in one launch.
"""
from __future__ import annotations

import numpy as np

from ._qp_capi import BatchedQP
from .robot import dynamics as dyn


def wrench_cone_bound_rows(mu, L, W, force_size=6):
    """The rows that form the lower bound l = - rows @ f.  The reference writes l out by hand (QP_utils.py:538-548) with rows 2, 3 =
    f_y -+ mu f_z, while rows 2, 3 of its ``Cmin`` repeat the f_x rows (:466-490): C and l are kept exactly as the reference has them
    (a drop-in, not a repaired cone: the lateral friction |f_y| <= mu f_z is NOT constrained by that QP either)."""
    R = wrench_cone_rows(mu, L, W, force_size)
    R[2, :2], R[3, :2] = (0.0, -1.0), (0.0, 1.0)
    return R


def wrench_cone_rows(mu, L, W, force_size=6):
    """The 9 rows per contact of QP_utils.py:466-490: |fx|, |fy| <= mu fz ; fz >= 0 ; |tau_x| <= W fz ; |tau_y| <= L fz."""
    if force_size == 3:
        return np.array([[-1, 0, mu], [1, 0, mu], [-1, 0, mu], [1, 0, mu], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1]], dtype=float)
    return np.array([[-1, 0, mu, 0, 0, 0], [1, 0, mu, 0, 0, 0], [-1, 0, mu, 0, 0, 0], [1, 0, mu, 0, 0, 0], [0, 0, 1, 0, 0, 0],
                     [0, 0, W, -1, 0, 0], [0, 0, W, 1, 0, 0], [0, 0, L, 0, -1, 0], [0, 0, L, 0, 1, 0]], dtype=float)


class IDSolver_ulim:
    def __init__(self, model, weights, nk, mu, L, W, contact_ids, force_size, verbose=False, library=None, batch=1, warm_start=False):
        """``warm_start``: from the second solve on, start from the previous solution (the 1 kHz loop solves a slowly
        varying sequence of QPs; ProxQP's own default is a cold equality-constrained start)."""
        self.warm_start = bool(warm_start)
        self.model, self.nk, self.contact_ids, self.mu, self.L, self.W, self.force_size = model, nk, list(contact_ids), mu, L, W, force_size
        self.baum_Kd = np.eye(3)  # velocity damping of the contact point (kd = 1)
        nv, fs = model.nv, force_size
        self.n, self.neq, self.nin = 2 * nv - 6 + fs * nk, nv + fs * nk, 9 * nk
        self.S = np.zeros((nv, nv - 6)); self.S[6:] = np.eye(nv - 6)
        self.Cmin = wrench_cone_rows(mu, L, W, fs)
        self.Cl = wrench_cone_bound_rows(mu, L, W, fs)
        self.H = np.zeros((self.n, self.n))
        self.H[:nv, :nv] = np.eye(nv) * weights[0]
        self.H[nv:nv + fs * nk, nv:nv + fs * nk] = np.eye(fs * nk) * weights[1]
        self.g = np.zeros(self.n)
        self.u = np.full(self.nin, 1e5)
        self.batch = int(batch)
        self.qp = BatchedQP(self.batch, self.n, self.neq, self.nin, box=False, library=library)
        self.qp.settings.eps_abs, self.qp.settings.max_iter, self.qp.settings.max_iter_in = 1e-3, 10, 10
        self.verbose = verbose
        self.last_info = None

    def computeMatrice(self, data, cs, v, a, forces, M):
        """-> A, b, C, l of one robot (same blocks as the reference assembles in place)."""
        nv, fs, nk = self.model.nv, self.force_size, self.nk
        Jc = np.zeros((nk * fs, nv)); gamma = np.zeros(nk * fs)
        for i in range(nk):
            if cs[i]:
                fid = self.contact_ids[i]
                Jc[i * fs:(i + 1) * fs] = dyn.frame_jacobian_local(self.model, data, fid)[:fs]
                gamma[i * fs:(i + 1) * fs] = dyn.frame_jdot_v_local(self.model, data, fid)[:fs]
                vel = dyn.frame_velocity_local(self.model, data, fid)
                gamma[i * fs:i * fs + 3] += self.baum_Kd @ vel.linear + self.baum_Kd @ vel.angular
        A = np.zeros((self.neq, self.n)); b = np.zeros(self.neq)
        A[:nv, :nv] = M; A[:nv, nv:nv + nk * fs] = -Jc.T; A[:nv, nv + nk * fs:] = -self.S; A[nv:, :nv] = Jc
        b[:nv] = -data.nle - M @ a + Jc.T @ forces
        b[nv:] = -gamma - Jc @ a
        C = np.zeros((self.nin, self.n)); l = np.zeros(self.nin)
        for i in range(nk):
            if cs[i]:
                l[9 * i:9 * (i + 1)] = -self.Cl @ forces[i * fs:(i + 1) * fs]
                C[9 * i:9 * (i + 1), nv + i * fs:nv + (i + 1) * fs] = self.Cmin
        return A, b, C, l

    def solve(self, data, cs, v, a, forces, M):
        """-> (a_new, new_forces, torque) as QP_utils.py:553-575."""
        out = self.solve_batch([(data, cs, v, a, forces, M)] + [None] * (self.batch - 1))
        return out[0]

    def solve_batch(self, items):
        """``items``: up to ``batch`` tuples (data, cs, v, a, forces, M); missing entries repeat the first problem."""
        mats = [self.computeMatrice(*it) if it is not None else None for it in items]
        first = next(m for m in mats if m is not None)
        mats = [m if m is not None else first for m in mats] + [first] * (self.batch - len(mats))
        A = np.stack([m[0] for m in mats]); b = np.stack([m[1] for m in mats]); C = np.stack([m[2] for m in mats]); l = np.stack([m[3] for m in mats])
        x, y, z, _, info = self.qp.solve(self.H, self.g, A, b, C, l, self.u)
        self.last_info = info
        if self.warm_start:
            self.qp.settings.warm_start = 1
        nv, fs, nk = self.model.nv, self.force_size, self.nk
        res = []
        for i, it in enumerate(items):
            if it is None:
                res.append(None)
                continue
            _, _, _, a, forces, _ = it
            res.append((a + x[i, :nv], forces + x[i, nv:nv + fs * nk], x[i, nv + fs * nk:].copy()))
        return res


    # ---- the whole QP on the device: no Pinocchio-like terms computed on the host, no matrices over PCIe ----
    def enable_device_assembly(self):
        """Upload the robot model (the tables of mpc_set_model, with this solver's contact frames registered) so that
        ``solve_batch_device`` builds M, nle, Jc, gamma and the QP matrices in one kernel per batch (csrc/qp_assemble.h)."""
        from .aligator._core import LoweringContext
        if self.force_size != 6:
            raise NotImplementedError("device assembly is written for 6-D contact wrenches")
        ctx = LoweringContext()
        self._frame_idx = np.array([ctx.frame_index(self.model, fid) for fid in self.contact_ids], dtype=np.int32)
        self.qp.set_model(*ctx.model_tables())
        self._weights = np.array([self.H[0, 0], self.H[self.model.nv, self.model.nv]])

    def solve_batch_device(self, x, a, forces, cs, return_matrices=False):
        """``x`` [B][nq+nv], ``a`` [B][nv], ``forces`` [B][6 nk], ``cs`` [B][nk] -> (a_new, new_forces, torque), each [B][...]."""
        if not hasattr(self, "_frame_idx"):
            self.enable_device_assembly()
        out = self.qp.solve_id(self._frame_idx, self._weights, self.Cmin, float(self.baum_Kd[0, 0]), x, a, forces, cs, return_matrices=return_matrices, cone_l=self.Cl)
        sol, info = out[0], out[3]
        self.last_info = info
        if self.warm_start:
            self.qp.settings.warm_start = 1
        nv, fs, nk = self.model.nv, self.force_size, self.nk
        a = np.broadcast_to(np.asarray(a, dtype=float), (self.batch, nv)); forces = np.broadcast_to(np.asarray(forces, dtype=float), (self.batch, fs * nk))
        res = (a + sol[:, :nv], forces + sol[:, nv:nv + fs * nk], sol[:, nv + fs * nk:].copy())
        return res + (out[4],) if return_matrices else res


    def low_level_steps(self, plan, sim, cs, tau_max, steps, dt, x=None):
        """``steps`` periods of the script's low-level loop (kinodynamic_talos.py:411-462) inside the library (mpc_qp_low_level_steps): ``plan`` / ``sim`` the
        NativeSolver handles of the MPC problem and of the simulator stand-in.  -> x_prev, x, torque, new_forces (of the last period)."""
        if not hasattr(self, "_frame_idx"):
            self.enable_device_assembly()
        x_prev, x_out, tau, forces, info = self.qp.low_level_steps(plan, sim, self._frame_idx, self._weights, self.Cmin, float(self.baum_Kd[0, 0]), cs, tau_max,
                                                                   steps, dt, x=x, cone_l=self.Cl)
        self.last_info = info
        if self.warm_start:
            self.qp.settings.warm_start = 1
        return x_prev, x_out, tau, forces


class IKIDSolver_f6:
    """Inverse kinematics + inverse dynamics in one QP (QP_utils.py:584-762, used at centroidal_talos.py:326, 435): unknowns
    ``x = (a, df, tau)``; tasks in the cost — posture (w0), foot accelerations (w1), centroidal momentum rate (w2), base and
    torso angular accelerations (w3), force increments (w4), each with PD gains ``K_gains[k] = (Kp, Kd)`` on the task errors —
    dynamics and contact-acceleration equalities, wrench-cone inequalities and the torque box ``|tau| <= effortLimit``.
    eps_abs = 1e-3, max_iter = max_iter_in = 100 as in the reference."""

    def __init__(self, model, weights, K_gains, nk, mu, L, W, contact_ids, base_id, torso_id, force_size, verbose=False, library=None, batch=1):
        self.model, self.weights, self.K_gains, self.nk = model, list(weights), K_gains, nk
        self.contact_ids, self.base_id, self.torso_id = list(contact_ids), base_id, torso_id
        self.mu, self.L, self.W, self.force_size = mu, L, W, force_size
        nv, fs = model.nv, force_size
        self.n, self.neq, self.nin = 2 * nv - 6 + fs * nk, nv + fs * nk, 9 * nk
        self.S = np.zeros((nv, nv - 6)); self.S[6:] = np.eye(nv - 6)
        self.Cmin = wrench_cone_rows(mu, L, W, fs)
        self.Cl = wrench_cone_bound_rows(mu, L, W, fs)
        self.l_box = np.full(self.n, -1e5); self.u_box = np.full(self.n, 1e5)
        self.l_box[nv + fs * nk:] = -np.asarray(model.effortLimit)[6:]
        self.u_box[nv + fs * nk:] = np.asarray(model.effortLimit)[6:]
        self.u = np.full(self.nin, 1e5)
        self.batch = int(batch)
        self.qp = BatchedQP(self.batch, self.n, self.neq, self.nin, box=True, library=library)
        self.qp.settings.eps_abs, self.qp.settings.max_iter, self.qp.settings.max_iter_in = 1e-3, 100, 100
        self.last_info = None

    def computeMatrice(self, data, cs, v, q_diff, dq_diff, LF_diff, dLF_diff, RF_diff, dRF_diff, base_diff, dbase_diff, torso_diff, dtorso_diff,
                       forces, dH, M):
        """-> H, g, A, b, C, l of one robot."""
        m, nv, fs, nk, w, K = self.model, self.model.nv, self.force_size, self.nk, self.weights, self.K_gains
        J = [dyn.frame_jacobian_local(m, data, fid) for fid in self.contact_ids]
        dJv = [dyn.frame_jdot_v_local(m, data, fid) for fid in self.contact_ids]
        Jb, Jt = dyn.frame_jacobian_local(m, data, self.base_id)[3:], dyn.frame_jacobian_local(m, data, self.torso_id)[3:]
        dJbv, dJtv = dyn.frame_jdot_v_local(m, data, self.base_id)[3:], dyn.frame_jdot_v_local(m, data, self.torso_id)[3:]
        H = np.zeros((self.n, self.n)); g = np.zeros(self.n)
        Haa = w[0] * np.eye(nv) + w[1] * (J[0].T @ J[0] + J[1].T @ J[1]) + w[2] * data.Ag.T @ data.Ag + w[3] * (Jb.T @ Jb + Jt.T @ Jt)
        H[:nv, :nv] = Haa
        H[nv:nv + fs * nk, nv:nv + fs * nk] = np.eye(fs * nk) * w[4]
        ga = w[0] * (-K[0][0] @ q_diff - K[0][1] @ dq_diff)
        ga = ga + w[1] * (dJv[0] - K[1][0] @ LF_diff - K[1][1] @ dLF_diff) @ J[0]
        ga = ga + w[1] * (dJv[1] - K[1][0] @ RF_diff - K[1][1] @ dRF_diff) @ J[1]
        ga = ga - w[2] * (dH - data.dAg_v) @ data.Ag
        ga = ga + w[3] * (dJbv - K[3][0] @ base_diff - K[3][1] @ dbase_diff) @ Jb
        ga = ga + w[3] * (dJtv - K[3][0] @ torso_diff - K[3][1] @ dtorso_diff) @ Jt
        g[:nv] = ga
        A = np.zeros((self.neq, self.n)); b = np.zeros(self.neq)
        A[:nv, :nv] = M; A[:nv, nv + fs * nk:] = -self.S
        b[:nv] = -data.nle
        C = np.zeros((self.nin, self.n)); l = np.zeros(self.nin)
        for i in range(nk):
            if cs[i]:
                A[:nv, nv + fs * i:nv + fs * (i + 1)] = -J[i].T
                A[nv + fs * i:nv + fs * (i + 1), :nv] = J[i]
                b[:nv] += J[i].T @ forces[fs * i:fs * (i + 1)]
                b[nv + fs * i:nv + fs * (i + 1)] = -dJv[i]
                l[9 * i:9 * (i + 1)] = -self.Cl @ forces[fs * i:fs * (i + 1)]
                C[9 * i:9 * (i + 1), nv + fs * i:nv + fs * (i + 1)] = self.Cmin
        return H, g, A, b, C, l

    def solve(self, data, cs, v, q_diff, dq_diff, LF_diff, dLF_diff, RF_diff, dRF_diff, base_diff, dbase_diff, torso_diff, dtorso_diff, forces, dH, M):
        """-> (a_new, new_forces, torque) as QP_utils.py:736-762 (the acceleration itself is an unknown here)."""
        mats = self.computeMatrice(data, cs, v, q_diff, dq_diff, LF_diff, dLF_diff, RF_diff, dRF_diff, base_diff, dbase_diff, torso_diff, dtorso_diff,
                                   forces, dH, M)
        x, y, z, zb, info = self.qp.solve(*[np.broadcast_to(a, (self.batch,) + a.shape) for a in mats], self.u, self.l_box, self.u_box)
        self.last_info = info
        nv, fs, nk = self.model.nv, self.force_size, self.nk
        return x[0, :nv].copy(), forces + x[0, nv:nv + fs * nk], x[0, nv + fs * nk:].copy()

    # ---- the whole QP on the device (mpc_qp_solve_ikid: csrc/qp_assemble.h) ----
    def enable_device_assembly(self):
        from .aligator._core import LoweringContext
        if self.force_size != 6 or self.nk != 2:
            raise NotImplementedError("device assembly is written for two 6-D contacts")
        ctx = LoweringContext()
        self._frame_idx = np.array([ctx.frame_index(self.model, fid) for fid in self.contact_ids], dtype=np.int32)
        self._base_idx, self._torso_idx = ctx.frame_index(self.model, self.base_id), ctx.frame_index(self.model, self.torso_id)
        self.qp.set_model(*ctx.model_tables())
        K = self.K_gains
        self._gains = np.concatenate([np.asarray(K[0][0], dtype=float).reshape(-1), np.asarray(K[0][1], dtype=float).reshape(-1),
                                      np.asarray(K[1][0], dtype=float).reshape(-1), np.asarray(K[1][1], dtype=float).reshape(-1),
                                      np.asarray(K[3][0], dtype=float).reshape(-1), np.asarray(K[3][1], dtype=float).reshape(-1)])

    def solve_batch_device(self, x, q_diff, dq_diff, LF_diff, dLF_diff, RF_diff, dRF_diff, base_diff, dbase_diff, torso_diff, dtorso_diff, forces, dH, cs,
                           return_matrices=False):
        """Every argument per robot ([B][...]): the state ``x`` (nq + nv) and the task errors the script computes from its references.
        -> (a, new_forces, torque), each [B][...] (, the matrices the library assembled)."""
        if not hasattr(self, "_frame_idx"):
            self.enable_device_assembly()
        B, nv = self.batch, self.model.nv
        bc = lambda a, k: np.broadcast_to(np.asarray(a, dtype=float), (B, k))
        ik = np.concatenate([bc(q_diff, nv), bc(dq_diff, nv), bc(LF_diff, 6), bc(dLF_diff, 6), bc(RF_diff, 6), bc(dRF_diff, 6),
                             bc(base_diff, 3), bc(dbase_diff, 3), bc(torso_diff, 3), bc(dtorso_diff, 3), bc(dH, 6)], axis=1)
        out = self.qp.solve_ikid(self._frame_idx, self._base_idx, self._torso_idx, self.weights, self._gains, self.Cmin, self.l_box, self.u_box,
                                 x, ik, forces, cs, return_matrices=return_matrices, cone_l=self.Cl)
        sol = out[0]
        self.last_info = out[4]
        fs, nk = self.force_size, self.nk
        forces = np.broadcast_to(np.asarray(forces, dtype=float), (B, fs * nk))
        res = (sol[:, :nv].copy(), forces + sol[:, nv:nv + fs * nk], sol[:, nv + fs * nk:].copy())
        return res + (out[5],) if return_matrices else res
