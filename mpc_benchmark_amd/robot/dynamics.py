"""Rigid-body quantities the whole-body QPs of the reference read from Pinocchio (QP_utils.py:517-535: ``data.nle``, the
joint-space inertia ``M``, ``getFrameJacobian`` / ``getFrameJacobianTimeVariation`` / ``getFrameVelocity`` in the LOCAL
frame), on the duck-typed model of ``minipin`` in plain numpy.  Host-side glue for the N3 row: the hot path computes the same
quantities inside the stage kernel; this file exists so that the QP classes can be driven and tested without Pinocchio.

World-frame formulation: the motion subspace column of dof k is ``S_k = Ad(oMi) s_k`` at the world origin, velocities and
accelerations accumulate down the tree, forces back up (recursive Newton-Euler); ``[lin; ang]`` ordering throughout.
"""
from __future__ import annotations

import numpy as np

from . import minipin as pin


def _crm(v):
    """motion cross product matrix: (v x) m"""
    X = np.zeros((6, 6))
    w, l = pin.skew(v[3:]), pin.skew(v[:3])
    X[:3, :3] = w; X[:3, 3:] = l; X[3:, 3:] = w
    return X


def _crf(v):
    """force cross product matrix: v x* f = -(v x)^T f"""
    return -_crm(v).T


def _local_columns(jm):
    k = jm.shortname()
    if k == "JointModelFreeFlyer":
        return np.eye(6)
    c = np.zeros((6, 1))
    c[3 + {"JointModelRX": 0, "JointModelRY": 1, "JointModelRZ": 2}[k], 0] = 1.0
    return c


def compute_all_terms(model, data, q, v):
    """Fills ``data``: oMi / oMf, ``S`` (6 x nv world columns), ``v_w`` / ``a0_w`` (spatial velocity and the velocity-product
    acceleration of every joint, world frame), ``M`` (nv x nv), ``nle`` (Coriolis + gravity), ``Yw`` (world spatial inertias),
    ``Ag`` / ``dAg_v`` (centroidal momentum matrix and (dAg/dt) v)."""
    q = np.asarray(q, dtype=float); v = np.asarray(v, dtype=float)
    pin.framesForwardKinematics(model, data, q)
    nj, nv = model.njoints, model.nv
    S = np.zeros((6, nv))
    Yw = [np.zeros((6, 6)) for _ in range(nj)]
    vw = [np.zeros(6) for _ in range(nj)]
    a0 = [np.zeros(6) for _ in range(nj)]
    for i in range(1, nj):
        jm = model.joints[i]
        Ad = data.oMi[i].action()
        cols = Ad @ _local_columns(jm)
        S[:, jm.idx_v:jm.idx_v + jm.nv] = cols
        Ainv = np.linalg.inv(Ad)
        Yw[i] = Ainv.T @ model.inertias[i].matrix() @ Ainv
        p = model.parents[i]
        vj = cols @ v[jm.idx_v:jm.idx_v + jm.nv]
        vw[i] = vw[p] + vj
        a0[i] = a0[p] + _crm(vw[i]) @ vj
    data.S, data.Yw, data.v_w, data.a0_w = S, Yw, vw, a0

    def rnea(acc, with_velocity, gravity):
        a = [np.zeros(6) for _ in range(nj)]
        f = [np.zeros(6) for _ in range(nj)]
        base = np.zeros(6)
        if gravity:
            base[:3] = -np.asarray(model.gravity.linear)
        for i in range(1, nj):
            jm = model.joints[i]
            p = model.parents[i]
            a[i] = (a[p] if p else base) + S[:, jm.idx_v:jm.idx_v + jm.nv] @ acc[jm.idx_v:jm.idx_v + jm.nv]
            if with_velocity:
                a[i] = a[i] + (a0[i] - a0[p])
            f[i] = Yw[i] @ a[i]
            if with_velocity:
                f[i] = f[i] + _crf(vw[i]) @ (Yw[i] @ vw[i])
        tau = np.zeros(nv)
        for i in range(nj - 1, 0, -1):
            jm = model.joints[i]
            tau[jm.idx_v:jm.idx_v + jm.nv] = S[:, jm.idx_v:jm.idx_v + jm.nv].T @ f[i]
            p = model.parents[i]
            if p:
                f[p] = f[p] + f[i]
        return tau

    data.nle = rnea(np.zeros(nv), True, True)
    # centroidal momentum matrix Ag (6 x nv, [lin; ang] about the CoM, world axes) and its drift (dAg/dt) v: momentum of the
    # subtree below each dof at the world origin, moved to the CoM; drift = net force of the zero-acceleration motion
    mtot = sum(Y.mass for Y in model.inertias[1:])
    com = sum(Y.mass * data.oMi[i].act(Y.lever) for i, Y in enumerate(model.inertias) if i) / mtot
    Yc = [Y.copy() for Y in Yw]
    for i in range(nj - 1, 0, -1):
        if model.parents[i]:
            Yc[model.parents[i]] = Yc[model.parents[i]] + Yc[i]
    Ao = np.zeros((6, nv)); fo = np.zeros(6)
    for i in range(1, nj):
        jm = model.joints[i]
        Ao[:, jm.idx_v:jm.idx_v + jm.nv] = Yc[i] @ S[:, jm.idx_v:jm.idx_v + jm.nv]
        fo += Yw[i] @ a0[i] + _crf(vw[i]) @ (Yw[i] @ vw[i])
    cx = pin.skew(com)
    data.Ag = np.vstack((Ao[:3], Ao[3:] - cx @ Ao[:3]))
    data.dAg_v = np.concatenate((fo[:3], fo[3:] - cx @ fo[:3]))
    data.com[0] = com
    M = np.zeros((nv, nv))
    for k in range(nv):
        e = np.zeros(nv); e[k] = 1.0
        M[:, k] = rnea(e, False, False)
    data.M = 0.5 * (M + M.T)
    return data


def _supports(model, joint, dof_joint):
    """True if ``dof_joint`` is ``joint`` or one of its ancestors."""
    i = joint
    while i:
        if i == dof_joint:
            return True
        i = model.parents[i]
    return False


def frame_jacobian_local(model, data, frame_id):
    """6 x nv Jacobian of the frame velocity expressed in the frame (pin.getFrameJacobian(..., pin.LOCAL))."""
    f = model.frames[frame_id]
    Ainv = np.linalg.inv(data.oMf[frame_id].action())
    J = np.zeros((6, model.nv))
    for i in range(1, model.njoints):
        if _supports(model, f.parentJoint, i):
            jm = model.joints[i]
            J[:, jm.idx_v:jm.idx_v + jm.nv] = Ainv @ data.S[:, jm.idx_v:jm.idx_v + jm.nv]
    return J


def frame_velocity_local(model, data, frame_id):
    """pin.getFrameVelocity(model, data, id) (LOCAL): Motion with .linear / .angular"""
    f = model.frames[frame_id]
    vl = np.linalg.inv(data.oMf[frame_id].action()) @ data.v_w[f.parentJoint]
    return pin.Motion(vl[:3], vl[3:])


def frame_jdot_v_local(model, data, frame_id):
    """(dJ/dt) v of the LOCAL frame Jacobian = spatial acceleration of the frame at zero joint acceleration, expressed in the
    frame (what QP_utils.py multiplies out as ``getFrameJacobianTimeVariation(...) @ v``)."""
    f = model.frames[frame_id]
    return np.linalg.inv(data.oMf[frame_id].action()) @ data.a0_w[f.parentJoint]


def frame_jacobian_time_variation_local(model, data, frame_id):
    """6 x nv time derivative of the LOCAL frame Jacobian (pin.getFrameJacobianTimeVariation(..., pin.LOCAL)): with J = X_f^-1 S,
    X_f' = (v_f x) X_f and S_k' = (v_i x) S_k, column k of a supporting joint i is X_f^-1 ((v_i - v_f) x) S_k.  Its product with v is
    ``frame_jdot_v_local`` (the contributions of v_f cancel: v_f x v_f = 0)."""
    f = model.frames[frame_id]
    Ainv = np.linalg.inv(data.oMf[frame_id].action())
    vf = data.v_w[f.parentJoint]
    dJ = np.zeros((6, model.nv))
    for i in range(1, model.njoints):
        if _supports(model, f.parentJoint, i):
            jm = model.joints[i]
            dJ[:, jm.idx_v:jm.idx_v + jm.nv] = Ainv @ _crm(data.v_w[i] - vf) @ data.S[:, jm.idx_v:jm.idx_v + jm.nv]
    return dJ


def centroidal_matrix_time_variation(model, data, v):
    """dAg/dt (6 x nv; pin.dccrba's ``data.dAg``): Ag = X_c^* [Yc_i S_k] with the composite inertias Yc (world frame, world origin);
    Yw_j' = (v_j x*) Yw_j - Yw_j (v_j x), S_k' = (v_i x) S_k, and the shift to the centre of mass c moves with c' = h_lin / m.
    ``dAg @ v`` equals ``data.dAg_v`` (net force of the zero-acceleration motion)."""
    nj, nv = model.njoints, model.nv
    S, Yw, vw = data.S, data.Yw, data.v_w
    dY = [np.zeros((6, 6)) for _ in range(nj)]
    Yc = [Y.copy() for Y in Yw]
    for i in range(1, nj):
        dY[i] = _crf(vw[i]) @ Yw[i] - Yw[i] @ _crm(vw[i])
    for i in range(nj - 1, 0, -1):
        p = model.parents[i]
        if p:
            Yc[p] = Yc[p] + Yc[i]
            dY[p] = dY[p] + dY[i]
    Ao = np.zeros((6, nv)); dAo = np.zeros((6, nv))
    for i in range(1, nj):
        jm = model.joints[i]
        sl = slice(jm.idx_v, jm.idx_v + jm.nv)
        Ao[:, sl] = Yc[i] @ S[:, sl]
        dAo[:, sl] = dY[i] @ S[:, sl] + Yc[i] @ (_crm(vw[i]) @ S[:, sl])
    mtot = sum(Y.mass for Y in model.inertias[1:])
    c = data.com[0]
    cdot = Ao[:3] @ np.asarray(v, dtype=float) / mtot
    return np.vstack((dAo[:3], dAo[3:] - pin.skew(cdot) @ Ao[:3] - pin.skew(c) @ dAo[:3]))
