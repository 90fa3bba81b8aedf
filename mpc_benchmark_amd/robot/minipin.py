"""Minimal duck-typed stand-ins for the handful of `pinocchio` objects the reference scripts hand to
`aligator`.  Host-side glue only: the hot path never runs through this file.

The reference builds its problems from a real ``pin.Model`` (talos_utils.py:31-41) which does not exist
in this environment; the shim in ``mpc_benchmark_amd.aligator`` reads models by *attribute name*, so either a
real ``pin.Model`` or the ``Model`` defined here works.  Conventions follow Pinocchio:

* free-flyer configuration ``q = [x y z, qx qy qz qw]``, velocity ``v = [v_lin(body), w(body)]``;
* spatial motion = ``[linear; angular]``, spatial force = ``[force; torque]``;
* ``SE3 * SE3`` composes, ``SE3.act(p)`` maps a point.
"""
from __future__ import annotations

import copy as _copy
import numpy as np

LOCAL = 0
WORLD = 1
LOCAL_WORLD_ALIGNED = 2


class ContactType:
    CONTACT_3D = 1
    CONTACT_6D = 2


def skew(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def exp3(w):
    th = float(np.linalg.norm(w))
    K = skew(w)
    if th < 1e-10:
        return np.eye(3) + K + 0.5 * K @ K
    return np.eye(3) + np.sin(th) / th * K + (1.0 - np.cos(th)) / th ** 2 * (K @ K)


def log3(R):
    c = 0.5 * (np.trace(R) - 1.0)
    c = min(1.0, max(-1.0, c))
    th = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-8:
        return 0.5 * v * (1.0 + th * th / 6.0)
    if th > np.pi - 1e-6:
        # near pi: use the diagonal
        A = 0.5 * (R + np.eye(3))
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[:, k] / ax[k]
        if v @ ax < 0:
            ax = -ax
        return th * ax / np.linalg.norm(ax)
    return th / (2.0 * np.sin(th)) * v


def quat_to_rot(qxyzw):
    x, y, z, w = np.asarray(qxyzw, dtype=float) / np.linalg.norm(qxyzw)
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
    ])


def rot_to_quat(R):
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = np.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = np.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = np.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    q = np.array([x, y, z, w])
    if w < 0:
        q = -q
    return q / np.linalg.norm(q)


class SE3:
    """Rigid placement (rotation, translation) — mirrors the part of ``pin.SE3`` the scripts touch."""

    def __init__(self, rotation=None, translation=None):
        # pin.SE3(R, p) ; pin.SE3(other) ; pin.SE3(H) with a 4 x 4 homogeneous matrix (plot.py:138-144 rebuilds poses that way)
        if translation is None and isinstance(rotation, SE3):
            rotation, translation = rotation.rotation, rotation.translation
        elif translation is None and rotation is not None and np.shape(rotation) == (4, 4):
            H = np.array(rotation, dtype=float)
            rotation, translation = H[:3, :3], H[:3, 3]
        self.rotation = np.eye(3) if rotation is None else np.array(rotation, dtype=float)
        self.translation = np.zeros(3) if translation is None else np.array(translation, dtype=float)

    @staticmethod
    def Identity():
        return SE3()

    def copy(self):
        c = SE3.__new__(SE3)
        c.rotation, c.translation = self.rotation.copy(), self.translation.copy()
        return c

    def __mul__(self, other):
        return SE3(self.rotation @ other.rotation, self.rotation @ other.translation + self.translation)

    def inverse(self):
        return SE3(self.rotation.T, -self.rotation.T @ self.translation)

    def act(self, p):
        return self.rotation @ np.asarray(p, dtype=float) + self.translation

    def actInv(self, other):
        return self.inverse() * other

    @property
    def homogeneous(self):
        H = np.eye(4)
        H[:3, :3] = self.rotation
        H[:3, 3] = self.translation
        return H

    def action(self):
        """6x6 motion action matrix Ad(M) on [lin; ang] vectors."""
        A = np.zeros((6, 6))
        A[:3, :3] = self.rotation
        A[3:, 3:] = self.rotation
        A[:3, 3:] = skew(self.translation) @ self.rotation
        return A

    def __repr__(self):
        return "SE3(R=\n%s,\n p=%s)" % (self.rotation, self.translation)


def exp6(nu):
    v = np.asarray(nu[:3], dtype=float)
    w = np.asarray(nu[3:], dtype=float)
    th = float(np.linalg.norm(w))
    K = skew(w)
    R = exp3(w)
    if th < 1e-10:
        V = np.eye(3) + 0.5 * K + K @ K / 6.0
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    return SE3(R, V @ v)


def log6(M):
    w = log3(M.rotation)
    th = float(np.linalg.norm(w))
    K = skew(w)
    if th < 1e-8:
        Vinv = np.eye(3) - 0.5 * K + K @ K / 12.0
    else:
        Vinv = np.eye(3) - 0.5 * K + (1.0 / th ** 2 - (1 + np.cos(th)) / (2 * th * np.sin(th))) * (K @ K)
    return np.concatenate((Vinv @ M.translation, w))


class Motion:
    def __init__(self, linear=None, angular=None):
        self.np = np.zeros(6)
        if linear is not None:
            self.np[:3] = linear
        if angular is not None:
            self.np[3:] = angular

    @staticmethod
    def Zero():
        return Motion()

    @property
    def linear(self):
        return self.np[:3]

    @property
    def angular(self):
        return self.np[3:]

    @property
    def vector(self):
        return self.np


class Force(Motion):
    pass


class Inertia:
    """mass, lever (CoM in the joint frame) and rotational inertia about the CoM."""

    def __init__(self, mass, lever, inertia):
        self.mass = float(mass)
        self.lever = np.array(lever, dtype=float)
        self.inertia = np.array(inertia, dtype=float)

    def copy(self):
        return Inertia(self.mass, self.lever.copy(), self.inertia.copy())

    def se3Action(self, M):
        """Inertia expressed in the frame a, given this inertia in frame b and aMb = M."""
        R = M.rotation
        return Inertia(self.mass, M.act(self.lever), R @ self.inertia @ R.T)

    def __add__(self, o):
        m = self.mass + o.mass
        if m <= 0:
            return Inertia(0.0, np.zeros(3), np.zeros((3, 3)))
        c = (self.mass * self.lever + o.mass * o.lever) / m
        I = np.zeros((3, 3))
        for b in (self, o):
            d = b.lever - c
            I += b.inertia + b.mass * (d @ d * np.eye(3) - np.outer(d, d))
        return Inertia(m, c, I)

    def matrix(self):
        """6x6 spatial inertia at the frame origin, [lin; ang] ordering."""
        S = skew(self.lever)
        Y = np.zeros((6, 6))
        Y[:3, :3] = self.mass * np.eye(3)
        Y[:3, 3:] = -self.mass * S
        Y[3:, :3] = self.mass * S
        Y[3:, 3:] = self.inertia - self.mass * S @ S
        return Y


_JOINT_NQ = {"JointModelFreeFlyer": 7, "JointModelRX": 1, "JointModelRY": 1, "JointModelRZ": 1}
_JOINT_NV = {"JointModelFreeFlyer": 6, "JointModelRX": 1, "JointModelRY": 1, "JointModelRZ": 1}


class JointModel:
    def __init__(self, kind, idx_q, idx_v):
        self._kind = kind
        self.idx_q = idx_q
        self.idx_v = idx_v
        self.nq = _JOINT_NQ.get(kind, 0)
        self.nv = _JOINT_NV.get(kind, 0)

    def shortname(self):
        return self._kind


class FrameType:
    OP_FRAME, JOINT, FIXED_JOINT, BODY, SENSOR = 1, 2, 4, 8, 16


OP_FRAME = FrameType.OP_FRAME


class Frame:
    """``Frame(name, parentJoint, placement)`` (this module's own order) or Pinocchio's
    ``Frame(name, parentJoint, parentFrame, placement, type)`` (talos_utils.py:48-96)."""

    def __init__(self, name, parentJoint, placement, parentFrame=0, type=FrameType.OP_FRAME):
        if isinstance(placement, (int, np.integer)) and isinstance(parentFrame, SE3):  # Pinocchio's argument order
            placement, parentFrame = parentFrame, int(placement)
        self.name = name
        self.parentJoint = int(parentJoint)
        self.parent = int(parentJoint)
        self.parentFrame = parentFrame
        self.placement = placement
        self.type = type


class _Names(list):
    """``model.names``: Pinocchio's StdVec_StdString has ``.tolist()`` and keeps it under slicing
    (``rmodel.names[1:].tolist()``, fulldynamic_talos.py:46)."""

    def tolist(self):
        return list(self)

    def __getitem__(self, i):
        r = list.__getitem__(self, i)
        return _Names(r) if isinstance(i, slice) else r


class Data:
    def __init__(self, model):
        self.oMi = [SE3() for _ in range(model.njoints)]
        self.oMf = [SE3() for _ in range(len(model.frames))]
        self.com = [np.zeros(3)]
        self.hg = Force()


class Model:
    """Kinematic tree with the attribute names of ``pin.Model`` that the shim reads."""

    def __init__(self):
        self.names = _Names(["universe"])
        self.parents = [0]
        self.jointPlacements = [SE3()]
        self.inertias = [Inertia(0.0, np.zeros(3), np.zeros((3, 3)))]
        self.joints = [JointModel("JointModelUniverse", -1, -1)]
        self.frames = [Frame("universe", 0, SE3())]
        self.nq = 0
        self.nv = 0
        self.effortLimit = np.zeros(0)
        self.velocityLimit = np.zeros(0)
        self.upperPositionLimit = np.zeros(0)
        self.lowerPositionLimit = np.zeros(0)
        self.referenceConfigurations = {}
        self.gravity = Motion(linear=[0, 0, -9.81])

    @property
    def njoints(self):
        return len(self.parents)

    @property
    def nframes(self):
        return len(self.frames)

    def addJoint(self, parent, kind, placement, name, effort=0.0, lower=None, upper=None):
        nq, nv = _JOINT_NQ[kind], _JOINT_NV[kind]
        self.names.append(name)
        self.parents.append(int(parent))
        self.jointPlacements.append(placement.copy())
        self.inertias.append(Inertia(0.0, np.zeros(3), np.zeros((3, 3))))
        self.joints.append(JointModel(kind, self.nq, self.nv))
        self.nq += nq
        self.nv += nv
        big = 1e30
        lo = np.full(nq, -big) if lower is None else np.atleast_1d(np.asarray(lower, dtype=float))
        hi = np.full(nq, big) if upper is None else np.atleast_1d(np.asarray(upper, dtype=float))
        self.effortLimit = np.concatenate((self.effortLimit, np.full(nv, float(effort))))
        self.velocityLimit = np.concatenate((self.velocityLimit, np.full(nv, big)))
        self.lowerPositionLimit = np.concatenate((self.lowerPositionLimit, lo))
        self.upperPositionLimit = np.concatenate((self.upperPositionLimit, hi))
        jid = self.njoints - 1
        self.frames.append(Frame(name, jid, SE3()))
        return jid

    def appendBodyToJoint(self, jid, inertia, placement=None):
        Y = inertia if placement is None else inertia.se3Action(placement)
        self.inertias[jid] = self.inertias[jid] + Y if self.inertias[jid].mass > 0 else Y.copy()

    def addFrame(self, frame):
        self.frames.append(frame)
        return len(self.frames) - 1

    def getFrameId(self, name):
        for i, f in enumerate(self.frames):
            if f.name == name:
                return i
        return len(self.frames)

    def getJointId(self, name):
        for i, n in enumerate(self.names):
            if n == name:
                return i
        return len(self.names)

    def existFrame(self, name):
        return self.getFrameId(name) < len(self.frames)

    def copy(self):
        return _copy.deepcopy(self)

    def createData(self):
        return Data(self)


def neutral(model):
    q = np.zeros(model.nq)
    for j in model.joints[1:]:
        if j.shortname() == "JointModelFreeFlyer":
            q[j.idx_q + 6] = 1.0
    return q


def _joint_transform(jm, q):
    k = jm.shortname()
    if k == "JointModelFreeFlyer":
        return SE3(quat_to_rot(q[jm.idx_q + 3: jm.idx_q + 7]), q[jm.idx_q: jm.idx_q + 3])
    ax = {"JointModelRX": 0, "JointModelRY": 1, "JointModelRZ": 2}[k]
    w = np.zeros(3)
    w[ax] = q[jm.idx_q]
    return SE3(exp3(w), np.zeros(3))


_AXIS = {"JointModelRX": 0, "JointModelRY": 1, "JointModelRZ": 2}


def _se3_raw(R, p):
    M = SE3.__new__(SE3)
    M.rotation, M.translation = R, p
    return M


def forwardKinematics(model, data, q, v=None):
    # (plain 3 x 3 arrays in the loop: the MPC loops call this every tick for the measured foot poses)
    q = np.asarray(q, dtype=float)
    data._q = q  # the algorithms that Pinocchio runs on "the data of the last forward kinematics" (computeJointJacobians) read it
    data._v = None if v is None else np.asarray(v, dtype=float)
    cq, sq = np.cos(q), np.sin(q)
    oMi = data.oMi
    for i in range(1, model.njoints):
        jm, pl = model.joints[i], model.jointPlacements[i]
        k = jm.shortname()
        if k == "JointModelFreeFlyer":
            R = pl.rotation @ quat_to_rot(q[jm.idx_q + 3: jm.idx_q + 7])
            t = pl.rotation @ q[jm.idx_q: jm.idx_q + 3] + pl.translation
        else:
            a = _AXIS[k]
            b, d = (a + 1) % 3, (a + 2) % 3
            c_, s_ = cq[jm.idx_q], sq[jm.idx_q]
            P = pl.rotation
            R = np.empty((3, 3))
            R[:, a] = P[:, a]                       # P @ Rot(axis a, angle): the axis column stays,
            R[:, b] = c_ * P[:, b] + s_ * P[:, d]   # the other two rotate into each other
            R[:, d] = c_ * P[:, d] - s_ * P[:, b]
            t = pl.translation
        p = model.parents[i]
        if p != 0:
            Mp = oMi[p]
            t = Mp.rotation @ t + Mp.translation
            R = Mp.rotation @ R
        elif t is pl.translation:
            t = t.copy()
        oMi[i] = _se3_raw(R, t)
    if isinstance(data.oMf, _LazyFrames):
        data.oMf.invalidate()


class _LazyFrames:
    """``data.oMf``: frame placements computed from ``data.oMi`` when they are read (the scripts read two feet out of ~70 frames per
    tick); ``updateFramePlacements`` marks them stale, which is all Pinocchio's eager update amounts to for a reader."""

    def __init__(self, model, data):
        self._model, self._data = model, data
        self._cache = [None] * len(model.frames)

    def invalidate(self):
        self._cache = [None] * len(self._cache)

    def __len__(self):
        return len(self._cache)

    def __getitem__(self, i):
        M = self._cache[i]
        if M is None:
            f = self._model.frames[i]
            M = self._data.oMi[f.parentJoint] * f.placement if f.parentJoint > 0 else f.placement.copy()
            self._cache[i] = M
        return M

    def __setitem__(self, i, M):
        self._cache[i] = M

    def __iter__(self):
        return (self[i] for i in range(len(self._cache)))


def updateFramePlacements(model, data):
    if not isinstance(data.oMf, _LazyFrames):
        data.oMf = _LazyFrames(model, data)
    data.oMf.invalidate()


def framesForwardKinematics(model, data, q):
    forwardKinematics(model, data, q)
    updateFramePlacements(model, data)


def computeTotalMass(model):
    return float(sum(Y.mass for Y in model.inertias))


def centerOfMass(model, data, q, v=None):
    forwardKinematics(model, data, q)
    m = 0.0
    c = np.zeros(3)
    for i in range(1, model.njoints):
        Y = model.inertias[i]
        c += Y.mass * data.oMi[i].act(Y.lever)
        m += Y.mass
    data.com[0] = c / m
    return data.com[0].copy()


def integrate(model, q, dv):
    """q (+) dv with Pinocchio's free-flyer convention (body-frame twist, right multiplication)."""
    out = np.array(q, dtype=float)
    for j in model.joints[1:]:
        if j.shortname() == "JointModelFreeFlyer":
            M = SE3(quat_to_rot(q[j.idx_q + 3: j.idx_q + 7]), q[j.idx_q: j.idx_q + 3]) * exp6(dv[j.idx_v: j.idx_v + 6])
            out[j.idx_q: j.idx_q + 3] = M.translation
            out[j.idx_q + 3: j.idx_q + 7] = rot_to_quat(M.rotation)
        else:
            out[j.idx_q] = q[j.idx_q] + dv[j.idx_v]
    return out


def difference(model, q0, q1):
    """Tangent vector d such that q0 (+) d = q1."""
    d = np.zeros(model.nv)
    for j in model.joints[1:]:
        if j.shortname() == "JointModelFreeFlyer":
            M0 = SE3(quat_to_rot(q0[j.idx_q + 3: j.idx_q + 7]), q0[j.idx_q: j.idx_q + 3])
            M1 = SE3(quat_to_rot(q1[j.idx_q + 3: j.idx_q + 7]), q1[j.idx_q: j.idx_q + 3])
            d[j.idx_v: j.idx_v + 6] = log6(M0.inverse() * M1)
        else:
            d[j.idx_v] = q1[j.idx_q] - q0[j.idx_q]
    return d


class _Corrector:
    def __init__(self, n):
        self.Kp = np.zeros(n)
        self.Kd = np.zeros(n)


class RigidConstraintData:
    def __init__(self):
        self.contact_force = Force()


class RigidConstraintModel:
    """Mirror of the constructor used at fulldynamic_talos.py:84-92."""

    def __init__(self, type, model, joint1_id, joint1_placement, joint2_id=0, joint2_placement=None,
                 reference_frame=LOCAL):
        self.type = type
        self.joint1_id = int(joint1_id)
        self.joint1_placement = joint1_placement.copy()
        self.joint2_id = int(joint2_id)
        self.joint2_placement = SE3() if joint2_placement is None else joint2_placement.copy()
        self.reference_frame = reference_frame
        self.corrector = _Corrector(6 if type == ContactType.CONTACT_6D else 3)
        self.name = ""

    def size(self):
        return 6 if self.type == ContactType.CONTACT_6D else 3

    def createData(self):
        return RigidConstraintData()


class ProximalSettings:
    def __init__(self, accuracy=1e-12, mu=0.0, max_iter=1):
        self.absolute_accuracy = accuracy
        self.accuracy = accuracy
        self.mu = mu
        self.max_iter = max_iter


def buildReducedModel(model, locked_joint_ids, q_ref):
    """Lock the given joints at ``q_ref`` and fold their bodies into the parents
    (what ``robot.buildReducedRobot`` does at talos_utils.py:37)."""
    locked = set(int(j) for j in locked_joint_ids)
    red = Model()
    data = model.createData()
    # placement of every original joint w.r.t. its closest kept ancestor (or universe)
    kept_id = {0: 0}
    rel = {0: SE3()}
    qmap = []
    for i in range(1, model.njoints):
        p = model.parents[i]
        liMi = model.jointPlacements[i]
        if i in locked:
            M = rel[p] * liMi * _joint_transform(model.joints[i], q_ref)
            kept_id[i] = kept_id[p]
            rel[i] = M
            red.appendBodyToJoint(kept_id[i], model.inertias[i], M) if kept_id[i] > 0 else None
        else:
            jm = model.joints[i]
            sl = slice(jm.idx_q, jm.idx_q + jm.nq)
            slv = slice(jm.idx_v, jm.idx_v + jm.nv)
            jid = red.addJoint(kept_id[p], jm.shortname(), rel[p] * liMi, model.names[i],
                               effort=0.0, lower=model.lowerPositionLimit[sl], upper=model.upperPositionLimit[sl])
            red.effortLimit[red.joints[jid].idx_v: red.joints[jid].idx_v + jm.nv] = model.effortLimit[slv]
            red.appendBodyToJoint(jid, model.inertias[i])
            kept_id[i] = jid
            rel[i] = SE3()
            qmap.append((sl, slice(red.joints[jid].idx_q, red.joints[jid].idx_q + jm.nq)))
    red.frames = [Frame("universe", 0, SE3())]
    for f in model.frames[1:]:
        pj = f.parentJoint
        red.frames.append(Frame(f.name, kept_id[pj], rel[pj] * f.placement))
    for name, q in model.referenceConfigurations.items():
        qr = np.zeros(red.nq)
        for src, dst in qmap:
            qr[dst] = q[src]
        red.referenceConfigurations[name] = qr
    del data
    return red


def frame_placements_batch(model, Q, frame_ids):
    """World placements of a few frames for a batch of configurations ``Q`` [B, nq]: (R [B, 3, 3], p [B, 3]) per frame, walking only
    the joints on the way from the root to the frame (ensembles read the two sole frames of every robot each tick)."""
    Q = np.asarray(Q, dtype=float)
    B = Q.shape[0]
    out = []
    for fid in frame_ids:
        f = model.frames[fid]
        chain = []
        j = f.parentJoint
        while j > 0:
            chain.append(j)
            j = model.parents[j]
        R = np.broadcast_to(np.eye(3), (B, 3, 3))
        p = np.zeros((B, 3))
        for i in reversed(chain):
            jm, pl = model.joints[i], model.jointPlacements[i]
            k = jm.shortname()
            if k == "JointModelFreeFlyer":
                qn = Q[:, jm.idx_q + 3: jm.idx_q + 7]
                x, y, z, w = (qn / np.linalg.norm(qn, axis=1, keepdims=True)).T
                Rj = np.empty((B, 3, 3))
                Rj[:, 0, 0], Rj[:, 0, 1], Rj[:, 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)
                Rj[:, 1, 0], Rj[:, 1, 1], Rj[:, 1, 2] = 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)
                Rj[:, 2, 0], Rj[:, 2, 1], Rj[:, 2, 2] = 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)
                Rl = pl.rotation @ Rj
                pl_t = Q[:, jm.idx_q: jm.idx_q + 3] @ pl.rotation.T + pl.translation
            else:
                a = _AXIS[k]
                b_, d_ = (a + 1) % 3, (a + 2) % 3
                c_, s_ = np.cos(Q[:, jm.idx_q]), np.sin(Q[:, jm.idx_q])
                P = pl.rotation
                Rl = np.empty((B, 3, 3))
                Rl[:, :, a] = P[:, a]
                Rl[:, :, b_] = c_[:, None] * P[:, b_] + s_[:, None] * P[:, d_]
                Rl[:, :, d_] = c_[:, None] * P[:, d_] - s_[:, None] * P[:, b_]
                pl_t = np.broadcast_to(pl.translation, (B, 3))
            p = np.einsum("bij,bj->bi", R, pl_t) + p
            R = R @ Rl
        out.append((R @ f.placement.rotation, np.einsum("bij,j->bi", R, f.placement.translation) + p))
    return out


# ---- rigid-body algorithms the scripts call around their low-level QPs (kinodynamic_talos.py:424-430, centroidal_talos.py:421-432, -------
# fulldynamic_talos.py:491-492; QP_utils.py:520-526, 667-674; talos_utils.py:376-400).  Host-side glue, numpy (robot/dynamics.py): the hot
# path computes the same quantities inside the stage kernel.  One evaluation per (q, v) fills everything; the individual calls are views.
def _terms(model, data, q=None, v=None):
    from . import dynamics
    q = getattr(data, "_q", None) if q is None else np.asarray(q, dtype=float)
    if q is None:
        raise RuntimeError("call forwardKinematics(model, data, q) first")
    if v is None:
        v = getattr(data, "_v", None)
        if v is None:
            v = getattr(data, "_v_terms", None)
    v = np.zeros(model.nv) if v is None else np.asarray(v, dtype=float)
    key = (q.tobytes(), v.tobytes())
    if getattr(data, "_terms_key", None) != key:
        dynamics.compute_all_terms(model, data, q, v)
        data._terms_key = key
        data._v_terms = v
    return data


def computeJointJacobians(model, data, q=None):
    _terms(model, data, q)
    return data.S


def computeJointJacobiansTimeVariation(model, data, q, v):
    _terms(model, data, q, v)


def crba(model, data, q):
    """Joint-space inertia matrix, symmetric (the Python binding of Pinocchio fills both triangles)."""
    return _terms(model, data, q).M


def nonLinearEffects(model, data, q, v):
    return _terms(model, data, q, v).nle


def computeCentroidalMomentum(model, data, q=None, v=None):
    """``data.hg`` = Ag(q) v about the centre of mass, world axes (fulldynamic_talos.py:491-492, centroidal_talos.py:415-417)."""
    _terms(model, data, q, v)
    h = data.Ag @ data._v_terms
    data.hg = Force(h[:3], h[3:])
    return data.hg


def ccrba(model, data, q, v):
    computeCentroidalMomentum(model, data, q, v)
    return data.Ag


def dccrba(model, data, q, v):
    """``data.Ag`` and its time derivative ``data.dAg`` (QP_utils.py:696 multiplies ``data.dAg @ v``)."""
    from . import dynamics
    _terms(model, data, q, v)
    data.dAg = dynamics.centroidal_matrix_time_variation(model, data, data._v_terms)
    computeCentroidalMomentum(model, data, q, v)
    return data.dAg


def getFrameJacobian(model, data, frame_id, reference_frame=LOCAL):
    from . import dynamics
    _terms(model, data)
    J = dynamics.frame_jacobian_local(model, data, frame_id)
    if reference_frame == LOCAL:
        return J
    R = data.oMf[frame_id].rotation
    if reference_frame == LOCAL_WORLD_ALIGNED:
        return np.vstack((R @ J[:3], R @ J[3:]))
    return data.oMf[frame_id].action() @ J


def getFrameJacobianTimeVariation(model, data, frame_id, reference_frame=LOCAL):
    from . import dynamics
    if reference_frame != LOCAL:
        raise NotImplementedError("getFrameJacobianTimeVariation: LOCAL only (all the scripts use)")
    _terms(model, data)
    return dynamics.frame_jacobian_time_variation_local(model, data, frame_id)


def getFrameVelocity(model, data, frame_id, reference_frame=LOCAL):
    from . import dynamics
    _terms(model, data)
    vl = dynamics.frame_velocity_local(model, data, frame_id)
    if reference_frame == LOCAL:
        return vl
    R = data.oMf[frame_id].rotation
    if reference_frame == LOCAL_WORLD_ALIGNED:
        return Motion(R @ vl.linear, R @ vl.angular)
    w = data.oMf[frame_id].action() @ vl.np
    return Motion(w[:3], w[3:])


def _quat_to_rot_batch(Q):
    qn = Q / np.linalg.norm(Q, axis=1, keepdims=True)
    x, y, z, w = qn.T
    R = np.empty((Q.shape[0], 3, 3))
    R[:, 0, 0], R[:, 0, 1], R[:, 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)
    R[:, 1, 0], R[:, 1, 1], R[:, 1, 2] = 2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)
    R[:, 2, 0], R[:, 2, 1], R[:, 2, 2] = 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)
    return R


def difference_batch(model, Q0, Q1):
    """``difference(model, q0, q1)`` for B configurations at once (free-flyer root followed by revolute joints): [B, nv]."""
    Q0, Q1 = np.asarray(Q0, dtype=float), np.asarray(Q1, dtype=float)
    B = Q0.shape[0]
    d = np.zeros((B, model.nv))
    for j in model.joints[1:]:
        if j.shortname() == "JointModelFreeFlyer":
            R0, R1 = _quat_to_rot_batch(Q0[:, j.idx_q + 3:j.idx_q + 7]), _quat_to_rot_batch(Q1[:, j.idx_q + 3:j.idx_q + 7])
            R = np.einsum("bji,bjk->bik", R0, R1)
            p = np.einsum("bji,bj->bi", R0, Q1[:, j.idx_q:j.idx_q + 3] - Q0[:, j.idx_q:j.idx_q + 3])
            c = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) / 2.0, -1.0, 1.0)
            th = np.arccos(c)
            wv = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
            small = th < 1e-8
            ths = np.where(small, 1.0, th)
            w = np.where(small[:, None], 0.5 * wv, (ths / (2.0 * np.sin(ths)))[:, None] * wv)
            K = np.zeros((B, 3, 3))
            K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -w[:, 2], w[:, 1], w[:, 2], -w[:, 0], -w[:, 1], w[:, 0]
            coef = np.where(small, 1.0 / 12.0, 1.0 / ths ** 2 - (1 + np.cos(ths)) / (2 * ths * np.sin(ths)))
            Vinv = np.eye(3) - 0.5 * K + coef[:, None, None] * (K @ K)
            d[:, j.idx_v:j.idx_v + 3] = np.einsum("bij,bj->bi", Vinv, p)
            d[:, j.idx_v + 3:j.idx_v + 6] = w
        else:
            d[:, j.idx_v] = Q1[:, j.idx_q] - Q0[:, j.idx_q]
    return d
