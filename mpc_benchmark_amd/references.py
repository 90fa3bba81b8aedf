"""Reference generators that feed the hot path's per-tick parameters ("next" row N1 of SURVEY.md §8f).

What the Talos scripts do around ``solver.run`` every MPC tick (fulldynamic_talos.py:444-463, centroidal_talos.py:357-384):
advance the take-off / landing countdowns of both feet (talos_utils.py:350-373), regenerate the swing-foot placement
references over the horizon (talos_utils.py:187-327: a degree-8 Bezier with four repeated control points at either end —
zero velocity, acceleration and jerk at take-off and landing — whose middle point is lifted by the swing apex, plus an
interpolated rotation) and push them into the stages with ``setReference``.

The reference builds the curve with ndcurves (``bezier3`` + ``SE3Curve``); that package is not available here and is not
needed: the curve is evaluated in closed form with numpy.  Poses are duck-typed (``.translation`` (3), ``.rotation``
(3x3), ``.copy()``) so that ``pin.SE3`` and ``mpc_benchmark_amd.robot.minipin.SE3`` both work.
"""
from __future__ import annotations

from math import comb

import numpy as np

# ---- contact schedule ----------------------------------------------------------------------------------------------

DOUBLE, LEFT_ONLY, RIGHT_ONLY = (True, True), (True, False), (False, True)


def walking_contact_phases(T_ds, T_ss, total_steps, horizon):
    """[left, right] contact flags per tick of the walk of fulldynamic_talos.py:254-266: double support, `total_steps`
    pairs of (left stance, double, right stance, double), one more left stance + double, then 2 horizons of standing."""
    ph = [list(DOUBLE)] * T_ds
    for _ in range(total_steps):
        ph += [list(LEFT_ONLY)] * T_ss + [list(DOUBLE)] * T_ds + [list(RIGHT_ONLY)] * T_ss + [list(DOUBLE)] * T_ds
    ph += [list(LEFT_ONLY)] * T_ss + [list(DOUBLE)] * T_ds
    ph += [list(DOUBLE)] * horizon * 2
    return ph


def contact_event_times(contact_phases, horizon):
    """Tick indices (shifted by the horizon, as the scripts do: the event enters the horizon's far end first) at which
    each foot takes off / lands: -> (takeoff_RFs, takeoff_LFs, land_RFs, land_LFs)   (fulldynamic_talos.py:268-280)."""
    takeoff_RFs, takeoff_LFs, land_RFs, land_LFs = [], [], [], []
    for i in range(1, len(contact_phases)):
        cur, prev = tuple(contact_phases[i]), tuple(contact_phases[i - 1])
        if cur == LEFT_ONLY and prev == DOUBLE:
            takeoff_RFs.append(i + horizon)
        elif cur == RIGHT_ONLY and prev == DOUBLE:
            takeoff_LFs.append(i + horizon)
        elif cur == DOUBLE and prev == LEFT_ONLY:
            land_RFs.append(i + horizon)
        elif cur == DOUBLE and prev == RIGHT_ONLY:
            land_LFs.append(i + horizon)
    return takeoff_RFs, takeoff_LFs, land_RFs, land_LFs


def scan_list(countdowns):
    """One tick passes: every countdown decreases, an expired head is dropped (in place; talos_utils.py:350-354)."""
    for i in range(len(countdowns)):
        countdowns[i] -= 1
    if countdowns and countdowns[0] == -1:
        del countdowns[0]


def update_timings(land_LFs, land_RFs, takeoff_LFs, takeoff_RFs):
    """Advance the four countdown lists by one tick and return the next events (``-1`` = none pending), in the order the
    scripts unpack them: takeoff_RF, takeoff_LF, land_RF, land_LF   (talos_utils.py:356-373)."""
    for lst in (land_LFs, land_RFs, takeoff_LFs, takeoff_RFs):
        scan_list(lst)

    def head(lst):
        return lst[0] if lst else -1
    return head(takeoff_RFs), head(takeoff_LFs), head(land_RFs), head(land_LFs)


def shapeState(q_current, v_current, nq, nxq, cj_ids):
    """The simulator's full state reduced to the controlled joints: base pose and twist as they are, then the position / velocity of
    every controlled joint (``cj_ids``: joint ids of the full model, the free flyer's id <= 1 skipped) in list order
    (talos_utils.py:337-348).  ``nq``: configuration dimension of the REDUCED model, ``nxq`` = nq + nv of it."""
    q_current, v_current = np.asarray(q_current, dtype=float), np.asarray(v_current, dtype=float)
    ids = np.array([j for j in cj_ids if j > 1], dtype=int)
    x = np.zeros(nxq)
    x[:7] = q_current[:7]
    x[nq:nq + 6] = v_current[:6]
    x[7:7 + ids.size] = q_current[ids + 5]
    x[nq + 6:nq + 6 + ids.size] = v_current[ids + 4]
    return x


def compute_ID_references(space, rmodel, rdata, LF_id, RF_id, base_id, torso_id, x0_multibody, x_measured, LF_refs, RF_refs, dt):
    """Task errors of the inverse-dynamics QP (centroidal_talos.py:408; talos_utils.py:375-402): posture, foot placement (position and
    ``log3`` orientation), base / torso orientation against the RIGHT foot reference's rotation, and their rates from the next
    reference sample.  ``rdata`` must hold the forward kinematics (with velocities) and frame placements at ``x_measured``.  Returns
    q_diff, dq_diff, LF_diff, dLF_diff, RF_diff, dRF_diff, base_diff, dbase_diff, torso_diff, dtorso_diff."""
    from .robot import minipin as pin
    nv = rmodel.nv
    d = -np.asarray(space.difference(x0_multibody, x_measured), dtype=float)
    q_diff, dq_diff = d[:nv], d[nv:]

    def pose_err(ref0, ref1, fid):
        vel = pin.getFrameVelocity(rmodel, rdata, fid, pin.LOCAL)
        M = rdata.oMf[fid]
        e, de = np.zeros(6), np.zeros(6)
        e[:3] = ref0.translation - M.translation
        e[3:] = -pin.log3(ref0.rotation.T @ M.rotation)
        de[:3] = (ref1.translation - ref0.translation) / dt - vel.linear
        de[3:] = pin.log3(ref0.rotation.T @ ref1.rotation) / dt - vel.angular
        return e, de

    LF_diff, dLF_diff = pose_err(LF_refs[0], LF_refs[1], LF_id)
    RF_diff, dRF_diff = pose_err(RF_refs[0], RF_refs[1], RF_id)
    yaw_rate = pin.log3(RF_refs[0].rotation.T @ RF_refs[1].rotation) / dt

    def orient_err(fid):
        return (-pin.log3(RF_refs[0].rotation.T @ rdata.oMf[fid].rotation),
                yaw_rate - pin.getFrameVelocity(rmodel, rdata, fid, pin.LOCAL).angular)

    base_diff, dbase_diff = orient_err(base_id)
    torso_diff, dtorso_diff = orient_err(torso_id)
    return q_diff, dq_diff, LF_diff, dLF_diff, RF_diff, dRF_diff, base_diff, dbase_diff, torso_diff, dtorso_diff


# ---- swing curve -----------------------------------------------------------------------------------------------------

_BINOM8 = np.array([comb(8, i) for i in range(9)], dtype=float)


def bezier_eval(control_points, s):
    """Degree-(m-1) Bezier curve with control points ``control_points`` (3 x m) at parameter(s) ``s`` in [0, 1]."""
    P = np.asarray(control_points, dtype=float)
    m = P.shape[1] - 1
    s = np.atleast_1d(np.asarray(s, dtype=float))
    binom = _BINOM8 if m == 8 else np.array([comb(m, i) for i in range(m + 1)], dtype=float)
    i = np.arange(m + 1)
    basis = binom[None, :] * s[:, None] ** i[None, :] * (1.0 - s[:, None]) ** (m - i)[None, :]  # Bernstein
    # the control points are added one after the other: the same bits whether one parameter value is evaluated or a hundred (a matrix
    # product rounds differently for different shapes — a one-by-one evaluation, as talos_utils.py:303-318 makes it, must equal the batched one)
    out = np.zeros((s.size, P.shape[0]))
    for j in range(m + 1):
        out += basis[:, j:j + 1] * P[:, j][None, :]
    return out[0] if out.shape[0] == 1 else out


def swing_control_points(p_init, p_final, apex):
    """The 9 control points of talos_utils.py:276-285: 4 x start, lifted point at 3/4 start + 1/4 end, 4 x end."""
    p_init = np.asarray(p_init, dtype=float)
    p_final = np.asarray(p_final, dtype=float)
    wps = np.zeros((3, 9))
    wps[:, :4] = p_init[:, None]
    wps[:, 4] = 0.75 * p_init + 0.25 * p_final
    wps[2, 4] += apex
    wps[:, 5:] = p_final[:, None]
    return wps


def _log3(R):
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(c)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if th < 1e-10:
        return 0.5 * w
    if np.pi - th < 1e-6:  # near pi: axis from the diagonal
        A = (R + np.eye(3)) / 2.0
        ax = np.sqrt(np.maximum(np.diag(A), 0.0))
        k = int(np.argmax(ax))
        ax = A[:, k] / ax[k]
        return th * ax / np.linalg.norm(ax)
    return th / (2.0 * np.sin(th)) * w


def _exp3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-10:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1.0 - np.cos(th)) / th ** 2 * (K @ K)


def slerp_rotation(R0, R1, s):
    """Geodesic interpolation R0 exp(s log(R0^T R1)) — what ndcurves' SE3Curve does for the rotation part."""
    return np.asarray(R0) @ _exp3(s * _log3(np.asarray(R0).T @ np.asarray(R1)))


class SwingCurve:
    """Swing-foot placement curve on s in [0, 1]."""

    def __init__(self, pose_init, pose_final, apex):
        self.wps = swing_control_points(pose_init.translation, pose_final.translation, apex)
        self.R0 = np.array(pose_init.rotation, dtype=float)
        self.R1 = np.array(pose_final.rotation, dtype=float)

    def translation(self, s):
        return bezier_eval(self.wps, s)

    def rotation(self, s):
        return slerp_rotation(self.R0, self.R1, s)


def yaw_rotation(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    return np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])


def extract_yaw(R):
    return float(np.arctan2(R[1, 0], R[0, 0]))


class FootTrajectory:
    """Horizon-long placement references of both feet, regenerated every tick (interface of talos_utils.footTrajectory:
    same constructor arguments, ``updateForward`` and ``updateTrajectory``)."""

    def __init__(self, start_pose_left, start_pose_right, T_ss, T_ds, nsteps, swing_apex, x_forward, y_forward, foot_angle, y_gap, z_height):
        self.translationRight = np.array([x_forward, -y_gap - y_forward, z_height], dtype=float)
        self.translationLeft = np.array([x_forward, y_gap, z_height], dtype=float)
        self.rotationDiff = yaw_rotation(foot_angle)
        self.start_pose_left = start_pose_left
        self.start_pose_right = start_pose_right
        self.final_pose_left = start_pose_left
        self.final_pose_right = start_pose_right
        self.T_ds, self.T_ss, self.nsteps, self.swing_apex = T_ds, T_ss, nsteps, swing_apex

    def updateForward(self, x_f_left, x_f_right, y_gap, y_forward, z_height_left, z_height_right, swing_apex):
        self.translationRight = np.array([x_f_right, -y_gap - y_forward, z_height_right], dtype=float)
        self.translationLeft = np.array([x_f_left, y_gap, z_height_left], dtype=float)
        self.swing_apex = swing_apex

    # next footholds: the swing foot lands beside the stance foot (offset in the stance foot's yaw frame), and the foot
    # after that beside the new foothold
    def _plan_right_then_left(self, LF_pose, RF_pose):
        self.start_pose_right = RF_pose.copy()
        self.final_pose_right = LF_pose.copy()
        self.final_pose_right.translation = self.final_pose_right.translation + yaw_rotation(extract_yaw(LF_pose.rotation)) @ self.translationRight
        self.final_pose_right.rotation = self.rotationDiff @ self.final_pose_right.rotation
        self.start_pose_left = LF_pose.copy()
        self.final_pose_left = self.final_pose_right.copy()
        self.final_pose_left.translation = self.final_pose_left.translation + yaw_rotation(extract_yaw(self.final_pose_right.rotation)) @ self.translationLeft

    def _plan_left_then_right(self, LF_pose, RF_pose):
        self.start_pose_left = LF_pose.copy()
        self.final_pose_left = RF_pose.copy()
        self.final_pose_left.translation = self.final_pose_left.translation + yaw_rotation(extract_yaw(RF_pose.rotation)) @ self.translationLeft
        self.start_pose_right = RF_pose.copy()
        self.final_pose_right = self.final_pose_left.copy()
        self.final_pose_right.translation = self.final_pose_right.translation + yaw_rotation(extract_yaw(self.final_pose_left.rotation)) @ self.translationRight
        self.final_pose_right.rotation = self.rotationDiff @ self.final_pose_right.rotation

    def updateTrajectory(self, takeoff_RF, takeoff_LF, land_RF, land_LF, LF_pose, RF_pose):
        """-> (LF_refs, RF_refs): ``nsteps`` placements each, knot j = j ticks ahead.  A foot with no landing pending is
        pinned at its measured pose; a foot about to take off (countdown inside the double-support window) gets a new
        foothold; while it swings the reference follows the curve, clamped to the end poses outside the swing."""
        if land_LF < 0:
            self.start_pose_left = LF_pose.copy()
            self.final_pose_left = LF_pose.copy()
        if land_RF < 0:
            self.start_pose_right = RF_pose.copy()
            self.final_pose_right = RF_pose.copy()
        if 0 <= takeoff_RF < self.T_ds:
            self._plan_right_then_left(LF_pose, RF_pose)
        if 0 <= takeoff_LF < self.T_ds:
            self._plan_left_then_right(LF_pose, RF_pose)
        left = self._horizon_refs(land_LF, self.start_pose_left, self.final_pose_left)
        right = self._horizon_refs(land_RF, self.start_pose_right, self.final_pose_right)
        return left, right

    def _horizon_refs(self, time_to_land, pose_init, pose_final):
        if time_to_land <= -1:
            return [pose_init for _ in range(self.nsteps)]
        curve = SwingCurve(pose_init, pose_final, self.swing_apex)
        ts = np.arange(time_to_land, time_to_land - self.nsteps, -1)  # ticks left until landing at each knot
        swing = np.nonzero((ts > 0) & (ts <= self.T_ss))[0]
        # every knot inside the swing in one evaluation: Bernstein basis (k x 9) times the control points; the rotation moves on the
        # geodesic from R0 to R1 — one logarithm for the curve, one exponential per knot (none if the two rotations are the same)
        if swing.size:
            svals = (self.T_ss - ts[swing]).astype(float) / float(self.T_ss)
            trans = np.atleast_2d(bezier_eval(curve.wps, svals))
            w = _log3(curve.R0.T @ curve.R1)
            still = not np.any(w)
        out = []
        si = 0
        for t in ts:
            if t <= 0:
                out.append(pose_final)
            elif t > self.T_ss:
                out.append(pose_init)
            else:
                pose = pose_init.copy()
                pose.translation = trans[si].copy()
                pose.rotation = curve.R0.copy() if still else curve.R0 @ _exp3(svals[si] * w)
                si += 1
                out.append(pose)
        return out


# ---- the same generator for B robots at once (ensembles with per-instance references) ---------------------------------------------
def _yaw_rotation_batch(yaw):
    c, s = np.cos(yaw), np.sin(yaw)
    R = np.zeros(yaw.shape + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 1, 0], R[..., 1, 1], R[..., 2, 2] = c, -s, s, c, 1.0
    return R


def _log3_batch(R):
    """Rotation vectors of R (..., 3, 3) away from angle pi (foot yaw differences are small)."""
    c = np.clip((np.trace(R, axis1=-2, axis2=-1) - 1.0) / 2.0, -1.0, 1.0)
    th = np.arccos(c)
    w = np.stack([R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], axis=-1)
    small = th < 1e-10
    scale = np.where(small, 0.5, th / (2.0 * np.sin(np.where(small, 1.0, th))))
    return scale[..., None] * w


def _exp3_batch(w):
    th = np.linalg.norm(w, axis=-1)
    K = np.zeros(w.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0], K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -w[..., 2], w[..., 1], w[..., 2], -w[..., 0], -w[..., 1], w[..., 0]
    small = th < 1e-10
    ths = np.where(small, 1.0, th)
    a = np.where(small, 1.0, np.sin(ths) / ths)[..., None, None]
    b = np.where(small, 0.0, (1.0 - np.cos(ths)) / ths ** 2)[..., None, None]
    return np.eye(3) + a * K + b * (K @ K)


class FootTrajectoryBatch:
    """``FootTrajectory`` for B robots in one set of arrays: poses are (R [B, 3, 3], p [B, 3]) pairs, ``updateTrajectory`` returns the
    references of both feet as [B, nsteps, 12] blocks (rotation row-major, then translation: the layout of the stage parameter tables).
    Same rules, same countdown arguments (the contact schedule is shared by the robots of an ensemble); instance b gets what the
    scalar class would give for its measured poses."""

    def __init__(self, LF_R, LF_p, RF_R, RF_p, T_ss, T_ds, nsteps, swing_apex, x_forward, y_forward, foot_angle, y_gap, z_height):
        self.tR = np.array([x_forward, -y_gap - y_forward, z_height], dtype=float)
        self.tL = np.array([x_forward, y_gap, z_height], dtype=float)
        self.rotationDiff = yaw_rotation(foot_angle)
        self.sL, self.fL = (LF_R.copy(), LF_p.copy()), (LF_R.copy(), LF_p.copy())
        self.sR, self.fR = (RF_R.copy(), RF_p.copy()), (RF_R.copy(), RF_p.copy())
        self.T_ds, self.T_ss, self.nsteps, self.swing_apex = T_ds, T_ss, nsteps, swing_apex
        self.floor_z = None  # EnsembleMPC.enable_walk(floor=...): no foothold is planned below this height (mpc_walk_config.floor_z)

    def updateForward(self, x_f_left, x_f_right, y_gap, y_forward, z_height_left, z_height_right, swing_apex):
        self.tR = np.array([x_f_right, -y_gap - y_forward, z_height_right], dtype=float)
        self.tL = np.array([x_f_left, y_gap, z_height_left], dtype=float)
        self.swing_apex = swing_apex

    @staticmethod
    def _yaw(R):
        return np.arctan2(R[:, 1, 0], R[:, 0, 0])

    def _beside(self, pose, offset, rotate):
        R, p = pose
        p2 = p + np.einsum("bij,j->bi", _yaw_rotation_batch(self._yaw(R)), offset)
        if self.floor_z is not None:
            p2[:, 2] = np.maximum(p2[:, 2], self.floor_z)
        R2 = (self.rotationDiff @ R) if rotate else R.copy()
        return R2, p2

    def updateTrajectory(self, takeoff_RF, takeoff_LF, land_RF, land_LF, LF_R, LF_p, RF_R, RF_p):
        LF, RF = (LF_R, LF_p), (RF_R, RF_p)
        cp = lambda P: (P[0].copy(), P[1].copy())
        if land_LF < 0:
            self.sL, self.fL = cp(LF), cp(LF)
        if land_RF < 0:
            self.sR, self.fR = cp(RF), cp(RF)
        if 0 <= takeoff_RF < self.T_ds:  # right foot next to the left one, then the left foot next to that foothold
            self.sR = cp(RF)
            self.fR = self._beside(LF, self.tR, True)
            self.sL = cp(LF)
            self.fL = self._beside(self.fR, self.tL, False)
        if 0 <= takeoff_LF < self.T_ds:
            self.sL = cp(LF)
            self.fL = self._beside(RF, self.tL, False)
            self.sR = cp(RF)
            self.fR = self._beside(self.fL, self.tR, True)
        return self._horizon_refs(land_LF, self.sL, self.fL), self._horizon_refs(land_RF, self.sR, self.fR)

    def _horizon_refs(self, time_to_land, init, final):
        B, N = init[0].shape[0], self.nsteps
        flat = lambda P: np.concatenate([P[0].reshape(B, 9), P[1]], axis=1)
        out = np.empty((B, N, 12))
        if time_to_land <= -1:
            out[:] = flat(init)[:, None, :]
            return out
        ts = np.arange(time_to_land, time_to_land - N, -1)
        out[:, ts <= 0] = flat(final)[:, None, :]
        out[:, ts > self.T_ss] = flat(init)[:, None, :]
        swing = np.nonzero((ts > 0) & (ts <= self.T_ss))[0]
        if swing.size:
            s = (self.T_ss - ts[swing]).astype(float) / float(self.T_ss)
            m = 8
            i = np.arange(m + 1)
            basis = _BINOM8[None, :] * s[:, None] ** i[None, :] * (1.0 - s[:, None]) ** (m - i)[None, :]  # [k, 9]
            wps = np.empty((B, 3, 9))
            wps[:, :, :4] = init[1][:, :, None]
            wps[:, :, 4] = 0.75 * init[1] + 0.25 * final[1]
            wps[:, 2, 4] += self.swing_apex
            wps[:, :, 5:] = final[1][:, :, None]
            trans = np.einsum("ks,bcs->bkc", basis, wps)
            w = _log3_batch(np.einsum("bji,bjk->bik", init[0], final[0]))
            if np.any(w):
                Rk = init[0][:, None] @ _exp3_batch(s[None, :, None] * w[:, None, :])
            else:
                Rk = np.broadcast_to(init[0][:, None], (B, swing.size, 3, 3))
            out[:, swing, :9] = Rk.reshape(B, swing.size, 9)
            out[:, swing, 9:] = trans
        return out
