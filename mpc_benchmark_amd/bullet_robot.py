"""Headless stand-in for the reference's ``bullet_robot.BulletRobot`` ("next" row N2 of SURVEY.md §8f): the methods the three scripts call
(bullet_robot.py:15-24 constructor, :90 initializeJoints, :138-145 execute, :157-162 apply_force, :164-168 changeCamera, :172-196
measureState, plus the marker calls) with the PyBullet physics replaced by the solver library's own rigid-contact dynamics
(``mpc_simulate_torque``, include/mpc_abi.h): no GUI, no URDF, no PyBullet.

What it simulates: the CONTROLLED joints of the complete model (the others stay locked at the configuration handed to
``initializeJoints`` — in PyBullet they are held by the default position controller, bullet_robot.py:71-73), one semi-implicit Euler step
of ``simuStep`` per ``execute(torques)``, with the feet that stand on the ground held by 6-D rigid contacts with Baumgarte correction
(the contact model of fulldynamic_talos.py:84-96).  A foot is "on the ground" while its sole is within ``ground_tol`` of the ground
plane z = 0 ... and pushes on it: a contact whose normal force stays below ``-release_force`` for ``release_steps`` consecutive steps is released (a one-step
transient — the torque jump of the low-level QP when a stage changes its contact state — only unloads a real sole for a millisecond, it does not
lift it), a free foot that has left the ground and comes back to it, or that sinks below the ground plane, is caught there (its world-side
placement is re-captured at the landing pose).  That is a deliberately simple contact rule — enough to close the loop around the MPC headlessly
and deterministically; it is not a physics engine.

Differences from PyBullet worth knowing: ``measureState`` returns the base velocity in the LOCAL frame of the base (Pinocchio's
convention, which is what the scripts assume when they copy it into the state, talos_utils.py:337-348); PyBullet reports it in the
world frame."""
from __future__ import annotations

import numpy as np

from . import _capi as K
from .aligator import _core as core
from .aligator import dynamics as _dyn
from .aligator import manifolds as _manifolds
from .robot import minipin as pin


def _sim_options():
    """option block of a simulator handle (never solves: only the library's own consistency checks look at it)"""
    o = K.default_options(1e-5, 1e-8)
    o.force_initial_condition, o.rollout_linear = 1, 1
    return o


class BulletRobot:
    record_default = False  # tools: keep (state, contact flags, sole heights) of every step in ``history``

    def __init__(self, controlledJoints, modelPath=None, URDF_filename=None, simuStep=1e-3, rmodelComplete=None, robotPose=(0.0, 0.0, 1.01927),
                 inertiaOffset=True, talos=True, library=None, contact_frames=("left_sole_link", "right_sole_link"), ground_tol=5e-3, release_steps=5, release_force=1.0):
        if rmodelComplete is None:
            raise ValueError("the complete robot model is needed (5th positional argument, as in the scripts)")
        self._lib = library
        self.dt = float(simuStep)
        self.complete = rmodelComplete
        self.controlled = [n for n in controlledJoints if n not in ("universe", "root_joint")]
        self.contact_frames = tuple(contact_frames)
        self.ground_tol = float(ground_tol)
        self.release_steps = int(release_steps)
        self.release_force = float(release_force)  # N: the ground "pulls" when the normal force is below minus this
        self.robotPose = np.asarray(robotPose, dtype=float)
        self.localInertiaPos = np.zeros(3)
        self._native = None
        self._pending_force = None
        self.markers = None
        self.camera = None
        self.steps = 0
        self.trace_from = None     # tools: print contact forces from this step on
        self.max_steps = None      # tools: stop a script's endless loop after this many execute() calls
        self.history = []          # (q, v) after every step when ``record`` is set
        self.record = bool(self.record_default)

    # -- model ------------------------------------------------------------------------------------------------------------------
    def initializeJoints(self, q0CompleteStart):
        mc = self.complete
        q0 = np.array(q0CompleteStart, dtype=float).reshape(-1)
        self.q_complete = q0.copy()
        self.v_complete = np.zeros(mc.nv)
        keep = set(self.controlled)
        locked = [j for j in range(2, mc.njoints) if mc.names[j] not in keep]
        self.model = pin.buildReducedModel(mc, locked, q0) if locked else mc
        m = self.model
        # complete <-> reduced coordinate maps (joint i >= 2 of the complete model has q index i + 5, v index i + 4)
        self._qmap = [(mc.joints[mc.getJointId(n)].idx_q, m.joints[m.getJointId(n)].idx_q) for n in m.names[2:]]
        self._vmap = [(mc.joints[mc.getJointId(n)].idx_v, m.joints[m.getJointId(n)].idx_v) for n in m.names[2:]]
        self.x = np.zeros(m.nq + m.nv)
        self.x[:7] = q0[:7]
        for src, dst in self._qmap:
            self.x[dst] = q0[src]
        self.data = m.createData()
        pin.framesForwardKinematics(m, self.data, self.x[:m.nq])
        self.frame_ids = [m.getFrameId(n) for n in self.contact_frames]
        self.ground_z = min(float(self.data.oMf[f].translation[2]) for f in self.frame_ids)
        self.in_contact = [True, True]
        self._z_prev = [float(self.data.oMf[f].translation[2]) for f in self.frame_ids]
        self._lifted = [False, False]
        self._pulling = [0, 0]   # consecutive steps with a negative normal force
        self._contact_pose = [self.data.oMf[f].copy() for f in self.frame_ids]
        self._build_native()

    def _contact_models(self):
        m = self.model
        cms = []
        for name, fid, pose in zip(self.contact_frames, self.frame_ids, self._contact_pose):
            cm = pin.RigidConstraintModel(pin.ContactType.CONTACT_6D, m, m.frames[fid].parentJoint, m.frames[fid].placement, 0, pose, pin.LOCAL)
            cm.corrector.Kp[:] = (0, 0, 10, 0, 0, 0)      # fulldynamic_talos.py:93-94
            cm.corrector.Kd[:] = (50, 50, 50, 50, 50, 50)
            cm.name = name
            cms.append(cm)
        return cms

    def _build_native(self):
        m = self.model
        nu = m.nv - 6
        space = _manifolds.MultibodyPhaseSpace(m)
        self._ctx = ctx = core.LoweringContext()
        cms = self._contact_models()
        act = np.eye(m.nv, nu, -6)
        prox = pin.ProximalSettings(1e-9, 1e-10, 1)
        self._stage_tables = {}
        for mask in ((True, True), (True, False), (False, True)):
            ode = _dyn.MultibodyConstraintFwdDynamics(space, act, [c for c, on in zip(cms, mask) if on], prox)
            cost = core.CostStack(space, nu)
            cost.addCost(core.QuadraticControlCost(space, np.zeros(nu), np.eye(nu)))
            stage = core.StageModel(cost, _dyn.IntegratorSemiImplEuler(ode, self.dt))
            self._stage_tables[mask] = core.lower_stage(ctx, stage.cost, stage.dynamics, stage.constraints)
        tcost = core.CostStack(space, nu)
        tcost.addCost(core.QuadraticStateCost(space, nu, space.neutral(), np.eye(space.ndx)))
        term = core.lower_stage(ctx, tcost, None, core._ConstraintStack())
        if self._native is None:
            lib = self._lib if self._lib is not None else K.load_hip_library()
            d = K.MpcDims()
            d.horizon, d.batch, d.space = 1, 1, K.SPACE_MULTIBODY
            d.nx, d.ndx, d.nu, d.nc_max = space.nx, space.ndx, nu, 1
            d.max_stage_ints = 8 + 8 * 24
            d.max_stage_doubles = max(t[1].size for t in self._stage_tables.values()) + term[1].size + 1024
            d.device = 0
            self._native = K.NativeSolver(lib, d)
            self._native.set_options(_sim_options())
        self._native.set_model(*ctx.model_tables())
        self._native.set_stage(1, *term)
        self._mask_uploaded = None

    def _upload_mask(self):
        mask = tuple(bool(c) for c in self.in_contact)
        if mask == (False, False):
            raise RuntimeError("headless BulletRobot: both feet off the ground (flight phases are not simulated)")
        if mask != self._mask_uploaded:
            self._native.set_stage(0, *self._stage_tables[mask])
            self._mask_uploaded = mask

    # -- simulation -------------------------------------------------------------------------------------------------------------
    def execute(self, torques):
        if self.max_steps is not None and self.steps >= self.max_steps:
            raise StopIteration("headless BulletRobot: step budget of %d reached" % self.max_steps)
        tau = np.asarray(torques, dtype=float).reshape(-1)
        m = self.model
        if tau.size != m.nv - 6:
            raise ValueError("expected %d joint torques, got %d" % (m.nv - 6, tau.size))
        self._upload_mask()
        if self._pending_force is not None:  # (apply_force: the push of fulldynamic_talos.py:524-526 — generalized force on the base)
            raise NotImplementedError("apply_force: use mpc_simulate_push through EnsembleMPC for disturbance runs")
        x, wr = self._native.simulate_torque(self.x, tau, 1, self.dt, wrenches=True)
        self.x = x[0]
        self.steps += 1
        if self.trace_from is not None and self.steps >= self.trace_from:
            import sys
            sys.stderr.write("   [sim] step %d in_contact %s fz L %.2f R %.2f z %s\n" % (self.steps, self.in_contact, wr[0][0][2], wr[0][1][2], ["%.4f" % v for v in self._z_prev]))
        self._update_contacts(wr[0])
        if self.record:
            self.history.append((self.x.copy(), tuple(self.in_contact), tuple(self._z_prev)))

    def _update_contacts(self, wrenches):
        """Unilateral contact by rule: an active contact whose normal force turned negative is released (the ground cannot pull); a free
        foot that is moving down and reaches the ground plane is caught at the pose it lands with (flattened onto the ground)."""
        m = self.model
        pin.framesForwardKinematics(m, self.data, self.x[:m.nq])
        relanded = False
        for i, fid in enumerate(self.frame_ids):
            z = float(self.data.oMf[fid].translation[2])
            if self.in_contact[i]:
                self._pulling[i] = self._pulling[i] + 1 if wrenches[i][2] < -self.release_force else 0  # (an unloaded sole, 0 +- round-off, rests on the ground)
                if self._pulling[i] >= self.release_steps and sum(self.in_contact) > 1:
                    self.in_contact[i] = False
                    self._lifted[i] = False
                    self._pulling[i] = 0
            elif z > self.ground_z + 2.0 * self.ground_tol:
                self._lifted[i] = True  # (a released foot is caught again only after it has really left the ground ...)
            elif (z <= self.ground_z + self.ground_tol and self._lifted[i]) or (z < self.ground_z and z < self._z_prev[i]):  # (... or sinks into it)
                pose = self.data.oMf[fid].copy()
                pose.translation[2] = self.ground_z
                yaw = np.arctan2(pose.rotation[1, 0], pose.rotation[0, 0])
                c, s = np.cos(yaw), np.sin(yaw)
                pose.rotation = np.array([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]])
                self._contact_pose[i] = pose
                self.in_contact[i] = True
                relanded = True
            self._z_prev[i] = z
        if relanded:
            self._build_native()

    def measureState(self):
        """-> (q, v) of the COMPLETE model (bullet_robot.py:172-196): locked joints at their initial positions, zero velocity."""
        m = self.model
        q, v = self.q_complete.copy(), self.v_complete.copy()
        q[:7] = self.x[:7]
        v[:6] = self.x[m.nq:m.nq + 6]
        for src, dst in self._qmap:
            q[src] = self.x[dst]
        for src, dst in self._vmap:
            v[src] = self.x[m.nq + dst]
        return q, v

    def resetState(self, q0Start):
        m = self.model
        self.x[:m.nq] = np.asarray(q0Start, dtype=float)[:m.nq]
        self.x[m.nq:] = 0.0

    def apply_force(self, force, position):
        self._pending_force = (np.asarray(force, dtype=float), np.asarray(position, dtype=float))

    # -- GUI calls of the scripts: recorded, nothing to draw ------------------------------------------------------------------------
    def changeCamera(self, cameraDistance, cameraYaw, cameraPitch, cameraTargetPos):
        self.camera = (cameraDistance, cameraYaw, cameraPitch, tuple(cameraTargetPos))

    def showTargetToTrack(self, LF_pose, RF_pose):
        self.markers = (np.array(LF_pose.translation), np.array(RF_pose.translation))

    def moveMarkers(self, LF_trans, RF_trans):
        self.markers = (np.array(LF_trans), np.array(RF_trans))

    def showQuadrupedFeet(self, *poses):
        self.markers = tuple(np.array(p.translation) for p in poses)

    def moveQuadrupedFeet(self, *trans):
        self.markers = tuple(np.array(t) for t in trans)

    def setFrictionCoefficients(self, link_id, lateral_friction, spinning_friction):
        pass

    def addStairs(self, path, position, orientation):
        raise NotImplementedError("headless BulletRobot: flat ground only")

    def close(self):
        if self._native is not None:
            self._native.close()
            self._native = None
