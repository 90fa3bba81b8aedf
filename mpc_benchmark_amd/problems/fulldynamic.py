"""Whole-body (constrained forward dynamics) Talos walking OCP — the problem fulldynamic_talos.py builds
(lines 62-245, 248-266, 362-386), expressed through the ``aligator`` mirror.  6D contacts enter the
dynamics as rigid constraints with Baumgarte correction; x = (q, v) on MultibodyPhaseSpace, u = joint torques."""
from __future__ import annotations

import numpy as np

from .. import aligator
from ..aligator import constraints, dynamics, manifolds
from ..robot import minipin as pin
from . import common

T_DS, T_SS, TOTAL_STEPS = 30, 80, 3  # fulldynamic_talos.py:248-255

# tangent-space state weights, fulldynamic_talos.py:121-134 (reduced 22-joint model): per joint group
_W_POS = {"base": [0, 0, 0, 100, 100, 100], "leg": [0.1] * 6, "torso": [10, 10], "arm": [1.0]}
_W_VEL = {"base": [1] * 6, "leg": [0.1, 0.1, 0.1, 0.1, 0.01, 0.01], "torso": [10, 10], "arm": [1.0]}


def state_weights(model):
    """Diagonal of w_x in tangent order.  For the reduced model this reproduces the 56 numbers of
    fulldynamic_talos.py:121-134; on the complete model the extra arm/gripper/head joints get the arm weight."""
    pos, vel = list(_W_POS["base"]), list(_W_VEL["base"])
    for name in list(model.names)[2:]:
        parts = name.split("_")
        if parts[0] == "leg":
            i = int(parts[2]) - 1
            pos.append(_W_POS["leg"][i])
            vel.append(_W_VEL["leg"][i])
        elif parts[0] == "torso":
            i = int(parts[1]) - 1
            pos.append(_W_POS["torso"][i])
            vel.append(_W_VEL["torso"][i])
        else:
            pos.append(_W_POS["arm"][0])
            vel.append(_W_VEL["arm"][0])
    return np.array(pos + vel, dtype=float)


class FullDynamicsProblem:
    def __init__(self, horizon=common.HORIZON, dt=common.DT, robot=None, complete_model=False):
        self.robot = rb = robot or common.Robot(complete=complete_model)
        m = rb.model
        self.horizon, self.dt = horizon, dt
        self.nv, self.nu = m.nv, m.nv - 6
        self.space = manifolds.MultibodyPhaseSpace(m)
        self.x0 = rb.x0.copy()
        self.u0 = np.zeros(self.nu)
        self.act_matrix = np.eye(m.nv, self.nu, -6)                 # fulldynamic_talos.py:76
        self.prox_settings = pin.ProximalSettings(1e-9, 1e-10, 1)   # fulldynamic_talos.py:77
        self.constraint_models = []
        for name, fid, jid, oMf in zip(common.FOOT_FRAMES, rb.foot_frame_ids, rb.foot_joint_ids, rb.foot_placements):
            cm = pin.RigidConstraintModel(pin.ContactType.CONTACT_6D, m, jid, m.frames[fid].placement, 0, oMf, pin.LOCAL)
            cm.corrector.Kp[:] = (0, 0, 10, 0, 0, 0)                # fulldynamic_talos.py:93-94
            cm.corrector.Kd[:] = (50, 50, 50, 50, 50, 50)
            cm.name = name
            self.constraint_models.append(cm)
        self.w_x = np.diag(state_weights(m))
        self.w_u = 1e-4 * np.eye(self.nu)
        self.w_foot = 2000.0
        self.w_cent = np.diag([0.0, 0.0, 10.0, 0.0, 0.0, 10.0])
        self.w_forces = 1e-4 * np.eye(6)
        self.umax = m.effortLimit[6:].copy()
        f_half = rb.mass * 9.81 / 2.0
        self.force_ref = np.array([0.0, 0.0, f_half, 0.0, 0.0, 0.0])  # LF_force_refs[0], fulldynamic_talos.py:290-294
        self.contact_phases = common.contact_schedule(T_DS, T_SS, TOTAL_STEPS, horizon, final_left_step=True)
        self.t_mpc = len(self.contact_phases)

    def _dynamics(self, space, cs):
        if cs[0] and not cs[1]:
            cms = [self.constraint_models[0]]
        elif cs[1] and not cs[0]:
            cms = [self.constraint_models[1]]
        else:
            cms = self.constraint_models
        ode = dynamics.MultibodyConstraintFwdDynamics(space, self.act_matrix, cms, self.prox_settings)
        return dynamics.IntegratorSemiImplEuler(ode, self.dt), cms

    def create_stage(self, cs, lf_target, rf_target, lf_force=None, rf_force=None):
        rb, m, nu = self.robot, self.robot.model, self.nu
        lf_force = self.force_ref if lf_force is None else lf_force
        rf_force = self.force_ref if rf_force is None else rf_force
        space = manifolds.MultibodyPhaseSpace(m)
        ndx = space.ndx
        lf_id, rf_id = rb.foot_frame_ids
        dyn, cms = self._dynamics(space, cs)
        cost = aligator.CostStack(space, nu)
        cost.addCost(aligator.QuadraticStateCost(space, nu, self.x0, self.w_x))                       # component 0
        cost.addCost(aligator.QuadraticControlCost(space, self.u0, self.w_u))                         # component 1
        # the foot that is NOT the only support gets tracked: weights keyed on the *other* foot's contact flag
        w_lf = self.w_foot * np.eye(6) if cs[1] else np.zeros((6, 6))                                # fulldynamic_talos.py:177-182
        w_rf = self.w_foot * np.eye(6) if cs[0] else np.zeros((6, 6))
        cost.addCost(aligator.QuadraticResidualCost(space, aligator.CentroidalMomentumResidual(ndx, nu, m, np.zeros(6)), self.w_cent))
        cost.addCost(aligator.QuadraticResidualCost(space, aligator.FramePlacementResidual(ndx, nu, m, lf_target, lf_id), w_lf))  # 3
        cost.addCost(aligator.QuadraticResidualCost(space, aligator.FramePlacementResidual(ndx, nu, m, rf_target, rf_id), w_rf))  # 4
        for active, cm, fref in zip(cs, self.constraint_models, (lf_force, rf_force)):
            if active:
                res = aligator.ContactForceResidual(ndx, m, self.act_matrix, cms, self.prox_settings, fref, cm.name)
                cost.addCost(aligator.QuadraticResidualCost(space, res, self.w_forces))
        stage = aligator.StageModel(cost, dyn)
        stage.addConstraint(aligator.ControlErrorResidual(ndx, np.zeros(nu)), constraints.BoxConstraint(-self.umax, self.umax))
        joint_fn = aligator.StateErrorResidual(space, nu, space.neutral())[6:self.nv]
        # sign-flipped bounds reproduced verbatim from fulldynamic_talos.py:209
        stage.addConstraint(joint_fn, constraints.BoxConstraint(-m.upperPositionLimit[7:], -m.lowerPositionLimit[7:]))
        for active, cm in zip(cs, self.constraint_models):
            if active:
                cone = aligator.MultibodyWrenchConeResidual(ndx, m, self.act_matrix, cms, self.prox_settings, cm.name,
                                                            common.FRICTION_MU, common.FOOT_HALF_LENGTH, common.FOOT_HALF_WIDTH)
                stage.addConstraint(cone, constraints.NegativeOrthant())
        return stage

    def terminal_cost(self):
        rb, m, nu = self.robot, self.robot.model, self.nu
        ndx = self.space.ndx
        lf_id, rf_id = rb.foot_frame_ids
        lf, rf = rb.foot_placements
        tc = aligator.CostStack(self.space, nu)
        tc.addCost(aligator.QuadraticStateCost(self.space, nu, self.x0, self.w_x))
        tc.addCost(aligator.QuadraticResidualCost(self.space, aligator.CentroidalMomentumResidual(ndx, nu, m, np.zeros(6)), self.w_cent))
        tc.addCost(aligator.QuadraticResidualCost(self.space, aligator.FramePlacementResidual(ndx, nu, m, lf, lf_id), self.w_foot * np.eye(6)))
        tc.addCost(aligator.QuadraticResidualCost(self.space, aligator.FramePlacementResidual(ndx, nu, m, rf, rf_id), self.w_foot * np.eye(6)))
        return tc

    def terminal_com_constraint(self, com_target):
        fn = aligator.CenterOfMassTranslationResidual(self.space.ndx, self.nu, self.robot.model, com_target)
        return aligator.StageConstraint(fn, constraints.EqualityConstraintSet())

    def stage_for_tick(self, t):
        lf, rf = self.robot.foot_placements
        return self.create_stage(self.contact_phases[t], lf.copy(), rf.copy())

    def build(self, with_terminal_constraint=False):
        lf, rf = self.robot.foot_placements
        one = self.create_stage(self.contact_phases[0], lf.copy(), rf.copy())
        problem = aligator.TrajOptProblem(self.x0, [one] * self.horizon, self.terminal_cost())  # aliased list, fulldynamic_talos.py:371
        if with_terminal_constraint:
            problem.addTerminalConstraint(self.terminal_com_constraint(self.robot.com0))
        return problem

    def walk_spec(self):
        """What the loop body of the script does to the problem every tick (EnsembleMPC.enable_walk, problems/walking_loop.py)."""
        return {"T_SS": T_SS, "T_DS": T_DS, "x_forward": 0.0,                      # fulldynamic_talos.py:248-249, :352
                "kind": "pose", "state_key": 0, "pose_keys": (3, 4), "terminal_feet": True,       # :461-463, :499-510
                "forward_rule": lambda takeoff_RF, takeoff_LF, land_RF, land_LF: land_LF == -1, "forward_z_left": -0.01}  # :448-449

    def make_solver(self, **kw):
        solver = aligator.SolverProxDDP(1e-5, 1e-8, **kw)  # fulldynamic_talos.py:374-386
        solver.rollout_type = aligator.ROLLOUT_LINEAR
        solver.linear_solver_choice = aligator.LQ_SOLVER_PARALLEL
        solver.force_initial_condition = True
        solver.setNumThreads(8)
        solver.max_iters = 100
        return solver

    def initial_guess(self):
        return [self.x0.copy() for _ in range(self.horizon + 1)], [np.zeros(self.nu) for _ in range(self.horizon)]
