"""Kinodynamic Talos walking OCP — the problem kinodynamic_talos.py builds (lines 37-180, 183-292), expressed through
the ``aligator`` mirror.  x = (q, v) on MultibodyPhaseSpace, u = [left wrench (6), right wrench (6), joint
accelerations (nv - 6)]; the base acceleration follows from the centroidal momentum balance."""
from __future__ import annotations

import numpy as np

from .. import aligator
from ..aligator import constraints, dynamics, manifolds
from ..robot import minipin as pin
from . import common

T_DS, T_SS, TOTAL_STEPS = 20, 80, 3  # kinodynamic_talos.py:183-190

# tangent-space state weights (x10), kinodynamic_talos.py:74-88
_W_POS = {"base": [0, 0, 1000, 1000, 1000, 1000], "leg": [0.1] * 6, "torso": [1, 1000], "arm": [1, 1, 10, 10]}
_W_VEL = {"base": [0.1, 0.1, 0.1, 1000, 1000, 1000], "leg": [1] * 6, "torso": [0.1, 100], "arm": [10] * 4}


def state_weights(model):
    pos, vel = list(_W_POS["base"]), list(_W_VEL["base"])
    for name in list(model.names)[2:]:
        parts = name.split("_")
        if parts[0] == "leg":
            i = int(parts[2]) - 1
            pos.append(_W_POS["leg"][i]); vel.append(_W_VEL["leg"][i])
        elif parts[0] == "torso":
            i = int(parts[1]) - 1
            pos.append(_W_POS["torso"][i]); vel.append(_W_VEL["torso"][i])
        else:
            i = (int(parts[2]) - 1) % 4 if parts[0] == "arm" else 3
            pos.append(_W_POS["arm"][i]); vel.append(_W_VEL["arm"][i])
    return 10.0 * np.array(pos + vel, dtype=float)


class KinodynamicProblem:
    def __init__(self, horizon=common.HORIZON, dt=common.DT, robot=None, complete_model=False):
        self.robot = rb = robot or common.Robot(complete=complete_model)
        m = rb.model
        self.horizon, self.dt = horizon, dt
        self.nv = m.nv
        self.nu = m.nv - 6 + 12                                   # kinodynamic_talos.py:43
        self.space = manifolds.MultibodyPhaseSpace(m)
        self.gravity = np.array([0.0, 0.0, -9.81])
        self.x0 = rb.x0.copy()
        self.w_x = np.diag(state_weights(m))
        w_force = np.concatenate(([0.001, 0.001, 0.01], np.full(3, 0.1)))
        self.w_u = np.diag(np.concatenate((w_force, w_force, np.full(m.nv - 6, 1e-4))))  # :89-98
        self.w_foot = 100000.0                                    # :99
        self.w_cent = np.diag([0.0, 0.0, 1.0, 0.1, 0.1, 10.0])    # :100-102
        self.w_centder = np.diag([0.0, 0.0, 0.0, 0.1, 0.1, 0.1])  # :103-105
        self.contact_phases = common.contact_schedule(T_DS, T_SS, TOTAL_STEPS, horizon)
        refs, f_full, f_half = common.force_reference_ramp(rb.mass, T_DS, T_SS, TOTAL_STEPS, horizon, self.nu)
        for j in range(T_DS):  # kinodynamic_talos.py:227-231
            u = np.zeros(self.nu)
            u[2] = f_half * (j + 1) / float(T_DS)
            u[8] = f_full * (T_DS - j) / float(T_DS) + f_half * j / float(T_DS)
            refs.append(u)
        for _ in range(2 * horizon):  # :233-237
            u = np.zeros(self.nu)
            u[2] = u[8] = f_half
            refs.append(u)
        self.urefs = refs
        self.t_mpc = len(self.contact_phases)
        f_ref = np.array([0.0, 0.0, rb.mass * 9.81 / 2.0, 0.0, 0.0, 0.0])
        self.u_init = np.concatenate((f_ref, f_ref, np.zeros(m.nv - 6)))  # :296

    def create_stage(self, contact_state, lf_pose, rf_pose, uforce):
        rb, m, nu = self.robot, self.robot.model, self.nu
        space = manifolds.MultibodyPhaseSpace(m)
        ndx = space.ndx
        lf_id, rf_id = rb.foot_frame_ids
        v_ref = pin.Motion()
        cost = aligator.CostStack(space, nu)
        cost.addCost("state_cost", aligator.QuadraticStateCost(space, nu, self.x0, self.w_x))
        cost.addCost("control_cost", aligator.QuadraticControlCost(space, uforce, self.w_u))
        w_lf = self.w_foot * np.eye(6) if contact_state[1] else np.zeros((6, 6))  # kinodynamic_talos.py:143-148
        w_rf = self.w_foot * np.eye(6) if contact_state[0] else np.zeros((6, 6))
        cost.addCost("centroidal_cost", aligator.QuadraticResidualCost(
            space, aligator.CentroidalMomentumResidual(ndx, nu, m, np.zeros(6)), self.w_cent))
        cost.addCost("centroidal_derivative_cost", aligator.QuadraticResidualCost(
            space, aligator.CentroidalMomentumDerivativeResidual(ndx, m, self.gravity, contact_state, [lf_id, rf_id], 6), self.w_centder))
        cost.addCost("left_sole_link_pose_cost", aligator.QuadraticResidualCost(
            space, aligator.FramePlacementResidual(ndx, nu, m, lf_pose, lf_id), w_lf))
        cost.addCost("right_sole_link_pose_cost", aligator.QuadraticResidualCost(
            space, aligator.FramePlacementResidual(ndx, nu, m, rf_pose, rf_id), w_rf))
        ode = dynamics.KinodynamicsFwdDynamics(space, m, self.gravity, contact_state, [lf_id, rf_id], 6)
        stage = aligator.StageModel(cost, dynamics.IntegratorSemiImplEuler(ode, self.dt))
        joint_fn = aligator.StateErrorResidual(space, nu, space.neutral())[6:self.nv]
        stage.addConstraint(joint_fn, constraints.BoxConstraint(-m.upperPositionLimit[7:], -m.lowerPositionLimit[7:]))  # :161-162
        for k, (active, fid) in enumerate(zip(contact_state, (lf_id, rf_id))):
            if active:  # :164-171
                cone = aligator.CentroidalWrenchConeResidual(ndx, nu, k, common.FRICTION_MU, common.FOOT_HALF_LENGTH, common.FOOT_HALF_WIDTH)
                stage.addConstraint(cone, constraints.NegativeOrthant())
                stage.addConstraint(aligator.FrameVelocityResidual(ndx, nu, m, v_ref, fid, pin.LOCAL), constraints.EqualityConstraintSet())
        return stage

    def terminal_com_constraint(self, com_target):
        fn = aligator.CenterOfMassTranslationResidual(self.space.ndx, self.nu, self.robot.model, com_target)
        return aligator.StageConstraint(fn, constraints.EqualityConstraintSet())

    def stage_key(self, t):
        return int(t)  # every tick has its own force reference (kinodynamic_talos.py:200-237)

    def stage_for_tick(self, t):
        lf, rf = self.robot.foot_placements
        return self.create_stage(self.contact_phases[t], lf.copy(), rf.copy(), self.urefs[t])

    def build(self, with_terminal_constraint=True):
        lf, rf = self.robot.foot_placements
        one = self.create_stage(self.contact_phases[0], lf.copy(), rf.copy(), self.urefs[0])
        problem = aligator.TrajOptProblem(self.x0, [one] * self.horizon, aligator.CostStack(self.space, self.nu))  # :274-275
        if with_terminal_constraint:
            problem.addTerminalConstraint(self.terminal_com_constraint(self.robot.com0))  # :276
        return problem

    def walk_spec(self):
        return {"T_SS": T_SS, "T_DS": T_DS, "x_forward": 0.3,                      # kinodynamic_talos.py:183-184, :257
                "kind": "pose", "state_key": "state_cost", "pose_keys": ("left_sole_link_pose_cost", "right_sole_link_pose_cost"), "terminal_feet": False,  # :384-385, :407-409
                "forward_rule": lambda takeoff_RF, takeoff_LF, land_RF, land_LF: land_RF == -1 and takeoff_RF == -1, "forward_z_left": 0.0,  # :368-370
                "setup_each_tick": False}                                                # :487: ``#solver.setup(problem)``

    def make_solver(self, **kw):
        solver = aligator.SolverProxDDP(1e-5, 1e-8, **kw)  # kinodynamic_talos.py:281-292
        solver.rollout_type = aligator.ROLLOUT_LINEAR
        solver.linear_solver_choice = aligator.LQ_SOLVER_PARALLEL
        solver.force_initial_condition = True
        solver.setNumThreads(8)
        solver.max_iters = 100
        return solver

    def initial_guess(self):
        return [self.x0.copy() for _ in range(self.horizon + 1)], [self.u_init.copy() for _ in range(self.horizon)]
