"""Centroidal Talos walking OCP — the problem centroidal_talos.py builds (lines 40-48, 100-116, 185-277),
expressed through the ``aligator`` mirror.  x = [com; linear momentum; angular momentum] (VectorSpace(9)),
u = two 6D contact wrenches."""
from __future__ import annotations

import numpy as np

from .. import aligator
from ..aligator import constraints, dynamics, manifolds
from . import common

T_DS, T_SS, TOTAL_STEPS = 20, 80, 1  # centroidal_talos.py:100-108


class CentroidalProblem:
    def __init__(self, horizon=common.HORIZON, dt=common.DT, robot=None):
        self.robot = robot or common.Robot()
        self.horizon, self.dt = horizon, dt
        self.nx, self.nu = 9, 12
        self.space = manifolds.VectorSpace(self.nx)
        self.gravity = np.array([0.0, 0.0, -9.81])
        self.x0 = self.space.neutral()
        self.x0[:3] = self.robot.com0
        self.u0 = np.zeros(self.nu)
        self.u0[2] = self.u0[8] = -self.gravity[2] * self.robot.mass / 2.0  # centroidal_talos.py:73-74
        # weights, centroidal_talos.py:187-200
        self.w_com = np.zeros((3, 3))
        self.w_linear_mom = np.diag([0.01, 0.01, 100.0])
        self.w_linear_acc = 0.01 * np.eye(3)
        self.w_angular_mom = np.diag([0.1, 0.1, 1000.0])
        self.w_angular_acc = 0.01 * np.eye(3)
        self.w_control = np.diag(np.tile(np.concatenate((np.full(3, 0.001), np.full(3, 0.1))), 2))

        self.contact_phases = common.contact_schedule(T_DS, T_SS, TOTAL_STEPS, horizon)
        refs, f_full, f_half = common.force_reference_ramp(self.robot.mass, T_DS, T_SS, TOTAL_STEPS, horizon, self.nu)
        for j in range(T_DS):  # centroidal_talos.py:159-163
            u = np.zeros(self.nu)
            u[8] = f_half * (j + 1) / float(T_DS)
            u[2] = f_full * (T_DS - j) / float(T_DS) + f_half * j / float(T_DS)
            refs.append(u)
        for _ in range(2 * horizon):  # centroidal_talos.py:165-169
            u = np.zeros(self.nu)
            u[2] = u[8] = f_half
            refs.append(u)
        self.urefs = refs
        self.t_mpc = len(self.contact_phases)

    def create_stage(self, contact_state, lf_pose, rf_pose, uref):
        rb = self.robot
        cmap = aligator.ContactMap(list(common.FOOT_FRAMES), contact_state, [lf_pose.translation, rf_pose.translation])
        cost = aligator.CostStack(self.space, self.nu)
        lin_acc = aligator.CentroidalAccelerationResidual(self.nx, self.nu, rb.mass, self.gravity, cmap, 6)
        ang_acc = aligator.AngularAccelerationResidual(self.nx, self.nu, rb.mass, self.gravity, cmap, 6)
        lin_mom = aligator.LinearMomentumResidual(self.nx, self.nu, np.zeros(3))
        ang_mom = aligator.AngularMomentumResidual(self.nx, self.nu, np.zeros(3))
        com = aligator.CentroidalCoMResidual(self.nx, self.nu, rb.com0)
        # keys as in centroidal_talos.py:224-240 ("state_cost" really is the control cost there)
        cost.addCost("state_cost", aligator.QuadraticControlCost(self.space, uref, self.w_control))
        cost.addCost("com_cost", aligator.QuadraticResidualCost(self.space, com, self.w_com))
        cost.addCost("linear_mom_cost", aligator.QuadraticResidualCost(self.space, lin_mom, self.w_linear_mom))
        cost.addCost("angular_mom_cost", aligator.QuadraticResidualCost(self.space, ang_mom, self.w_angular_mom))
        cost.addCost("angular_acc_cost", aligator.QuadraticResidualCost(self.space, ang_acc, self.w_angular_acc))
        cost.addCost("linear_acc_cost", aligator.QuadraticResidualCost(self.space, lin_acc, self.w_linear_acc))
        ode = dynamics.CentroidalFwdDynamics(self.space, rb.mass, self.gravity, cmap, 6)
        stage = aligator.StageModel(cost, dynamics.IntegratorEuler(ode, self.dt))
        for i, active in enumerate(contact_state):
            if active:
                cone = aligator.CentroidalWrenchConeResidual(self.space.ndx, self.nu, i, common.FRICTION_MU,
                                                             common.FOOT_HALF_LENGTH, common.FOOT_HALF_WIDTH)
                stage.addConstraint(cone, constraints.NegativeOrthant())
        return stage

    def stage_key(self, t):
        return int(t)  # every tick has its own force reference (centroidal_talos.py:132-169)

    def stage_for_tick(self, t):
        lf, rf = self.robot.foot_placements
        return self.create_stage(self.contact_phases[t], lf, rf, self.urefs[t])

    def build(self):
        lf, rf = self.robot.foot_placements
        stages = [self.create_stage(self.contact_phases[0], lf, rf, self.urefs[0]) for _ in range(self.horizon)]
        term_cost = aligator.CostStack(self.space, self.nu)  # empty, centroidal_talos.py:249
        return aligator.TrajOptProblem(self.x0, stages, term_cost)

    def walk_spec(self):
        return {"T_SS": T_SS, "T_DS": T_DS, "x_forward": 0.2,                      # centroidal_talos.py:100-101, :175
                "kind": "contact_poses", "terminal_feet": False,                  # :374-384 ; empty terminal cost, :249
                "forward_rule": lambda takeoff_RF, takeoff_LF, land_RF, land_LF: land_RF == -1, "forward_z_left": -0.01}  # :365-366

    def make_solver(self, **kw):
        solver = aligator.SolverProxDDP(1e-5, 1e-8, **kw)  # centroidal_talos.py:265-277
        solver.rollout_type = aligator.ROLLOUT_LINEAR
        solver.linear_solver_choice = aligator.LQ_SOLVER_PARALLEL
        solver.force_initial_condition = True
        solver.setNumThreads(2)
        solver.max_iters = 100
        return solver

    def initial_guess(self):
        return [self.x0.copy() for _ in range(self.horizon + 1)], [self.u0.copy() for _ in range(self.horizon)]
