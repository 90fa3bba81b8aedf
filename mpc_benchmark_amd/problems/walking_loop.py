"""The receding-horizon loop body of fulldynamic_talos.py:438-550 written against the ``aligator`` mirror and the
reference generators of ``mpc_benchmark_amd.references`` — what a user of the reference script executes every 10 ms,
minus PyBullet: the next measured state is the state the previous solution predicted ("perfect-model" feedback,
SURVEY.md §8d).

Per tick: foot poses of the measured state -> take-off / landing countdowns -> swing references over the horizon
(``setReference`` on components 3 / 4 of every stage) -> ``replaceStageCircular`` with the stage of the contact phase that
enters the horizon -> terminal CoM constraint between the last foot references -> warm-start shift -> ``setup`` + ``run``.
"""
from __future__ import annotations

import numpy as np

from .. import references as refgen
from ..robot import minipin as pin


class WalkingMPCLoop:
    def __init__(self, problem_def, solver, swing_apex=0.15, x_forward=0.0, y_forward=0.0, foot_yaw=0.0, y_gap=0.18, z_height=0.0,
                 start_tick=0, move_terminal_constraint=True):
        """``problem_def``: a FullDynamicsProblem; ``solver``: its ``make_solver()`` (any backend library).
        ``start_tick`` fast-forwards the contact schedule (the countdown lists are advanced accordingly)."""
        self.pd, self.solver = problem_def, solver
        rb = problem_def.robot
        self.model, self.data = rb.model, rb.model.createData()
        self.lf_id, self.rf_id = rb.foot_frame_ids
        N = problem_def.horizon
        self.N = N
        self.phases = problem_def.contact_phases
        ev = refgen.contact_event_times(self.phases, N)
        self.takeoff_RFs, self.takeoff_LFs, self.land_RFs, self.land_LFs = [list(e) for e in ev]
        lf, rf = rb.foot_placements
        self.foottraj = refgen.FootTrajectory(lf.copy(), rf.copy(), _T_SS(problem_def), _T_DS(problem_def), N, swing_apex, x_forward, y_forward,
                                              foot_yaw, y_gap, z_height)
        self.step_params = (x_forward, y_forward, y_gap, z_height, swing_apex)
        self.move_terminal_constraint = move_terminal_constraint
        self.problem = problem_def.build(with_terminal_constraint=True)
        # the horizon initially holds `N` copies of the first stage; fast-forward = the loop below without solving
        self.t = 0
        for _ in range(start_tick):
            refgen.update_timings(self.land_LFs, self.land_RFs, self.takeoff_LFs, self.takeoff_RFs)
            self.problem.replaceStageCircular(problem_def.stage_for_tick(self.t))
            self.t += 1
        solver.setup(self.problem)
        xs, us = problem_def.initial_guess()
        solver.run(self.problem, xs, us)
        solver.max_iters = 1
        self.xs, self.us = list(solver.results.xs), list(solver.results.us)
        self.x_measured = np.array(self.xs[0])
        self.history = []

    def foot_poses(self, x):
        pin.framesForwardKinematics(self.model, self.data, np.asarray(x)[:self.model.nq])
        return self.data.oMf[self.lf_id].copy(), self.data.oMf[self.rf_id].copy()

    def tick(self):
        pd, prob, solver, N = self.pd, self.problem, self.solver, self.N
        LF_pose, RF_pose = self.foot_poses(self.x_measured)
        takeoff_RF, takeoff_LF, land_RF, land_LF = refgen.update_timings(self.land_LFs, self.land_RFs, self.takeoff_LFs, self.takeoff_RFs)
        LF_refs, RF_refs = self.foottraj.updateTrajectory(takeoff_RF, takeoff_LF, land_RF, land_LF, LF_pose, RF_pose)
        for j in range(N):
            prob.stages[j].cost.getComponent(3).residual.setReference(LF_refs[j])
            prob.stages[j].cost.getComponent(4).residual.setReference(RF_refs[j])
        prob.replaceStageCircular(pd.stage_for_tick(self.t % pd.t_mpc))
        solver.workspace.cycleAppend(None)
        if self.move_terminal_constraint:
            com_final = pd.robot.com0.copy()
            com_final[:2] = 0.5 * (LF_refs[-1].translation[:2] + RF_refs[-1].translation[:2])
            prob.removeTerminalConstraint()
            prob.addTerminalConstraint(pd.terminal_com_constraint(com_final))
        prob.term_cost.components[2][0].residual.setReference(LF_refs[-1])
        prob.term_cost.components[3][0].residual.setReference(RF_refs[-1])
        # perfect-model feedback + warm-start shift (fulldynamic_talos.py:532-536)
        self.x_measured = np.array(self.xs[1])
        xs = self.xs[1:] + [self.xs[-1]]
        us = self.us[1:] + [self.us[-1]]
        xs[0] = self.x_measured
        prob.x0_init = self.x_measured
        solver.setup(prob)
        solver.run(prob, xs, us)
        self.xs, self.us = list(solver.results.xs), list(solver.results.us)
        self.t += 1
        rec = {"tick": self.t, "LF_ref": LF_refs[0].translation.copy(), "RF_ref": RF_refs[0].translation.copy(),
               "LF": LF_pose.translation.copy(), "RF": RF_pose.translation.copy(), "takeoff_RF": takeoff_RF, "land_RF": land_RF,
               "takeoff_LF": takeoff_LF, "land_LF": land_LF}
        self.history.append(rec)
        return rec


def _T_SS(pd):
    from . import fulldynamic
    return fulldynamic.T_SS


def _T_DS(pd):
    from . import fulldynamic
    return fulldynamic.T_DS
