"""The kinodynamic control pipeline of kinodynamic_talos.py:361-497 for an ensemble of robots, every stage on the solver library:

    MPC tick (kinodynamic OCP, one ProxDDP iteration)                                   kinodynamic_talos.py:482-490
      -> 10 low-level steps of 1 ms, each:
           measured state of every robot                                                 :412-418
           a0, forces = xdot(knot 0), us[0] corrected by the Riccati feedback K_0        :420-434
           whole-body inverse-dynamics QP (IDSolver_ulim), assembled and solved on the device  (mpc_qp_solve_id)   :438-446
           torque clamped to the effort limits                                           :448-450
           one simulator step under that torque                  (mpc_simulate_torque)   :458 (device.execute)
      -> the measurement of the tick before becomes the initial condition of the next solve   :484-486

The three native pieces are the library's own (include/mpc_abi.h, include/mpc_qp_abi.h); between them travel the small per-robot
vectors (states, K_0, torques), not problem data.  The simulator is the stand-in of ``mpc_simulate_torque``: the whole-body contact
dynamics of the CONTACT STATE OF KNOT 0 of the schedule (what ``problem.stages[0]`` says, :419) — rigid contacts at the measured-at-start
foot placements, no physics engine.  ``library``: the HIP library by default; tests pass the oracle to get the reference run."""
from __future__ import annotations

import numpy as np

from . import _capi as K
from . import qp_utils
from .aligator import _core as core
from .aligator import dynamics as _dyn
from .aligator import manifolds as _manifolds
from .ensemble import EnsembleMPC
from .problems import common
from .robot import minipin as pin


def _sim_options():
    """option block of a simulator handle (never solves: only the library's own consistency checks look at it)"""
    o = K.default_options(1e-5, 1e-8)
    o.force_initial_condition, o.rollout_linear = 1, 1
    return o


class KinodynamicPipeline:
    def __init__(self, problem_def, batch=1, library=None, walk=None, weights_id=(1.0, 10000.0), substeps=10, sim_dt=1e-3, x0=None, **ens_kw):
        """``problem_def``: a KinodynamicProblem.  ``walk``: keyword arguments of ``EnsembleMPC.enable_walk`` ({} = the script's 0.3 m steps)
        or None (references frozen at the initial footholds)."""
        self.pd, self.batch = problem_def, int(batch)
        self.lib = library if library is not None else K.load_hip_library()
        rb = problem_def.robot
        m = self.model = rb.model
        self.nq, self.nv = m.nq, m.nv
        self.substeps, self.sim_dt = int(substeps), float(sim_dt)
        self.mpc = EnsembleMPC(problem_def, batch=batch, library=self.lib, x0=x0, **ens_kw)
        self._walk_args = walk
        # the low-level QP of the script: weights [1, 10000] on acceleration / force increments (kinodynamic_talos.py:350-351)
        self.qp = qp_utils.IDSolver_ulim(m, list(weights_id), 2, common.FRICTION_MU, common.FOOT_HALF_LENGTH, common.FOOT_HALF_WIDTH,
                                         list(rb.foot_frame_ids), 6, library=self.lib, batch=self.batch)
        self.qp.enable_device_assembly()
        self.umax = np.asarray(m.effortLimit, dtype=float)[6:]
        self._build_simulator()
        self.x = np.array(self.mpc.x0, dtype=float)      # measured states, one row per robot
        self.x_prev = self.x.copy()                      # the measurement of the tick before (the solve's initial condition)
        self.torques = np.zeros((self.batch, m.nv - 6))
        self.forces = np.zeros((self.batch, 12))
        self._plan_stale = True

    # -- simulator stand-in: one handle, horizon 1, whole-body contact dynamics of the three contact patterns ----------------------
    def _build_simulator(self):
        m, rb = self.model, self.pd.robot
        nu = m.nv - 6
        space = _manifolds.MultibodyPhaseSpace(m)
        ctx = core.LoweringContext()
        cms = []
        for name, fid, jid, oMf in zip(common.FOOT_FRAMES, rb.foot_frame_ids, rb.foot_joint_ids, rb.foot_placements):
            cm = pin.RigidConstraintModel(pin.ContactType.CONTACT_6D, m, jid, m.frames[fid].placement, 0, oMf, pin.LOCAL)
            cm.corrector.Kp[:] = (0, 0, 10, 0, 0, 0)      # fulldynamic_talos.py:93-94
            cm.corrector.Kd[:] = (50, 50, 50, 50, 50, 50)
            cm.name = name
            cms.append(cm)
        act, prox = np.eye(m.nv, nu, -6), pin.ProximalSettings(1e-9, 1e-10, 1)
        self._sim_tables = {}
        for mask in ((True, True), (True, False), (False, True)):
            ode = _dyn.MultibodyConstraintFwdDynamics(space, act, [c for c, on in zip(cms, mask) if on], prox)
            cost = core.CostStack(space, nu)
            cost.addCost(core.QuadraticControlCost(space, np.zeros(nu), np.eye(nu)))
            st = core.StageModel(cost, _dyn.IntegratorSemiImplEuler(ode, self.sim_dt))
            self._sim_tables[mask] = core.lower_stage(ctx, st.cost, st.dynamics, st.constraints)
        tcost = core.CostStack(space, nu)
        tcost.addCost(core.QuadraticStateCost(space, nu, space.neutral(), np.eye(space.ndx)))
        term = core.lower_stage(ctx, tcost, None, core._ConstraintStack())
        d = K.MpcDims()
        d.horizon, d.batch, d.space = 1, self.batch, K.SPACE_MULTIBODY
        d.nx, d.ndx, d.nu, d.nc_max = space.nx, space.ndx, nu, 1
        d.max_stage_ints = 8 + 8 * 24
        d.max_stage_doubles = max(t[1].size for t in self._sim_tables.values()) + term[1].size + 1024
        d.device = self.mpc.dims.device
        self.sim = K.NativeSolver(self.lib, d)
        self.sim.set_options(_sim_options())
        self.sim.set_model(*ctx.model_tables())
        self.sim.set_stage(1, *term)
        self._sim_mask = None

    def _set_sim_contacts(self, mask):
        mask = (bool(mask[0]), bool(mask[1]))
        if mask != self._sim_mask:
            self.sim.set_stage(0, *self._sim_tables[mask])
            self._sim_mask = mask

    # -- the loop ---------------------------------------------------------------------------------------------------------------------
    def cold_solve(self, max_iters=100):
        st = self.mpc.cold_solve(max_iters=max_iters)
        if self._walk_args is not None:
            self.mpc.enable_walk(**self._walk_args)
        self._fetch()
        return st

    def _fetch(self):
        self._plan_stale = False
        r = self.mpc.native.get_results(gains=False)
        self.xs0, self.us0 = r["xs"][:, 0].copy(), r["us"][:, 0].copy()
        self.K0 = self.mpc.native.get_gain(0)[0]
        self.xdot0 = self.mpc.native.get_stage_data(0)[0]

    def contact_state(self):
        """[left, right] the low-level loop of this MPC period works with: ``problem.stages[0]`` AFTER this period's
        ``replaceStageCircular(stages_full[t])`` (kinodynamic_talos.py:393, 419-420) — the stage that was appended at tick t + 1 - N (the
        initial double support before that), one rotation later than knot 0 of the solution the feedback terms come from."""
        N, t = self.mpc.problem.num_steps, self.mpc.tick
        return self.pd.contact_phases[max(0, t + 1 - N) % self.pd.t_mpc]

    def low_level_step(self, cs):
        """One 1 kHz step of kinodynamic_talos.py:411-462 for every robot, the glue between the library calls on the host."""
        if self._plan_stale:
            self._fetch()
        nq, nv = self.nq, self.nv
        x = self.x
        d = np.concatenate([pin.difference_batch(self.model, x[:, :nq], self.xs0[:, :nq]), self.xs0[:, nq:] - x[:, nq:]], axis=1)  # space.difference(x_measured, xs[0])
        a0 = self.xdot0[:, nv:].copy()
        a0[:, 6:] = self.us0[:, 12:] - np.einsum("bij,bj->bi", self.K0[:, 12:], d)
        forces = self.us0[:, :12] - np.einsum("bij,bj->bi", self.K0[:, :12], d)
        a_new, f_new, tau = self.qp.solve_batch_device(x, a0, forces, np.tile(np.asarray(cs, dtype=np.int32), (self.batch, 1)))
        tau = np.clip(tau, -self.umax, self.umax)
        self.x = self.sim.simulate_torque(x, tau, 1, self.sim_dt)
        self.torques, self.forces = tau, f_new
        return tau

    def low_level_loop(self, cs):
        """The ``substeps`` low-level periods of one MPC period inside the library (mpc_qp_low_level_steps: feedback terms, QP, clamp and simulator step chained
        on the device, one synchronisation).  -> the measured states before the last period (the script reads x_measured BEFORE the last execute of the tick)."""
        cs_all = np.tile(np.asarray(cs, dtype=np.int32), (self.batch, 1))
        x_last, self.x, self.torques, self.forces = self.qp.low_level_steps(self.mpc.native, self.sim, cs_all, self.umax, self.substeps, self.sim_dt, x=self.x)
        return x_last

    def tick(self, host_glue=False):
        """One MPC period: the low-level loop on the current plan, then the next solve from the measurement of the tick before.  ``host_glue``: the low-level
        periods one at a time with the small vectors travelling through the host (``low_level_step``: the readable form, what the library call is tested against)."""
        cs = self.contact_state()
        self._set_sim_contacts(cs)
        if host_glue:
            if self._plan_stale:
                self._fetch()
            for _ in range(self.substeps):
                x_last = self.x.copy()   # (the script's x_measured is read BEFORE the last execute of the tick)
                self.low_level_step(cs)
        else:
            x_last = self.low_level_loop(cs)
        e = self.mpc
        if e._walk is not None:     # the references are planned from the state that becomes the initial condition (walking_loop.py)
            e._walk["x_measured"] = self.x_prev[0].copy()
            if "x_measured_all" in e._walk:
                e._walk["x_measured_all"] = self.x_prev.copy()
        e.native.set_x0(self.x_prev)
        st = e.step()
        self.x_prev = x_last
        self._plan_stale = True   # (knot 0 of the new plan is read on the device; the host copies only when the host glue asks)
        return st
