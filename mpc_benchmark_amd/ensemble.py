"""Ensemble receding-horizon driver: B independent Talos MPC instances (randomised initial states) advanced
tick by tick on one GPU — the batched form of the loop body of fulldynamic_talos.py:438-550
(replaceStageCircular -> warm-start shift -> setup -> run with max_iters = 1), with "perfect-model" feedback
(the next measured state is the state the previous solution predicted) instead of the PyBullet simulator.
Everything between ticks stays on the device: the stage ring buffer moves by one slot (``mpc_cycle``), the
warm-start shift and the feedback happen in ``mpc_run_shifted``.
"""
from __future__ import annotations

import os

import numpy as np

from . import _capi as K
from .aligator import _core as core


class EnsembleMPC:
    def __init__(self, problem_def, batch=1, library=None, device=0, seed=20250304, perturb=True, sigma_q=0.02, sigma_v=0.05, perturb_dofs=None,
                 closed_loop=None, forward_mode=0, tick_reuse=False, x0=None):
        """``problem_def``: a FullDynamicsProblem / CentroidalProblem-like builder (``build``, ``stage_for_tick``,
        ``make_solver``, ``initial_guess``).  ``forward_mode``: mpc_options.forward_mode (1 for shards that share a GPU).
        ``x0`` (batch, nx): explicit initial states (a shard of ``ensemble_initial_states``) instead of drawing them here."""
        self.pd = problem_def
        self.batch = int(batch)
        self.lib = library if library is not None else K.load_hip_library()
        self.problem = problem_def.build(with_terminal_constraint=True) if hasattr(problem_def, "terminal_com_constraint") else problem_def.build()
        self.ctx = core.LoweringContext()
        N = self.problem.num_steps
        first = self.problem.stages[0]
        space = first.xspace
        tables = [core.lower_stage(self.ctx, st.cost, st.dynamics, st.constraints) for st in self.problem.stages]
        tables.append(core.lower_stage(self.ctx, self.problem.term_cost, None, self.problem.term_constraints))
        # one lowered table per distinct stage of the schedule (contact pattern)
        self._tick_tables = {}
        d = K.MpcDims()
        d.horizon, d.batch = N, self.batch
        d.space = K.SPACE_MULTIBODY if hasattr(space, "model") else K.SPACE_VECTOR
        d.nx, d.ndx, d.nu = space.nx, space.ndx, first.nu
        d.nc_max = max(1, max(int(t[0][6]) for t in tables))  # the library keeps at least one row (an unconstrained problem has nc_max = 0)
        d.max_stage_ints = 8 + 8 * 24
        d.max_stage_doubles = max(t[1].size for t in tables) + 64
        d.device = int(device)
        self.dims = d
        self.native = K.NativeSolver(self.lib, d)
        solver = problem_def.make_solver()
        self.options = solver._options()
        self.options.forward_mode = int(forward_mode)
        self.native.set_options(self.options)
        if tick_reuse:  # the accepted full step's evaluation serves the next tick (bit-identical results, see mpc_abi.h)
            self.native.set_tick_reuse(True)
        if self.ctx.model is not None:
            self.native.set_model(*self.ctx.model_tables())
        for k, (desc, params) in enumerate(tables):
            self.native.set_stage(k, desc, params)
        self.tables = tables
        # randomised initial states: joints ~ N(0, sigma_q^2), joint velocities ~ N(0, sigma_v^2)
        if x0 is not None:
            self.x0 = np.ascontiguousarray(np.asarray(x0, dtype=float).reshape(self.batch, -1))
        elif perturb and hasattr(space, "model"):
            self.x0 = ensemble_initial_states(self.problem.x0_init, space, self.batch, seed, sigma_q, sigma_v, perturb_dofs)
        else:
            self.x0 = np.tile(np.asarray(self.problem.x0_init, dtype=float), (self.batch, 1))
        # closed_loop = (substeps, dt): the measured state of every tick comes from the simulation stand-in (N2: knot 0's dynamics
        # integrated under the feedback law of the low-level loop) instead of the model's own prediction xs[1]
        self.closed_loop = closed_loop
        self.tick = 0
        # ProxDDP iterations per MPC tick: 1 is the reference loop (fulldynamic_talos.py:407) ; 2 keeps randomised ensembles of the synthetic
        # robot stable over the whole schedule
        self.iters_per_tick = 1
        self.inflight = 0   # asynchronous ticks enqueued and not yet collected (step_async / wait)
        self._isolate = None  # enable_failure_isolation(): (auto_revive, source)
        self.lost, self.revived = [], 0
        self._walk = None   # enable_walk(): the reference loop's per-tick problem updates
        # solver.setup(problem) inside the loop: the full-dynamics and centroidal scripts call it every tick, the kinodynamic one does not
        # (fulldynamic_talos.py:539, centroidal_talos.py:461, kinodynamic_talos.py:487 commented out): multipliers and penalty carry over there
        self._setup_each_tick = bool(problem_def.walk_spec().get("setup_each_tick", True)) if hasattr(problem_def, "walk_spec") else True

    # -- stage tables of the schedule ---------------------------------------------------------------
    def _table_for_tick(self, t):
        # stages of one contact pattern share a table unless the problem says its stages differ tick by tick
        # (kinodynamic: a force reference per tick)
        if hasattr(self.pd, "stage_key"):
            key = self.pd.stage_key(t)
        else:
            key = tuple(self.pd.contact_phases[t]) if hasattr(self.pd, "contact_phases") else t
        if key not in self._tick_tables:
            st = self.pd.stage_for_tick(t)
            self._tick_tables[key] = core.lower_stage(self.ctx, st.cost, st.dynamics, st.constraints)
            if self.ctx.changed:
                self.native.set_model(*self.ctx.model_tables())
                self.ctx.changed = False
        return self._tick_tables[key]

    def prepare_schedule(self, n_ticks):
        for t in range(n_ticks):
            self._table_for_tick(t % self.pd.t_mpc)

    # -- solves -------------------------------------------------------------------------------------------
    def cold_solve(self, max_iters=100):
        xs, us = self.pd.initial_guess()
        xs = np.tile(np.array(xs)[None], (self.batch, 1, 1))
        xs[:, 0, :] = self.x0
        us = np.tile(np.array(us)[None], (self.batch, 1, 1))
        self.options.max_iters = int(max_iters)
        self.native.set_options(self.options)
        self.native.set_x0(self.x0)
        self.native.setup()
        stats = self.native.run(xs, us)
        self.options.max_iters = self.iters_per_tick
        self.native.set_options(self.options)
        self.native.set_x0(None)  # perfect-model feedback from here on
        return stats

    def save_episode(self):
        """Remember the current solver state (normally the cold-solved start) as the beginning of an episode: a checkpoint through
        the C-ABI (mpc_get_state: stage tables of the horizon, iterate, multipliers, measured state) plus, in walk mode, the state of
        the reference generator (countdown lists, planned footholds, last measurement)."""
        import copy
        walk = None
        if self._walk is not None:
            walk = copy.deepcopy({k: self._walk[k] for k in ("lists", "traj", "x_measured", "last", "replanning", "batch", "x_measured_all", "last_all") if k in self._walk})
            if self._walk.get("device"):
                walk["device_plan"] = self.native.walk_get_state()
        self._episode = (self.native.get_state(), self.tick, walk, getattr(self, "replanning_ticks", 0))

    def restart_episode(self):
        """Back to the saved start (mpc_set_state).  The synthetic scenario (perfect-model feedback, randomised states) is not meant
        to be replayed far past the first single-support phase; long runs walk it in episodes instead."""
        import copy
        state, tick0, walk, _ = self._episode
        self.native.set_state(state)
        self.tick = tick0
        self.episodes = getattr(self, "episodes", 0) + 1
        if self._walk is not None and walk is not None:
            walk = copy.deepcopy(walk)
            plan = walk.pop("device_plan", None)
            self._walk.update(walk)
            if plan is not None:
                self.native.walk_set_state(plan)
        elif self._walk is not None:
            self.enable_walk(**self._walk_args)  # saved before the walk was enabled: back to the start of the schedule

    def step(self, rescue=False):
        """One MPC tick for every instance of the ensemble (one ProxDDP iteration each).  ``rescue``: an instance whose
        trajectory has diverged (the library reports a failed factorisation) does not abort a long-running ensemble — the
        ensemble is re-solved from its initial states (counted in ``self.rescues``; the time of the re-solve stays inside
        whatever region the caller is timing)."""
        if self.closed_loop:
            self.native.simulate(*self.closed_loop)  # apply us[0] + feedback for one MPC period, measure
        if self._walk is not None:
            self._walk_references()
        desc, params = self._table_for_tick(self.tick % self.pd.t_mpc)
        self.native.cycle(desc, params)
        if self._walk is not None:
            self._walk_terminal()
        if self._setup_each_tick:
            self.native.setup()
        self.tick += 1
        try:
            if self._walk is not None:
                self.native.run_shifted_async()
                stats, xn = self.native.wait_state()
                self._walk["x_measured"] = xn[0].copy()
                self._walk["x_measured_all"] = xn
                return self._handle_lost(stats)
            return self._handle_lost(self.native.run_shifted())
        except RuntimeError as e:
            if not rescue or "factorisation failed" not in str(e):
                raise
            self.rescues = getattr(self, "rescues", 0) + 1
            return self.cold_solve(max_iters=20)

    def step_async(self):
        """Enqueue one tick without waiting (several shards on different streams overlap on the device); ``wait``
        completes the oldest tick in flight.  Two ticks may be in flight: enqueue tick t + 1, then wait for tick t
        (``self.inflight`` counts them)."""
        if getattr(self, "_need_drain", False):
            self._need_drain = False
            while self.inflight:
                self.wait()
        if self.closed_loop:
            self.native.simulate(*self.closed_loop)
        if self._walk is not None:
            self._walk_references()  # the reference's per-tick updates (fulldynamic_talos.py:444-510) before the stage is cycled
        desc, params = self._table_for_tick(self.tick % self.pd.t_mpc)
        self.native.cycle(desc, params)
        if self._walk is not None:
            self._walk_terminal()
        if self._setup_each_tick:
            self.native.setup()
        self.native.run_shifted_async()
        self.tick += 1
        self.inflight += 1

    def wait(self, rescue=False):
        """Complete the tick enqueued by ``step_async`` (``rescue`` as in ``step``)."""
        try:
            self.inflight = max(0, self.inflight - 1)
            if self._walk is not None:
                stats, xn = self.native.wait_state()
                self._walk["x_measured"] = xn[0].copy()  # instance 0's predicted next state: the measurement the generators plan from
                self._walk["x_measured_all"] = xn     # (per-instance references: everybody's)
                return self._handle_lost(stats)
            return self._handle_lost(self.native.wait())
        except RuntimeError as e:
            if not rescue or "factorisation failed" not in str(e):
                raise
            self.rescues = getattr(self, "rescues", 0) + 1
            while self.inflight > 0:  # a younger tick may already be queued behind the failed one: let it drain
                self.inflight -= 1
                try:
                    self.native.wait()
                except RuntimeError:
                    pass
            return self.cold_solve(max_iters=20)

    # -- one instance failing does not stop the ensemble -------------------------------------------------------------------
    def enable_failure_isolation(self, auto_revive=True, source=0):
        """mpc_set_failure_policy(1): an instance whose factorisation fails is reported (``stats.converged = -code``) and sits out the
        following ticks instead of failing the whole tick.  ``auto_revive``: as soon as no tick is in flight it is re-seeded from instance
        ``source`` (the nominal one: iterate, multipliers, measured state — mpc_revive_instance) and takes part again; ``self.lost`` keeps
        (tick, instance, code), ``self.revived`` counts."""
        self.native.set_failure_policy(True)
        self._isolate = (bool(auto_revive), int(source))

    def _handle_lost(self, stats):
        if self._isolate is None:
            return stats
        lost = [b for b, s in enumerate(stats) if s.converged < 0]
        for b in lost:
            if not any(t_b == b and done is None for (_, t_b, _, done) in self._lost_open()):
                self.lost.append([self.tick, b, -int(stats[b].converged), None])
        auto, src = self._isolate
        if auto and lost and self.inflight > 0:
            self._need_drain = True  # step_async completes the ticks in flight before it enqueues the next one: reviving needs an idle handle
        if auto and lost and self.inflight == 0 and src not in lost:
            for b in lost:
                self.native.revive_instance(b, src)
                self.revived += 1
                if self._walk is not None and self._walk.get("device"):  # the generator lives in the library: the plan of the source as well
                    plan = self.native.walk_get_state()
                    plan[b] = plan[src]
                    self.native.walk_set_state(plan)   # (the next update rewrites every knot's references)
                if self._walk is not None and "batch" in self._walk:  # per-instance references: the generator state of the source as well
                    g = self._walk["batch"]
                    for name in ("sL", "fL", "sR", "fR"):
                        R, p = getattr(g, name)
                        R[b], p[b] = R[src], p[src]
                    self._walk["x_measured_all"] = np.array(self._walk["x_measured_all"])
                    self._walk["x_measured_all"][b] = self._walk["x_measured_all"][src]
                for rec in self.lost:
                    if rec[1] == b and rec[3] is None:
                        rec[3] = self.tick
        return stats

    def _lost_open(self):
        return [tuple(r) for r in self.lost if r[3] is None]

    # -- the reference loop's per-tick problem updates on the shared stage tables ---------------------------------
    def enable_walk(self, swing_apex=0.15, x_forward=None, y_forward=0.0, foot_yaw=0.0, y_gap=0.18, z_height=0.0, per_instance=False, generator="host", floor=False):
        """From now on every tick does what the loop bodies of the scripts do to the problem before solving (fulldynamic_talos.py:444-510,
        kinodynamic_talos.py:361-409, centroidal_talos.py:357-384): ``FootTrajectory.updateTrajectory`` from the measured foot poses, the
        references written into every stage of the horizon (``setReference`` on the two foot-placement costs — integer keys 3 / 4 or the
        string keys of the kinodynamic script — or ``contact_poses[i] = ...`` on the three contact maps of a centroidal stage),
        ``replaceStageCircular``, the terminal CoM target between the last foot references (and the terminal foot references of the
        full-dynamics problem).  ``x_forward``: step length (default: the script's — 0 / 0.3 / 0.2 m); ``z_height``: height gained per
        step (0.10: the stairs of BASELINE.json's kinodynamic configuration; the ``z_height`` argument of ``footTrajectory``,
        talos_utils.py:188-192).  The scripts themselves only walk on flat ground: with ``z_height != 0`` the height of the posture
        reference's base (the state cost pins it with a weight of 1e4, kinodynamic_talos.py:74-88) and of the terminal CoM target follow
        the mean height of the two foot references of their knot — without that a robot that climbs 0.6 m in six steps is asked to keep
        its pelvis where it started (this build's definition of the stairs variant; flat walks are untouched).

        The stage tables of an ensemble are shared by its instances, so the references are planned from instance 0's state (the state
        the last COMPLETED tick predicted: with two ticks in flight that is one tick older than the reference script's measurement)
        and every instance tracks them.  The centroidal OCP has no whole-body model: its feet are where their references put them.

        ``per_instance=True`` (whole-body problems): every instance plans from ITS OWN measured foot poses and gets its own references
        (mpc_enable_instance_params: per-instance parameter tables; ``references.FootTrajectoryBatch`` and
        ``minipin.frame_placements_batch`` do the generator's and the forward kinematics' work for all instances in numpy arrays; one
        call carries the patches of every instance, of which only the changed ones travel).

        ``generator="device"`` (with ``per_instance=True``): the generator itself runs in the library (mpc_walk_init / mpc_walk_update,
        include/mpc_abi.h) — forward kinematics of the sole frames at every instance's predicted next state, foothold rules, swing curves and
        the references written into the instance tables by one kernel; per tick the host only advances the four countdowns.

        ``floor=True`` (per-instance references, flat walks): no foothold is planned below the height of the initial footholds — the floor stops a foot.
        An ensemble that feeds the solver's own prediction back has no ground; the full-dynamics script aims the left foot 1 cm below the right one's
        height at every step (fulldynamic_talos.py:449, ``forward_z_left``), which a floor stops and a prediction does not: without this the footholds of
        the benchmark ensemble sink 5 - 6 cm over the schedule's seven swings and the instances lost late in the schedule are lost on those stretched
        legs.  The mirror loops (a simulator with a floor measures their states) do not need it."""
        from . import references as refgen
        from .robot import minipin as pin
        pd, N = self.pd, self.problem.num_steps
        spec = pd.walk_spec()
        if x_forward is None:
            x_forward = spec["x_forward"]
        self._walk_args = dict(swing_apex=swing_apex, x_forward=x_forward, y_forward=y_forward, foot_yaw=foot_yaw, y_gap=y_gap, z_height=z_height, per_instance=per_instance,
                               generator=generator, floor=floor)
        if floor is not False and floor is not None and (not per_instance or z_height != 0.0):
            raise ValueError("floor=True: per-instance references on flat ground")
        if generator not in ("host", "device") or (generator == "device" and not per_instance):
            raise ValueError("generator: 'host', or 'device' together with per_instance=True")
        rb = pd.robot
        ev = refgen.contact_event_times(pd.contact_phases, N)
        lf, rf = rb.foot_placements
        T_SS, T_DS = spec["T_SS"], spec["T_DS"]
        # where the references live in the parameter tables
        slots = []
        st = pd.stage_for_tick(0)
        desc0, _ = core.lower_stage(self.ctx, st.cost, st.dynamics, st.constraints, slots)
        tslots = []
        core.lower_stage(self.ctx, self.problem.term_cost, None, self.problem.term_constraints, tslots)
        w = self._walk = {
            "lists": [list(e) for e in ev],  # takeoff_RFs, takeoff_LFs, land_RFs, land_LFs
            "traj": refgen.FootTrajectory(lf.copy(), rf.copy(), T_SS, T_DS, N, swing_apex, x_forward, y_forward, foot_yaw, y_gap, z_height),
            "data": rb.model.createData(), "pin": pin, "refgen": refgen, "spec": spec, "kind": spec["kind"],
            "step": dict(swing_apex=swing_apex, x_forward=x_forward, y_forward=y_forward, y_gap=y_gap, z_height=z_height),
            "x_measured": np.array(self.x0[0]), "patched": 0, "patches": 0, "last": None, "feet": None,
            "floor_z": (None if floor is False or floor is None else float(min(lf.translation[2], rf.translation[2])) if floor is True else float(floor)),  # (a number: that height — tests)
        }
        nterm = len(self.problem.term_cost.components)
        if spec["kind"] == "pose":
            keys = list(st.cost.components.keys())
            i_lf, i_rf = keys.index(spec["pose_keys"][0]), keys.index(spec["pose_keys"][1])
            w["off_lf"], w["off_rf"] = slots[i_lf][1], slots[i_rf][1]
            assert slots[i_lf][2] == 12 and slots[i_rf][2] == 12
            w["toff_com"] = tslots[nterm][1]
            assert tslots[nterm][2] == 3
            w["off_xref_z"] = slots[keys.index(spec["state_key"])][1] + 2  # base height of the posture reference (stairs)
            w["feet_z0"] = 0.5 * (float(lf.translation[2]) + float(rf.translation[2]))
            if spec["terminal_feet"]:
                w["toff_lf"], w["toff_rf"] = tslots[2][1], tslots[3][1]
                assert tslots[2][2] == 12 and tslots[3][2] == 12
        else:  # "contact_poses": p0 / p1 of the dynamics parameters and of the two acceleration residuals (include/mpc_abi.h)
            keys = list(st.cost.components.keys())
            dyn = int(desc0[4])
            w["pose_offs"] = [[dyn + 5 + 3 * i, slots[keys.index("angular_acc_cost")][1] + 4 + 4 * i + 1,
                               slots[keys.index("linear_acc_cost")][1] + 4 + 4 * i + 1] for i in (0, 1)]
            if per_instance:
                raise NotImplementedError("per-instance references: whole-body problems only")
        if per_instance and generator == "device":
            self.native.enable_instance_params()
            cfg = K.MpcWalkConfig()
            cfg.T_ss, cfg.T_ds = int(T_SS), int(T_DS)
            cfg.frame_lf, cfg.frame_rf = (self.ctx.frame_index(rb.model, f) for f in rb.foot_frame_ids)
            if self.ctx.changed:  # (a frame the stages had not referenced yet)
                self.native.set_model(*self.ctx.model_tables())
                self.ctx.changed = False
            cfg.off_lf, cfg.off_rf, cfg.off_xref_z = int(w["off_lf"]), int(w["off_rf"]), int(w["off_xref_z"])
            cfg.toff_com = int(w["toff_com"])
            cfg.toff_lf, cfg.toff_rf = (int(w["toff_lf"]), int(w["toff_rf"])) if spec["terminal_feet"] else (-1, -1)
            cfg.swing_apex = float(swing_apex)
            gen = w["traj"]
            cfg.t_left[:], cfg.t_right[:] = list(gen.translationLeft), list(gen.translationRight)
            cfg.rot_diff[:] = list(np.asarray(gen.rotationDiff, dtype=float).reshape(-1))
            cfg.com0[:] = list(np.asarray(rb.com0, dtype=float))
            cfg.feet_z0, cfg.xref_z0, cfg.z_follow = float(w["feet_z0"]), float(self.pd.x0[2]), (1.0 if z_height != 0.0 else 0.0)
            flat = lambda M: list(np.concatenate([np.asarray(M.rotation, dtype=float).reshape(-1), np.asarray(M.translation, dtype=float)]))
            cfg.lf0[:], cfg.rf0[:] = flat(lf), flat(rf)
            cfg.floor_z = w["floor_z"] if w["floor_z"] is not None else -1e308
            self.native.walk_init(cfg)
            w["device"] = True
        elif per_instance:
            B = self.batch
            self.native.enable_instance_params()
            bc = lambda M: (np.tile(np.asarray(M.rotation, dtype=float), (B, 1, 1)), np.tile(np.asarray(M.translation, dtype=float), (B, 1)))
            (LR, Lp), (RR, Rp) = bc(lf), bc(rf)
            w["batch"] = refgen.FootTrajectoryBatch(LR, Lp, RR, Rp, T_SS, T_DS, N, swing_apex, x_forward, y_forward, foot_yaw, y_gap, z_height)
            w["batch"].floor_z = w["floor_z"]
            w["x_measured_all"] = np.array(self.x0, dtype=float)
            i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
            # index arrays of the patches of one tick: per instance the N left-foot and N right-foot references, then (after the cycle) its terminal targets
            w["idx"] = (i32(np.repeat(np.arange(B), 2 * N)), i32(np.tile(np.concatenate([np.arange(N), np.arange(N)]), B)),
                        i32(np.tile(np.concatenate([np.full(N, w["off_lf"]), np.full(N, w["off_rf"])]), B)), i32(np.full(B * 2 * N, 12)))
            w["zidx"] = (i32(np.repeat(np.arange(B), N)), i32(np.tile(np.arange(N), B)), i32(np.full(B * N, w["off_xref_z"])), i32(np.full(B * N, 1)))
            if spec["terminal_feet"]:
                w["tidx"] = (i32(np.repeat(np.arange(B), 3)), i32(np.full(3 * B, N)), i32(np.tile([w["toff_com"], w["toff_lf"], w["toff_rf"]], B)), i32(np.tile([3, 12, 12], B)))
            else:
                w["tidx"] = (i32(np.arange(B)), i32(np.full(B, N)), i32(np.full(B, w["toff_com"])), i32(np.full(B, 3)))

    def _walk_forward_rule(self, gen, takeoff_RF, takeoff_LF, land_RF, land_LF):
        """the ``foottraj.updateForward`` call the scripts make once the walk is over (fulldynamic_talos.py:448-449,
        kinodynamic_talos.py:368-370, centroidal_talos.py:365-366)"""
        rule, p = self._walk["spec"]["forward_rule"], self._walk["step"]
        if rule(takeoff_RF, takeoff_LF, land_RF, land_LF):
            gen.updateForward(0, 0, p["y_gap"], p["y_forward"], self._walk["spec"]["forward_z_left"], 0, p["swing_apex"])

    def _walk_references(self):
        w, N = self._walk, self.problem.num_steps
        rb, pin, refgen = self.pd.robot, self._walk["pin"], self._walk["refgen"]
        takeoff_RFs, takeoff_LFs, land_RFs, land_LFs = w["lists"]
        if w.get("device"):  # the generator runs in the library: only the countdowns (and the scripts' updateForward rule) are host work
            takeoff_RF, takeoff_LF, land_RF, land_LF = refgen.update_timings(land_LFs, land_RFs, takeoff_LFs, takeoff_RFs)
            forward = None
            if w["spec"]["forward_rule"](takeoff_RF, takeoff_LF, land_RF, land_LF):
                p = w["step"]
                forward = ([0.0, p["y_gap"], w["spec"]["forward_z_left"]], [0.0, -p["y_gap"] - p["y_forward"], 0.0], p["swing_apex"])
            self.native.walk_update(takeoff_RF, takeoff_LF, land_RF, land_LF, forward)
            T_ds = w["traj"].T_ds
            w["replanning"] = (land_LF < 0 or land_RF < 0 or 0 <= takeoff_RF < T_ds or 0 <= takeoff_LF < T_ds)
            self.replanning_ticks = getattr(self, "replanning_ticks", 0) + int(w["replanning"])
            return
        if "batch" in w:  # per-instance references
            (LR, Lp), (RR, Rp) = pin.frame_placements_batch(rb.model, w["x_measured_all"][:, :rb.model.nq], rb.foot_frame_ids)
            takeoff_RF, takeoff_LF, land_RF, land_LF = refgen.update_timings(land_LFs, land_RFs, takeoff_LFs, takeoff_RFs)
            self._walk_forward_rule(w["batch"], takeoff_RF, takeoff_LF, land_RF, land_LF)
            Lb, Rb = w["batch"].updateTrajectory(takeoff_RF, takeoff_LF, land_RF, land_LF, LR, Lp, RR, Rp)
            vals = np.ascontiguousarray(np.concatenate([Lb, Rb], axis=1)).reshape(-1)
            self.native.update_instance_params_arrays(*w["idx"], vals)
            if w["step"]["z_height"] != 0.0:  # stairs: the posture reference climbs with the feet
                zref = self.pd.x0[2] + 0.5 * (Lb[:, :, 11] + Rb[:, :, 11]) - w["feet_z0"]
                self.native.update_instance_params_arrays(*w["zidx"], np.ascontiguousarray(zref).reshape(-1))
            w["last_all"] = (Lb[:, -1].copy(), Rb[:, -1].copy())
            w["replanning"] = (land_LF < 0 or land_RF < 0 or 0 <= takeoff_RF < w["batch"].T_ds or 0 <= takeoff_LF < w["batch"].T_ds)
            self.replanning_ticks = getattr(self, "replanning_ticks", 0) + int(w["replanning"])
            return
        if w["kind"] == "contact_poses":  # no whole-body state: the feet stand where the previous plan put them for this tick
            if w["feet"] is None:
                lf, rf = rb.foot_placements
                w["feet"] = (lf.copy(), rf.copy())
            LF_pose, RF_pose = w["feet"][0].copy(), w["feet"][1].copy()
        else:
            pin.framesForwardKinematics(rb.model, w["data"], np.asarray(w["x_measured"])[:rb.model.nq])
            LF_pose, RF_pose = w["data"].oMf[rb.foot_frame_ids[0]].copy(), w["data"].oMf[rb.foot_frame_ids[1]].copy()
        takeoff_RF, takeoff_LF, land_RF, land_LF = refgen.update_timings(land_LFs, land_RFs, takeoff_LFs, takeoff_RFs)
        self._walk_forward_rule(w["traj"], takeoff_RF, takeoff_LF, land_RF, land_LF)
        LF_refs, RF_refs = w["traj"].updateTrajectory(takeoff_RF, takeoff_LF, land_RF, land_LF, LF_pose, RF_pose)
        batch = []
        if w["kind"] == "contact_poses":
            w["feet"] = (LF_refs[1], RF_refs[1])
            # knot j holds the stage of schedule index j - N + tick (the first N - tick knots: the initial double-support stage), and only
            # the feet that stand in that stage get their pose (centroidal_talos.py:374-384: written BEFORE the new stage is rotated in)
            for j in range(N):
                cs = self.pd.contact_phases[max(0, j - N + self.tick) % self.pd.t_mpc]
                for i, refs in ((0, LF_refs), (1, RF_refs)):
                    if cs[i]:
                        for off in w["pose_offs"][i]:
                            batch.append((j, off, np.asarray(refs[j].translation, dtype=float)))
        else:
            flat = lambda M: np.concatenate([np.asarray(M.rotation, dtype=float).reshape(-1), np.asarray(M.translation, dtype=float)])
            for j in range(N):
                batch.append((j, w["off_lf"], flat(LF_refs[j])))
                batch.append((j, w["off_rf"], flat(RF_refs[j])))
            if w["step"]["z_height"] != 0.0:  # stairs: the posture reference climbs with the feet
                for j in range(N):
                    dz = 0.5 * (LF_refs[j].translation[2] + RF_refs[j].translation[2]) - w["feet_z0"]
                    batch.append((j, w["off_xref_z"], np.array([self.pd.x0[2] + dz])))
        self.native.update_stage_params_batch(batch)
        w["last"] = (LF_refs[-1], RF_refs[-1])
        # ticks on which the generator plans from the measured poses (a foot without a pending landing, a take-off inside the double-
        # support window): their references change for every knot, so no record of the previous tick can be reused
        w["replanning"] = (land_LF < 0 or land_RF < 0 or 0 <= takeoff_RF < w["traj"].T_ds or 0 <= takeoff_LF < w["traj"].T_ds)
        self.replanning_ticks = getattr(self, "replanning_ticks", 0) + int(w["replanning"])

    def _walk_terminal(self):
        w, N = self._walk, self.problem.num_steps
        if w["kind"] == "contact_poses":
            return  # the centroidal problem has no terminal target (centroidal_talos.py:249)
        feet = w["spec"]["terminal_feet"]
        if w.get("device"):
            return  # (written by the generator kernel together with the references)
        if "batch" in w:
            L_last, R_last = w["last_all"]
            com = np.tile(self.pd.robot.com0, (self.batch, 1))
            com[:, :2] = 0.5 * (L_last[:, 9:11] + R_last[:, 9:11])
            if w["step"]["z_height"] != 0.0:
                com[:, 2] += 0.5 * (L_last[:, 11] + R_last[:, 11]) - w["feet_z0"]
            vals = np.ascontiguousarray(np.concatenate([com, L_last, R_last] if feet else [com], axis=1)).reshape(-1)
            self.native.update_instance_params_arrays(*w["tidx"], vals)
            return
        LF_last, RF_last = w["last"]
        flat = lambda M: np.concatenate([np.asarray(M.rotation, dtype=float).reshape(-1), np.asarray(M.translation, dtype=float)])
        com_final = self.pd.robot.com0.copy()
        com_final[:2] = 0.5 * (LF_last.translation[:2] + RF_last.translation[:2])
        if w["step"]["z_height"] != 0.0:
            com_final[2] += 0.5 * (LF_last.translation[2] + RF_last.translation[2]) - w["feet_z0"]
        patches = [(N, w["toff_com"], com_final)]
        if feet:
            patches += [(N, w["toff_lf"], flat(LF_last)), (N, w["toff_rf"], flat(RF_last))]
        self.native.update_stage_params_batch(patches)

    def results(self, **kw):
        return self.native.get_results(**kw)


def ensemble_initial_states(x0, space, total, seed=20250304, sigma_q=0.02, sigma_v=0.05, perturb_dofs=None):
    """Initial states of an ensemble of ``total`` instances: ONE ``default_rng(seed)`` stream drawn in
    instance order — joints ~ N(0, sigma_q^2), joint velocities ~ N(0, sigma_v^2) around the nominal state, instance 0 unperturbed.
    The ensemble does not depend on how it is sharded: rank r of G takes the rows ``shard_instances(total, r, G)``."""
    rng = np.random.default_rng(seed)
    x0 = np.asarray(x0, dtype=float)
    out = np.tile(x0, (int(total), 1))
    if not hasattr(space, "model"):  # vector-space problems (centroidal): identical instances
        return out
    mdl = space.model
    nv = mdl.nv
    for b in range(int(total)):
        dq = np.zeros(2 * nv)
        dq[6:nv] = sigma_q * rng.standard_normal(nv - 6)
        dq[nv + 6:] = sigma_v * rng.standard_normal(nv - 6)
        if perturb_dofs is not None:  # e.g. upper body only: feet that must stay at rest are not disturbed
            keep = np.zeros(nv, dtype=bool)
            keep[np.asarray(perturb_dofs, dtype=int)] = True
            dq[:nv][~keep] = 0.0
            dq[nv:][~keep] = 0.0
        if b > 0:
            xb = space.integrate(x0, dq)
            # keep the measured configuration inside the joint limits: the initial state is fixed
            # (force_initial_condition), so a violated limit at knot 0 is an infeasible constraint that no
            # iteration can repair (two joints of the nominal posture sit exactly on a limit)
            if hasattr(mdl, "lowerPositionLimit"):
                xb[7:mdl.nq] = np.clip(xb[7:mdl.nq], mdl.lowerPositionLimit[7:], mdl.upperPositionLimit[7:])
            out[b] = xb
    return out


def shard_instances(total, rank, world):
    """Instance i of the ensemble lives on GPU i mod G."""
    return np.arange(int(rank), int(total), int(world))


def make_bench_shards(problem_def, library, batch_per_gpu, rank=0, world=1, streams=1, device=0, seed=20250304, legs=4, tick_reuse=True,
                      closed_loop=None):
    """The ensemble(s) bench.py drives on one GPU: this rank's ``batch_per_gpu`` instances of the ``batch_per_gpu * world`` ensemble,
    split into ``streams`` handles (1: one lock-step ensemble).  Also what tests/test_gpu_bench_config.py checks against the oracle."""
    prob = problem_def.build(with_terminal_constraint=True) if hasattr(problem_def, "terminal_com_constraint") else problem_def.build()
    total = int(batch_per_gpu) * int(world)
    x0s = ensemble_initial_states(prob.x0_init, prob.stages[0].xspace, total, seed)
    mine = shard_instances(total, rank, world)
    nshard = max(1, min(int(streams), int(batch_per_gpu)))
    parts = np.array_split(mine, nshard)
    shards = []
    for part in parts:
        e = EnsembleMPC(problem_def, batch=len(part), library=library, device=device, x0=x0s[part], closed_loop=closed_loop,
                        forward_mode=(1 if nshard > 1 else 0), tick_reuse=tick_reuse)
        e.instance_ids = part
        if legs > 0:
            e.options.riccati_legs = int(legs)
            e.native.set_options(e.options)
        shards.append(e)
    return shards


def result_blocks(shards):
    """(instance ids, one row per instance: xs | us | K_0 flattened) of the shards of this rank."""
    rows, ids = [], []
    for e in shards:
        r = e.results(gains=True)
        B = e.batch
        rows.append(np.concatenate([r["xs"].reshape(B, -1), r["us"].reshape(B, -1), r["K"][:, 0].reshape(B, -1)], axis=1))
        ids.append(np.asarray(getattr(e, "instance_ids", np.arange(B))))
    return np.concatenate(ids), np.concatenate(rows)


def allgather_results(shards, dist=None, device=None):
    """Round-end exchange of SURVEY.md §8e: every rank contributes the result blocks (xs, us, K_0) of its instances and receives
    the whole ensemble, ordered by instance id — ``torch.distributed.all_gather`` (RCCL over xGMI on the GPUs, gloo in the CPU
    tests).  Not part of a solve: the data path of the MPC ticks has no collective.  Returns (ids, blocks)."""
    ids, blk = result_blocks(shards)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        order = np.argsort(ids, kind="stable")
        return ids[order], blk[order]
    import torch
    t = torch.from_numpy(np.ascontiguousarray(np.concatenate([ids[:, None].astype(np.float64), blk], axis=1)))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    full = torch.cat(out).cpu().numpy()
    order = np.argsort(full[:, 0], kind="stable")
    return full[order, 0].astype(np.int64), full[order, 1:]


def lq_knot_doubles(n, m, c):
    """W_k of SURVEY.md §8d: Q,A (2 n^2), R (m^2), S,B (2 n m), q,f (2 n), r (m), [C D] (c (n + m)), d (c)."""
    return 2 * n * n + m * m + 2 * n * m + 2 * n + m + c * (n + m) + c


def gain_doubles(n, m, c):
    """G_k of SURVEY.md §8d: K,k ; dual gains ; co-state gains ; P,p."""
    return m * n + m + c * (n + 1) + n * (n + 1) + n * n + n
