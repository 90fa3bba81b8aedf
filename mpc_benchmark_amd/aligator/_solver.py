"""``aligator.SolverProxDDP`` mirror (fulldynamic_talos.py:374-397, 539-550): lowers the problem, keeps the
device copy in sync with in-place mutations of the Python objects, and calls the native solver through the
C-ABI.  The default backend is the HIP library; it is loaded on first use and there is no CPU fallback."""
from __future__ import annotations

import numpy as np

from .. import _capi as K
from . import _core as core
from . import manifolds as _manifolds

ROLLOUT_NONLINEAR = 0
ROLLOUT_LINEAR = 1
LQ_SOLVER_SERIAL = 0
LQ_SOLVER_PARALLEL = 1
LQ_SOLVER_STAGEDENSE = 2


class VerboseLevel:
    QUIET = 0
    VERBOSE = 1
    VERYVERBOSE = 2


class _ArrayList(list):
    """``results.xs`` / ``results.us``: a sequence of 1-D arrays with the ``.tolist()`` the scripts call."""

    def tolist(self):
        return list(self)


class _LazyGains:
    """``results.controlFeedbacks()`` / ``controlFeedforwards()``: element 0 — the only one the scripts read (fulldynamic_talos.py:522,
    :550) — is fetched with the results; the gains of the other knots (1.9 MB per instance on the complete model) are downloaded when
    something else is touched.  They are those of the solver's LAST run, like every other field of ``results``."""

    def __init__(self, first, fetch_all):
        self._first, self._fetch, self._all = first, fetch_all, None

    def _full(self):
        if self._all is None:
            self._all = self._fetch()
        return self._all

    def __getitem__(self, i):
        if isinstance(i, int) and i == 0:
            return self._first
        return self._full()[i]

    def __len__(self):
        return len(self._full())

    def __iter__(self):
        return iter(self._full())

    def tolist(self):
        return list(self._full())


class Results:
    def __init__(self):
        self.xs = _ArrayList()
        self.us = _ArrayList()
        self.vs = _ArrayList()
        self.lams = _ArrayList()
        self._K = []
        self._kff = []
        self.num_iters = 0
        self.conv = False
        self.traj_cost = 0.0
        self.merit_value = 0.0
        self.prim_infeas = 0.0
        self.dual_infeas = 0.0
        self.al_iter = 0

    def controlFeedbacks(self):
        return self._K

    def controlFeedforwards(self):
        return self._kff

    def __str__(self):
        return ("Results {\n  num_iters:    %d,\n  converged:    %s,\n  traj. cost:   %.6e,\n  merit.value:  %.6e,\n"
                "  prim_infeas:  %.6e,\n  dual_infeas:  %.6e,\n}" % (self.num_iters, self.conv, self.traj_cost,
                                                                  self.merit_value, self.prim_infeas, self.dual_infeas))


class _ContactForce:
    def __init__(self, w):
        self.linear = np.array(w[:3])
        self.angular = np.array(w[3:])


class _ConstraintData:
    def __init__(self, w):
        self.contact_force = _ContactForce(w)


class _ContinuousData:
    def __init__(self, xdot, wrenches):
        self.xdot = xdot
        self.constraint_datas = [_ConstraintData(w) for w in wrenches]


class _DynamicsData:
    def __init__(self, cont):
        self.continuous_data = cont


class _StageDataView:
    def __init__(self, cont):
        self.dynamics_data = _DynamicsData(cont)


class _StageDataSeq:
    """``workspace.problem_data.stage_data[k]`` fetched lazily from the native workspace."""

    def __init__(self, solver):
        self._solver = solver

    def __len__(self):
        return self._solver._dims.horizon

    def __getitem__(self, k):
        s = self._solver
        xdot, wr = s._native.get_stage_data(int(k))
        stage = s._problem.stages[int(k)]
        ode = stage.dynamics.differential_dynamics
        if isinstance(ode, core.MultibodyConstraintFwdDynamics):
            ids = [s._ctx.contact_index(cm) for cm in ode.constraint_models]
            wrenches = [wr[0, i] for i in ids]
        else:
            wrenches = []
        return _StageDataView(_ContinuousData(xdot[0].copy(), wrenches))


class _ProblemData:
    def __init__(self, solver):
        self.stage_data = _StageDataSeq(solver)


class Workspace:
    def __init__(self, solver):
        self.problem_data = _ProblemData(solver)

    def cycleAppend(self, stage_data):
        """Data rotation is implicit in the native ring buffer (fulldynamic_talos.py:497)."""


import os as _os

# mpc_options.corrector_prim_tol of a freshly constructed solver.  0 = off: a run takes exactly max_iters iterations, as the reference's.
# The robust setting for ensembles of perturbed robots driven on an iteration budget of one is 20.0 (N, N m, rad: a contact switch in the
# appended stage injects 100 - 300, an ordinary tick < 10) together with refine_appended_knot = 3 — what bench.py runs.
# (MPC_DEFAULT_CORRECTOR: developer override.)
DEFAULT_CORRECTOR_PRIM_TOL = float(_os.environ.get("MPC_DEFAULT_CORRECTOR", "0.0"))
ROBUST_CORRECTOR_PRIM_TOL = 20.0


class SolverProxDDP:
    def __init__(self, tol=1e-6, mu_init=1e-2, max_iters=1000, verbose=VerboseLevel.QUIET, _native_library=None):
        self.target_tol = float(tol)
        self.mu_init = float(mu_init)
        self.max_iters = int(max_iters)
        self.verbose = verbose
        self.rollout_type = ROLLOUT_NONLINEAR
        self.linear_solver_choice = LQ_SOLVER_SERIAL
        self.force_initial_condition = False
        self.reg_init = 1e-9
        self.num_threads = 1
        self.riccati_legs = None  # None: chosen by _legs()
        self.refine_appended_knot = 0  # mpc_options.refine_appended_knot (this build's extension: 0 = the scripts' plain warm-start shift)
        # mpc_options.corrector_prim_tol / corrector_window (this build's globalisation of an iteration budget of one, include/mpc_abi.h): a run
        # whose last iteration started from an iterate infeasible by more than this, or whose step was shortened, takes one more iteration.
        # Off by default (the reference's loops, nominal robot, closed loop, run to their last line without it: tools/check_dropin.py);
        # ROBUST_CORRECTOR_PRIM_TOL is what keeps ensembles of perturbed robots walking.
        self.corrector_prim_tol = DEFAULT_CORRECTOR_PRIM_TOL
        self.corrector_window = 0  # every run (ensembles driven asynchronously use a window: EnsembleMPC)
        self.batch = 1
        self.results = Results()
        self.workspace = None
        # dependency injection for tests (the CPU oracle); product code never passes this
        self._lib = _native_library
        self._native = None
        self._problem = None
        self._dims = None
        self._ctx = None
        self._uploaded = None
        self._opts_raw = None         # option block as last sent to the library
        self._cycles_since_run = 0    # stages rotated in (replaceStageCircular) since the last run
        self._last_results = None

    def setNumThreads(self, n):
        self.num_threads = int(n)

    # -- option block ---------------------------------------------------------------------------
    def _options(self):
        o = K.default_options(self.target_tol, self.mu_init)
        o.reg_init = self.reg_init
        o.max_iters = self.max_iters
        o.force_initial_condition = 1 if self.force_initial_condition else 0
        o.rollout_linear = 1 if self.rollout_type == ROLLOUT_LINEAR else 0
        o.num_threads = self.num_threads
        o.riccati_legs = self._legs() if self.linear_solver_choice == LQ_SOLVER_PARALLEL else 1
        o.refine_appended_knot = int(self.refine_appended_knot)
        o.corrector_prim_tol = float(self.corrector_prim_tol)
        o.corrector_window = int(self.corrector_window)
        return o

    def _legs(self):
        """LQ_SOLVER_PARALLEL + setNumThreads(n) (fulldynamic_talos.py:383-385) asks for a parallel-in-time sweep; how many legs is the
        backend's business.  The CPU libraries take one leg per thread.  On the GPU the number of host threads means nothing: the
        sweep of one instance is fastest in 32 legs, an ensemble that fills the device by itself in fewer —
        256 / batch, at least 4.  ``solver.riccati_legs = n`` overrides.  Same KKT system either way: results equal up to round-off."""
        if self.riccati_legs is not None:
            return int(self.riccati_legs)
        if self._native is not None and self._native.backend.startswith("hip"):
            return max(4, min(32, 256 // max(1, int(self.batch))))
        return self.num_threads

    # -- lowering / device sync -----------------------------------------------------------------
    def _lower_node(self, node, cost, dynamics, constraints):
        if node._dirty or node._lowered is None:
            slots = []
            node._lowered = core.lower_stage(self._ctx, cost, dynamics, constraints, slots)
            node._dirty = False
            node._patches = []
            for res, off, size in slots:
                if hasattr(res, "_ref"):
                    res._slot = (node, off, size)  # setReference writes here from now on (see _Owned._reference_changed)
            return True
        return False

    def _lower_all(self, problem):
        changed = []
        for k, st in enumerate(problem.stages):
            if self._lower_node(st, st.cost, st.dynamics, st.constraints):
                changed.append(k)
        t = problem._term
        if self._lower_node(t, problem.term_cost, None, problem.term_constraints):
            changed.append(len(problem.stages))
        return changed

    def _node(self, problem, k):
        return problem.stages[k] if k < len(problem.stages) else problem._term

    def setup(self, problem):
        if self.rollout_type != ROLLOUT_LINEAR or not self.force_initial_condition:
            raise NotImplementedError("only rollout_type=ROLLOUT_LINEAR with force_initial_condition=True is "
                                      "implemented (the configuration of fulldynamic_talos.py:381-384)")
        if self._lib is None:
            self._lib = K.load_hip_library()
        N = problem.num_steps
        first = problem.stages[0]
        space = first.xspace
        if self._problem is not problem or self._native is None:
            self._ctx = core.LoweringContext()
            for st in problem.stages:
                st._dirty = True
            problem._term._dirty = True
            problem._cycled = []
            self._lower_all(problem)
            nodes = [self._node(problem, k) for k in range(N + 1)]
            nc_max = max(int(nd._lowered[0][6]) for nd in nodes)
            d = K.MpcDims()
            d.horizon, d.batch = N, int(self.batch)
            d.space = K.SPACE_MULTIBODY if isinstance(space, _manifolds.MultibodyPhaseSpace) else K.SPACE_VECTOR
            d.nx, d.ndx, d.nu = space.nx, space.ndx, first.nu
            # every schedule of the reference starts in double support, which has the most constraint rows
            # (fulldynamic_talos.py:371, kinodynamic_talos.py:274); a later stage with more rows is rejected by the library
            d.nc_max = max(nc_max, 1)
            d.max_stage_ints = 8 + 8 * 24
            d.max_stage_doubles = max(nd._lowered[1].size for nd in nodes) + 1024
            d.device = 0
            self._dims = d
            self._native = K.NativeSolver(self._lib, d)
            self._reuse_on = False
            self._problem = problem
            self._opts_raw = None
            self._last_results = None
            self._uploaded = [None] * (N + 1)
            self.workspace = Workspace(self)
            self._model_uploaded = False
        self._send_options()
        self._sync(problem)
        self._native.setup()

    def _send_options(self):
        o = self._options()
        raw = bytes(o)
        if raw != self._opts_raw:  # (setup and run of every tick would otherwise each cost a call and, in the library, a reuse reset)
            self._native.set_options(o)
            self._opts_raw = raw

    def _sync(self, problem):
        """Bring the device copy of the stage tables up to date with the Python objects."""
        N = problem.num_steps
        nat = self._native
        # 1. stages rotated in by replaceStageCircular
        for st in problem._cycled:
            self._lower_node(st, st.cost, st.dynamics, st.constraints)
        if self._ctx.model is not None and (self._ctx.changed or not self._model_uploaded):
            nat.set_model(*self._ctx.model_tables())
            self._ctx.changed = False
            self._model_uploaded = True
        for st in problem._cycled:
            desc, params = st._lowered
            nat.cycle(desc, params)
            self._cycles_since_run += 1
            self._uploaded = self._uploaded[1:N] + [(desc, params), self._uploaded[N]]
        problem._cycled = []
        # 2. dirty nodes: re-lower; upload the parameter table, or the whole stage if its structure changed
        self._lower_all(problem)
        if self._ctx.changed:
            nat.set_model(*self._ctx.model_tables())
            self._ctx.changed = False
        batch = []
        for k in range(N + 1):
            node = self._node(problem, k)
            desc, params = node._lowered
            up = self._uploaded[k]
            if up is not None and up[0] is desc and up[1] is params:
                # unchanged structure: only the reference slots patched in place since the last upload travel
                for off, size in node._patches:
                    batch.append((k, off, params[off:off + size]))
                node._patches = []
                continue
            node._patches = []
            if up is not None and np.array_equal(up[0], desc) and up[1].size == params.size:
                nat.update_stage_params(k, 0, params)
            else:
                nat.set_stage(k, desc, params)
            self._uploaded[k] = (desc, params)
        nat.update_stage_params_batch(batch)

    def cycleProblem(self, problem, stage_data=None):
        """kinodynamic_talos.py:488 / centroidal_talos.py:460 — the rotation itself was recorded by
        ``problem.replaceStageCircular``; nothing else to do until the next run."""

    # -- solve ----------------------------------------------------------------------------------
    def run(self, problem, xs_init=None, us_init=None):
        if self._native is None or self._problem is not problem:
            raise RuntimeError("call solver.setup(problem) before solver.run")
        d = self._dims
        nat = self._native
        self._send_options()
        self._sync(problem)
        xs = np.asarray(xs_init, dtype=float).reshape(d.horizon + 1, d.nx)
        us = np.asarray(us_init, dtype=float).reshape(d.horizon, d.nu)
        x0 = np.asarray(problem.x0_init, dtype=float).reshape(-1)
        cycles, self._cycles_since_run = self._cycles_since_run, 0
        prev = self._last_results
        # The MPC loops pass the previous solution shifted by one knot, after one replaceStageCircular (fulldynamic_talos.py:532-540).
        # The device still holds that solution: it shifts it itself (nothing to upload) and, on multibody problems, reuses the
        # evaluation its last accepted full step left behind (tick reuse: bit-identical to evaluating afresh, include/mpc_abi.h).
        # (knot 0: the solver overwrites xs[0] with x0_init when force_initial_condition is set — the only mode setup accepts — so the
        # caller's xs[0] matters only without it; prev holds PRIVATE copies: in-place edits of results.xs / results.us by the caller
        # are seen as what they are, a warm start that is no longer the pure shift)
        shifted = (prev is not None and cycles == 1 and self.max_iters <= 4 and d.batch == 1
                   and (self.force_initial_condition or np.array_equal(xs[0], prev["xs"][0][1]))
                   and np.array_equal(xs[1:-1], prev["xs"][0][2:]) and np.array_equal(xs[-1], prev["xs"][0][-1])
                   and np.array_equal(us[:-1], prev["us"][0][1:]) and np.array_equal(us[-1], prev["us"][0][-1]))
        if shifted:
            if not self._reuse_on and d.space == K.SPACE_MULTIBODY:
                # from the first shifted run on (a handle that only ever gets plain runs keeps its records as the evaluation of the
                # iterate, which is what the phase dumps of the parity tests read)
                nat.set_tick_reuse(True)
                self._reuse_on = True
            predicted = np.array_equal(x0, prev["xs"][0][1])  # perfect-model feedback: knot 0 is reused as well
            nat.set_x0(None if predicted else x0)
            stats = nat.run_shifted()
        else:
            nat.set_x0(x0)
            stats = nat.run(xs, us)
        self._fetch(stats)
        return bool(self.results.conv)

    def _fetch(self, stats):
        out = self._native.get_results(gains=False, multipliers=False)
        K0, k0 = self._native.get_gain(0)
        r = self.results
        r.xs = _ArrayList(out["xs"][0])
        r.us = _ArrayList(out["us"][0])
        nat = self._native
        r._K = _LazyGains(K0[0], lambda: list(nat.get_results(gains=True)["K"][0]))
        r._kff = _LazyGains(k0[0], lambda: list(nat.get_results(gains=True)["kff"][0]))
        s = stats[0]
        r.num_iters, r.conv, r.al_iter = s.num_iters, bool(s.converged), s.al_iters
        r.traj_cost, r.merit_value, r.prim_infeas, r.dual_infeas = s.traj_cost, s.merit, s.prim_infeas, s.dual_infeas
        self._last_stats = stats
        self._last_results = {"xs": out["xs"].copy(), "us": out["us"].copy()}  # private: results.xs / results.us are the caller's to edit
