"""The fixtures of tools/check_dropin.py: what the reference's OWN scripts (fulldynamic_talos.py, kinodynamic_talos.py,
centroidal_talos.py, run unmodified in the build container against this repo's ``aligator`` mirror, with stand-ins for the third-party
packages only) built and solved — tests/golden/dropin_<script>.npz.

  * CPU, everywhere: ``mpc_benchmark_amd/problems/*.py`` (this repo's restatement of the scripts' problem construction) must lower to the
    SAME tables as the scripts' own code, bit for bit: every node of the problem, every stage of the schedule (``stages_full``), the
    schedule itself, x0, the solver attributes, the robot-model tables.
  * CPU (oracle) and ``-m gpu`` (HIP): the first 20 MPC ticks of every script replayed through
    ``mpc_benchmark_amd/problems/walking_loop.py`` (this repo's restatement of the loop bodies) from the recorded measurements: the tables
    uploaded for every solve must equal the recorded ones bit for bit, the trajectories / K_0 the recorded ones (the oracle's) within
    1e-6 per component (the oracle itself: 1e-9).
"""
import hashlib
import os

import numpy as np
import pytest

from mpc_benchmark_amd.aligator import _core as core
from mpc_benchmark_amd.problems import walking_loop
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
from tests._metrics import rel_cols

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PROBLEMS = {"fulldynamic": FullDynamicsProblem, "kinodynamic": KinodynamicProblem, "centroidal": CentroidalProblem}
# how each script starts: x_forward of its foot-trajectory generator, terminal constraint present at the cold solve
LOOP_ARGS = {"fulldynamic": dict(x_forward=0.0, terminal_constraint_at_start=False), "kinodynamic": dict(x_forward=0.3), "centroidal": dict(x_forward=0.2)}


def digest(desc, params):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(desc, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(params, dtype=np.float64).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8).copy()


def fixture(name):
    path = os.path.join(GOLDEN, "dropin_%s.npz" % name)
    if not os.path.exists(path):
        pytest.skip("no %s (tools/check_dropin.py writes it in the build container)" % path)
    return np.load(path)


def build_problem(name, pd):
    if name == "fulldynamic":
        return pd.build(with_terminal_constraint=False)  # the script adds its terminal constraint in the loop (fulldynamic_talos.py:372)
    return pd.build()


def assert_tables_equal(got, fx, prefix, k, what):
    d, p = got
    assert np.array_equal(d, fx["%s_desc_%d" % (prefix, k)]), "%s: descriptor differs" % what
    ref = fx["%s_params_%d" % (prefix, k)]
    assert p.shape == ref.shape and np.array_equal(p, ref), "%s: parameters differ (max %.3e)" % (what, np.max(np.abs(p - ref)) if p.shape == ref.shape else np.nan)


@pytest.mark.parametrize("name", ["fulldynamic", "kinodynamic", "centroidal"])
def test_problem_construction_equals_the_reference_scripts(name):
    fx = fixture(name)
    pd = PROBLEMS[name]()
    prob = build_problem(name, pd)
    N = int(fx["horizon"])
    assert prob.num_steps == N and pd.t_mpc == int(fx["schedule_len"])
    assert np.array_equal(np.asarray(prob.x0_init), fx["x0"])
    assert np.array_equal(np.array(pd.contact_phases, dtype=np.int8), fx["contact_phases"])
    solver = pd.make_solver(_native_library=object())
    attrs = np.array([solver.target_tol, solver.mu_init, float(solver.num_threads), float(solver.rollout_type), float(solver.linear_solver_choice),
                      float(solver.force_initial_condition)])
    assert np.array_equal(attrs, fx["solver_attrs"])
    # the problem as built: every node
    ctx = core.LoweringContext()
    tabs = [core.lower_stage(ctx, st.cost, st.dynamics, st.constraints) for st in prob.stages]
    tabs.append(core.lower_stage(ctx, prob.term_cost, None, prob.term_constraints))
    for k in sorted({0, N // 2, N - 1, N}):
        assert_tables_equal(tabs[k], fx, "problem", k, "%s node %d" % (name, k))
    got = np.stack([digest(d, p) for d, p in tabs])
    assert np.array_equal(got, fx["problem_digests"]), "nodes %s differ" % np.flatnonzero(np.any(got != fx["problem_digests"], axis=1))
    if "model_itab" in fx.files:
        it, dt = ctx.model_tables()
        assert np.array_equal(it, fx["model_itab"]) and np.array_equal(dt, fx["model_dtab"])
    # every stage of the schedule
    ctx2 = core.LoweringContext()
    full = []
    for t in range(pd.t_mpc):
        st = pd.stage_for_tick(t)
        full.append(core.lower_stage(ctx2, st.cost, st.dynamics, st.constraints))
    for t in fx["stages_full_sample"]:
        assert_tables_equal(full[int(t)], fx, "stages_full", int(t), "%s stages_full[%d]" % (name, t))
    got = np.stack([digest(d, p) for d, p in full])
    bad = np.flatnonzero(np.any(got != fx["stages_full_digests"], axis=1))
    assert bad.size == 0, "stages_full %s differ" % bad[:10]


def replay(name, library, tol, n_ticks=None):
    fx = fixture(name)
    pd = PROBLEMS[name]()
    solver = pd.make_solver(_native_library=library)
    loop = walking_loop.make_loop(pd, solver, **LOOP_ARGS[name])
    N = int(fx["horizon"])
    floor = 1e-3
    errs = {"cold": max(rel_cols(np.array(loop.xs), fx["cold_xs"], floor), rel_cols(np.array(loop.us), fx["cold_us"], floor))}
    assert loop.cold["num_iters"] == int(fx["cold_iters"]), "cold solve: %d iterations, the recorded run took %d" % (loop.cold["num_iters"], int(fx["cold_iters"]))
    assert errs["cold"] < tol, "cold solve deviates by %.3e" % errs["cold"]
    T = int(fx["n_ticks"]) if n_ticks is None else min(n_ticks, int(fx["n_ticks"]))
    prev_xs, prev_us = fx["cold_xs"], fx["cold_us"]
    worst = 0.0
    for t in range(T):
        loop.set_solution(prev_xs, prev_us)
        x_fk = pd.robot.x0 if t == 0 else fx["tick_x_measured"][t - 1]
        loop.tick(x_fk=x_fk, x0_init=fx["tick_x0_init"][t])
        got = np.stack([digest(*solver._node(loop.problem, k)._lowered) for k in range(N + 1)])
        bad = np.flatnonzero(np.any(got != fx["tick_digests"][t], axis=1))
        if bad.size:
            k = int(bad[0])
            msg = "%s tick %d: the tables of nodes %s differ from what the script uploaded" % (name, t, bad[:8])
            key = "tick%d_params_%d" % (t, k)
            if key in fx.files:
                p = solver._node(loop.problem, k)._lowered[1]
                msg += " (node %d: max |dp| %.3e at %s)" % (k, np.max(np.abs(p - fx[key])), np.flatnonzero(p != fx[key])[:8])
            raise AssertionError(msg)
        r = solver.results
        e = max(rel_cols(np.array(r.xs), fx["tick_xs"][t], floor), rel_cols(np.array(r.us), fx["tick_us"][t], floor),
                rel_cols(np.array(r.controlFeedbacks()[0]), fx["tick_K0"][t], floor))
        assert r.num_iters == int(fx["tick_iters"][t])
        assert e < tol, "%s tick %d: xs / us / K_0 deviate from the recorded run by %.3e" % (name, t, e)
        worst = max(worst, e)
        prev_xs, prev_us = fx["tick_xs"][t], fx["tick_us"][t]
    return worst


@pytest.mark.parametrize("name", ["fulldynamic", "kinodynamic", "centroidal"])
def test_loop_replay_on_the_oracle(name):
    """The restated loop bodies reproduce the script's uploads bit for bit; the oracle reproduces its own recorded trajectories."""
    from tests import _oracle
    replay(name, _oracle.load(), 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fulldynamic", "kinodynamic", "centroidal"])
def test_hip_reproduces_the_recorded_script_runs(name):
    """The HIP library driven through the restated loop from the recorded measurements: cold solve + 20 MPC ticks within 1e-6 per
    component of what the scripts got from the oracle."""
    from mpc_benchmark_amd import _capi
    worst = replay(name, _capi.load_hip_library(), 1e-6)
    print("%s: worst deviation over the recorded ticks %.3e" % (name, worst))
