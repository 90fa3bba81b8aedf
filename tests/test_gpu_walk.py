"""GPU parity of the walk mode of the ensemble driver (``EnsembleMPC.enable_walk``): every tick regenerates the foot references from
the measured state and patches them into the stage tables (FootTrajectory.updateTrajectory + 2 N setReference + terminal CoM
rebuild — the loop body of fulldynamic_talos.py:444-510), which is what ``bench.py --walk`` times.

* mpc_update_stage_params_batch compares with the host mirror and invalidates tick reuse PER KNOT: the result must be bit-identical
  to the plain path (tick reuse off), through replanning ticks (every knot's reference changes) and ticks with frozen swing
  references (nothing changes: every record is reused);
* the same loop on the CPU oracle: trajectories within 1e-6."""
import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC, make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from tests._metrics import rel_cols

pytestmark = pytest.mark.gpu


def _pipelined(e, count):
    for _ in range(count):
        e.step_async()
        if e.inflight == 2:
            e.wait()
    while e.inflight:
        e.wait()


def test_walk_with_tick_reuse_is_bit_identical_to_the_plain_path(hip_lib):
    """bench.py --walk's configuration at a horizon the test suite can afford twice: 16 instances, N = 40 (complete model, 4 legs,
    two ticks in flight).  45 ticks: on some the generator replans from the measured poses (every knot's reference changes), on the
    others the references of the previous tick are handed over unchanged."""
    res, replanning = {}, {}
    for reuse in (True, False):
        pd = FullDynamicsProblem(horizon=40, complete_model=True)
        (e,) = make_bench_shards(pd, hip_lib, 16, legs=4, tick_reuse=reuse)
        e.prepare_schedule(50)
        e.cold_solve(max_iters=100)
        e.enable_walk()
        _pipelined(e, 45)
        res[reuse] = e.results(gains=True)
        replanning[reuse] = e.replanning_ticks
    assert replanning[True] == replanning[False] and 1 <= replanning[True] < 45, replanning  # both kinds of ticks occur
    for key in ("xs", "us", "K"):
        assert np.array_equal(res[True][key], res[False][key]), key


def test_two_iterations_per_tick_with_tick_reuse_are_bit_identical_to_the_plain_path(hip_lib):
    """``iters_per_tick = 2`` (the robust setting of the ensemble driver): the second pass of a tick takes the records the first pass's
    accepted full-step candidate wrote for the same knots, the first pass of the next tick those of the second — same bits as evaluating
    everything every pass.  Two ticks in flight, walk mode (replanning and frozen ticks)."""
    res = {}
    for reuse in (True, False):
        pd = FullDynamicsProblem(horizon=40, complete_model=True)
        (e,) = make_bench_shards(pd, hip_lib, 8, legs=4, tick_reuse=reuse)
        e.iters_per_tick = 2
        e.prepare_schedule(40)
        e.cold_solve(max_iters=100)
        e.enable_walk()
        _pipelined(e, 30)
        res[reuse] = (e.results(gains=True), [int(s.num_iters) for s in e.native.wait()])
    assert res[True][1] == res[False][1] == [2] * 8
    for key in ("xs", "us", "K"):
        assert np.array_equal(res[True][0][key], res[False][0][key]), key


def test_two_iterations_per_tick_match_oracle(hip_lib, oracle_lib):
    traj = {}
    for name, lib in (("hip", hip_lib), ("ref", oracle_lib)):
        pd = FullDynamicsProblem(horizon=12)
        e = EnsembleMPC(pd, batch=2, library=lib, seed=5, sigma_q=0.005, sigma_v=0.01, tick_reuse=(name == "hip"))
        e.options.tol = 0.0
        e.iters_per_tick = 2
        e.prepare_schedule(40)
        e.cold_solve(max_iters=10)
        hist = []
        for _ in range(25):
            st = e.step()
            assert all(s.num_iters == 2 for s in st)
            r = e.results(gains=False)
            hist.append(np.concatenate([r["xs"].reshape(2, -1), r["us"].reshape(2, -1)], axis=1))
        traj[name] = np.array(hist)
    err = rel_cols(traj["hip"].reshape(-1, traj["hip"].shape[-1]), traj["ref"].reshape(-1, traj["ref"].shape[-1]), 1e-3)
    assert err < 1e-6, err


def test_walk_matches_oracle(hip_lib, oracle_lib):
    traj = {}
    for name, lib in (("hip", hip_lib), ("ref", oracle_lib)):
        pd = FullDynamicsProblem(horizon=12)
        e = EnsembleMPC(pd, batch=2, library=lib, seed=5, sigma_q=0.005, sigma_v=0.01, tick_reuse=(name == "hip"))
        e.options.tol = 0.0  # a fixed number of cold-solve iterations in both libraries (no convergence exit): same starting iterate
        e.prepare_schedule(50)
        e.cold_solve(max_iters=10)
        e.enable_walk()
        hist = []
        for _ in range(40):
            e.step()
            r = e.results(gains=False)
            hist.append(np.concatenate([r["xs"].reshape(2, -1), r["us"].reshape(2, -1)], axis=1))
        traj[name] = np.array(hist)
    err = rel_cols(traj["hip"].reshape(-1, traj["hip"].shape[-1]), traj["ref"].reshape(-1, traj["ref"].shape[-1]), 1e-3)
    assert err < 1e-6, err
