"""Shards of one GPU: several handles on their own streams, ticks enqueued without waiting (``step_async`` / ``wait``),
stages cycled without re-upload when unchanged, both forward sweeps.  Everything must give the numbers of the plain
synchronous single-handle path (same kernels, different scheduling) and match the oracle."""
import numpy as np
import pytest

from tests._metrics import traj_err

from tests import _oracle
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def _run(lib, mode, ticks=8, forward_mode=0):
    pd = FullDynamicsProblem(horizon=12)
    # two shards with different seeds; "sync" drives them one after the other, "async" interleaves them
    shards = [EnsembleMPC(pd, batch=3, library=lib, seed=11 + i, sigma_q=0.005, sigma_v=0.01, forward_mode=forward_mode) for i in range(2)]
    for e in shards:
        e.prepare_schedule(ticks + 2)
        e.cold_solve(max_iters=40)
    if mode == "sync":
        for _ in range(ticks):
            for e in shards:
                e.step()
    else:
        for e in shards:
            e.step_async()
        for _ in range(ticks - 1):
            for e in shards:
                e.step_async()  # two ticks in flight
                e.wait()
        for e in shards:
            e.wait()
    res = [e.results(gains=False) for e in shards]
    return np.concatenate([r["xs"].reshape(-1, r["xs"].shape[-1]) for r in res]), np.concatenate([r["us"].reshape(-1, r["us"].shape[-1]) for r in res])


def test_async_shards_equal_sync_and_oracle():
    hip = _capi.load_hip_library()
    sync = _run(hip, "sync")
    asyn = _run(hip, "async")
    assert np.array_equal(sync[0], asyn[0]) and np.array_equal(sync[1], asyn[1])  # same kernels on the same data: bit-identical
    ref = _run(_oracle.load(), "sync")
    assert traj_err(sync[0], sync[1], ref[0], ref[1]) < 1e-6  # component by component (tests/_metrics.py)


def test_forward_modes_agree():
    hip = _capi.load_hip_library()
    sweep = _run(hip, "sync", ticks=4, forward_mode=1)
    phi = _run(hip, "sync", ticks=4, forward_mode=2)
    assert traj_err(sweep[0], sweep[1], phi[0], phi[1]) < 1e-9


def test_profile_mask_times_only_selected_kernels():
    hip = _capi.load_hip_library()
    pd = FullDynamicsProblem(horizon=8)
    ens = EnsembleMPC(pd, batch=2, library=hip, seed=3, sigma_q=0.005, sigma_v=0.01)
    ens.prepare_schedule(6)
    ens.cold_solve(max_iters=30)
    ens.native.profile(2)
    ens.native.profile(1)
    ens.step()
    ens.native.profile(0)
    allk = ens.native.profile_read(slots=True)
    assert "k_riccati_backward" in allk and "k_eval_stage" in allk
    slot = allk["k_riccati_backward"][2]
    ens.native.profile(2)
    ens.native.profile(16 * (1 << slot))
    ens.step()
    ens.step()
    ens.native.profile(0)
    only = ens.native.profile_read()
    assert list(only) == ["k_riccati_backward"] and only["k_riccati_backward"][0] == 2


def test_episode_restart_replays_the_walk():
    """``restart_episode``: stage ring, initial states and iterate go back to the cold-solved start; the ticks that follow
    repeat the first episode up to the extra warm iteration taken at the restart (the cold solve stops at its tolerance, so one more
    iteration still moves the iterate a little)."""
    hip = _capi.load_hip_library()
    pd = FullDynamicsProblem(horizon=12)
    ens = EnsembleMPC(pd, batch=3, library=hip, seed=21, sigma_q=0.005, sigma_v=0.01)
    ens.prepare_schedule(40)
    ens.cold_solve(max_iters=60)
    ens.save_episode()
    first = []
    for _ in range(5):
        ens.step()
        first.append(ens.results(gains=False)["xs"].copy())
    for _ in range(20):  # far enough for the contact pattern at the head of the ring to have changed
        ens.step()
    ens.restart_episode()
    assert ens.tick == 0 and ens.episodes == 1
    for t in range(5):
        ens.step()
        assert _rel(ens.results(gains=False)["xs"], first[t]) < 1e-2


@pytest.mark.parametrize("horizon,ticks", [(12, 30), (40, 12)])
def test_tick_reuse_is_bit_identical(horizon, ticks):
    """mpc_set_tick_reuse: the accepted full step's evaluation (with derivatives) becomes the next tick's knot records.  Same
    arithmetic on the same points: trajectories, controls and gains equal the plain path bit for bit — across contact switches
    of the schedule, ticks with backtracking (the kept records are dropped then) and the asynchronous two-deep drive."""
    hip = _capi.load_hip_library()
    out = {}
    for reuse in (False, True):
        pd = FullDynamicsProblem(horizon=horizon)
        ens = EnsembleMPC(pd, batch=4, library=hip, seed=31, tick_reuse=reuse)
        ens.prepare_schedule(ticks + 2)
        ens.cold_solve(max_iters=60)
        hist, alphas = [], []
        ens.step_async()
        for _ in range(ticks - 1):
            ens.step_async()
            st = ens.wait()
            alphas.append([s.alpha for s in st])
        ens.wait()
        r = ens.results(gains=True)
        out[reuse] = (r["xs"].copy(), r["us"].copy(), r["K"].copy(), np.array(alphas))
    assert np.array_equal(out[False][3], out[True][3])
    for a, b in zip(out[False][:3], out[True][:3]):
        assert np.array_equal(a, b)


def test_tick_reuse_with_irregular_cycling_is_bit_identical():
    """The speculative evaluation of the appended knot assumes one mpc_cycle per tick with the table of the then last stage: two cycles
    before a run, a run without a cycle and a changed table must all fall back to the plain evaluation (same results bit for bit)."""
    hip = _capi.load_hip_library()
    out = {}
    for reuse in (False, True):
        pd = FullDynamicsProblem(horizon=12)
        ens = EnsembleMPC(pd, batch=3, library=hip, seed=17, tick_reuse=reuse)
        ens.prepare_schedule(40)
        ens.cold_solve(max_iters=60)
        for i in range(18):
            desc, params = ens._table_for_tick(ens.tick % pd.t_mpc)
            if i % 5 != 4:                       # (every fifth tick: no cycle at all)
                ens.native.cycle(desc, params)
            if i % 3 == 0:                       # (every third: a second cycle, with the table of a later tick)
                d2, p2 = ens._table_for_tick((ens.tick + 7) % pd.t_mpc)
                ens.native.cycle(d2, p2)
            ens.native.setup()
            ens.tick += 1
            ens.native.run_shifted()
        r = ens.results(gains=True)
        out[reuse] = (r["xs"].copy(), r["us"].copy(), r["K"].copy())
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a, b)


def test_tick_reuse_is_inert_on_vector_space_problems():
    """The centroidal problem keeps its knot records in knot order (no ring): mpc_set_tick_reuse is accepted and changes
    nothing, tick after tick."""
    from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
    hip = _capi.load_hip_library()
    out = {}
    for reuse in (False, True):
        pd = CentroidalProblem(horizon=20)
        ens = EnsembleMPC(pd, batch=2, library=hip, seed=5, perturb=False, tick_reuse=reuse)
        ens.prepare_schedule(10)
        ens.cold_solve(max_iters=40)
        for _ in range(8):
            ens.step()
        r = ens.results(gains=True)
        out[reuse] = (r["xs"].copy(), r["us"].copy(), r["K"].copy())
    for a, b in zip(out[False], out[True]):
        assert np.array_equal(a, b)
