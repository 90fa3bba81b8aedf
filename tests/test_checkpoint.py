"""Solver-state checkpoint through the C-ABI (mpc_state_size / mpc_get_state / mpc_set_state): a run continued
from a restored state reproduces the uninterrupted run — also when the state is restored into a FRESH handle: bit for bit with the
serial Riccati sweep; with the parallel-in-time sweep to round-off (1e-9), because the value-function guesses the legs keep at
their cuts from pass to pass are an accelerator, not part of the state (the first pass after a restore refreshes them with its extra
sweeps).  A state saved by the HIP library restores into the oracle (same layout): the next ticks agree at 1e-6."""
import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem


def _ens(lib, horizon, batch, legs=1, **kw):
    e = EnsembleMPC(FullDynamicsProblem(horizon=horizon), batch=batch, library=lib, seed=9, sigma_q=0.003, sigma_v=0.006, **kw)
    e.options.riccati_legs = legs
    e.native.set_options(e.options)
    e.prepare_schedule(40)
    return e


def _same(a, b, exact):
    return np.array_equal(a, b) if exact else float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) < 1e-9


def _roundtrip(lib, horizon, batch, legs=1, **kw):
    kw = dict(kw, legs=legs)
    a = _ens(lib, horizon, batch, **kw)
    a.options.tol = 0.0
    a.cold_solve(max_iters=6)
    for _ in range(3):
        a.step()
    state, tick = a.native.get_state(), a.tick
    for _ in range(4):
        a.step()
    ref = a.results(gains=True)
    # the same handle, rolled back
    a.native.set_state(state); a.tick = tick
    for _ in range(4):
        a.step()
    again = a.results(gains=True)
    # a fresh handle that never solved anything
    b = _ens(lib, horizon, batch, **kw)
    b.options.max_iters = 1
    b.options.tol = 0.0
    b.native.set_options(b.options)
    b.native.set_state(state); b.tick = tick
    for _ in range(4):
        b.step()
    fresh = b.results(gains=True)
    for key in ("xs", "us", "K"):
        assert _same(again[key], ref[key], legs == 1), key
        assert _same(fresh[key], ref[key], legs == 1), key
    return state, tick, ref


@pytest.mark.parametrize("legs", [1, 3])
def test_checkpoint_roundtrip_oracle(oracle_lib, legs):
    _roundtrip(oracle_lib, 6, 2, legs=legs)


def test_state_rejects_other_dimensions(oracle_lib):
    a, b = _ens(oracle_lib, 6, 2), _ens(oracle_lib, 5, 2)
    with pytest.raises(RuntimeError, match="other dimensions|truncated"):
        b.native.set_state(a.native.get_state())


@pytest.mark.gpu
@pytest.mark.parametrize("reuse,legs", [(False, 1), (True, 1), (True, 4)])
def test_checkpoint_roundtrip_hip(hip_lib, reuse, legs):
    _roundtrip(hip_lib, 12, 3, legs=legs, tick_reuse=reuse)


@pytest.mark.gpu
def test_hip_state_restores_into_the_oracle(hip_lib, oracle_lib):
    state, tick, ref = _roundtrip(hip_lib, 8, 2)
    o = _ens(oracle_lib, 8, 2)
    o.options.max_iters = 1
    o.native.set_options(o.options)
    o.native.set_state(state); o.tick = tick
    for _ in range(4):
        o.step()
    r = o.results(gains=True)
    for key in ("xs", "us"):
        assert float(np.max(np.abs(r[key] - ref[key])) / max(1.0, np.max(np.abs(ref[key])))) < 1e-6, key


def test_failure_policy_and_revive_on_the_oracle(oracle_lib):
    """mpc_set_failure_policy / mpc_revive_instance (include/mpc_abi.h): reviving instance 1 from instance 0 makes it continue as a copy
    of instance 0 — iterate, multipliers, measured state — while instance 2 is untouched."""
    from mpc_benchmark_amd.ensemble import EnsembleMPC
    from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
    pd = FullDynamicsProblem(horizon=8)
    e = EnsembleMPC(pd, batch=3, library=oracle_lib, seed=4, sigma_q=0.01, sigma_v=0.02)
    e.prepare_schedule(12)
    e.cold_solve(max_iters=8)
    e.enable_failure_isolation()
    for _ in range(2):
        e.step()
    before = e.results(gains=False)
    assert not np.array_equal(before["xs"][0], before["xs"][1])
    e.native.revive_instance(1, 0)
    st = e.step()
    r = e.results(gains=False)
    assert np.array_equal(r["xs"][0], r["xs"][1]) and np.array_equal(r["us"][0], r["us"][1])
    assert not np.array_equal(r["xs"][0], r["xs"][2])
    assert all(s.converged >= 0 for s in st) and e.lost == [] and e.revived == 0
    with pytest.raises(RuntimeError):
        e.native.revive_instance(1, 1)


def _revive_leaves_the_others_alone(lib):
    """mpc_revive_instance(dst >= 2, src) without a following mpc_setup: dst has the iterate AND the multipliers of src, every other
    instance keeps its own (the co-state rows are indexed (b (N + 1) + k) n: a copy with any other stride lands in the neighbours)."""
    from mpc_benchmark_amd.ensemble import EnsembleMPC
    from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
    pd = FullDynamicsProblem(horizon=6)
    e = EnsembleMPC(pd, batch=4, library=lib, seed=5, sigma_q=0.01, sigma_v=0.02)
    e.prepare_schedule(4)
    e.cold_solve(max_iters=3)
    before = e.native.get_results(gains=False, multipliers=True)
    assert not np.array_equal(before["lams"][1], before["lams"][2])
    e.native.revive_instance(2, 1)
    after = e.native.get_results(gains=False, multipliers=True)
    for key in ("xs", "us", "vs", "lams"):
        for b in (0, 1, 3):
            assert np.array_equal(after[key][b], before[key][b]), "%s of instance %d changed" % (key, b)
        assert np.array_equal(after[key][2], before[key][1]), "%s of the revived instance is not the source's" % key


def test_revive_keeps_the_multipliers_of_the_other_instances_oracle(oracle_lib):
    _revive_leaves_the_others_alone(oracle_lib)


@pytest.mark.gpu
def test_revive_keeps_the_multipliers_of_the_other_instances_hip(hip_lib):
    _revive_leaves_the_others_alone(hip_lib)


# ---- restore with the reference generator in the library (mpc_walk_init / mpc_walk_update) -----------------------------------------------------------
def _restore_mid_swing(lib, exact):
    """mpc_set_state re-uploads every stage, which hands all instances the SHARED parameter tables again.  With ``generator="device"`` the next
    mpc_walk_update must therefore rewrite the references of EVERY knot (not only of the appended one), in both libraries: an ensemble restored in the middle
    of a swing — the plan on the device untouched, no mpc_walk_set_state — must carry the tables of the uninterrupted run and continue like it."""
    import copy
    from tests.test_walk_generator import _ens
    e = _ens(lib, FullDynamicsProblem, "device")
    N, B = e.dims.horizon, e.batch
    for _ in range(42):   # through the planning window (the T_ds = 30 ticks before the take-off: ticks 8 .. 37 at this horizon) into the swing
        e.step()
    assert not e._walk["replanning"]
    state, tick, lists = e.native.get_state(), e.tick, copy.deepcopy(e._walk["lists"])
    plan = e.native.walk_get_state()

    def tables():
        return [[e.native.debug_get("inst_params", k, b).copy() for k in range(N + 1)] for b in range(B)]
    for _ in range(4):
        e.step()
        assert not e._walk["replanning"]
    assert np.array_equal(plan, e.native.walk_get_state())   # (nothing was planned in between: the device plan is the one of the checkpoint)
    ref_t, ref = tables(), e.results(gains=True)
    e.native.set_state(state); e.tick = tick; e._walk["lists"] = copy.deepcopy(lists)
    for _ in range(4):
        e.step()
    got_t, got = tables(), e.results(gains=True)
    for b in range(B):
        for k in range(N + 1):
            assert np.array_equal(ref_t[b][k], got_t[b][k]), "instance %d knot %d: table after the restore differs by %.3e" % (b, k, np.max(np.abs(ref_t[b][k] - got_t[b][k])))
    for key in ("xs", "us", "K"):
        assert _same(got[key], ref[key], exact), key


def test_restore_mid_swing_with_the_library_generator_oracle(oracle_lib):
    _restore_mid_swing(oracle_lib, True)


@pytest.mark.gpu
def test_restore_mid_swing_with_the_library_generator_hip(hip_lib):
    _restore_mid_swing(hip_lib, True)


def _host_patch_after_device_tick(lib):
    """The two ways of writing per-instance references may be mixed: after device-generated ticks a host patch of the same offsets must reach the
    library even when its values equal what the HOST last put there (the HIP library compares host patches with a host mirror that the device generator
    does not maintain: it poisons the ranges it writes, so that the comparison can never match)."""
    from tests.test_walk_generator import _ens
    e = _ens(lib, FullDynamicsProblem, "device")
    off, N = int(e._walk["off_lf"]), e.dims.horizon
    mine = np.arange(12, dtype=float) + 0.5
    for _ in range(8):                                                    # (the planning window of the first take-off opens at tick 8 at this horizon)
        e.step()
    e.native.update_instance_params_batch([(1, j, off, mine) for j in range(N)])   # the host writes: tables and mirror hold `mine`
    assert all(np.array_equal(e.native.debug_get("inst_params", j, 1)[off:off + 12], mine) for j in range(N))
    for _ in range(3):                                                    # device-generated ticks inside the planning window: every knot rewritten
        e.step()
        assert e._walk["replanning"]
    assert all(not np.array_equal(e.native.debug_get("inst_params", j, 1)[off:off + 12], mine) for j in range(N))   # the generator's values are in the tables
    e.native.update_instance_params_batch([(1, j, off, mine) for j in range(N)])   # the same values again
    for j in range(N):
        assert np.array_equal(e.native.debug_get("inst_params", j, 1)[off:off + 12], mine), "knot %d: the host patch was dropped" % j


def test_host_patch_after_device_generated_ticks_oracle(oracle_lib):
    _host_patch_after_device_tick(oracle_lib)


@pytest.mark.gpu
def test_host_patch_after_device_generated_ticks_hip(hip_lib):
    _host_patch_after_device_tick(hip_lib)


def _bad_walk_offsets(lib):
    """mpc_walk_init checks the offsets of the reference slots against the size of the parameter tables — the same error in both libraries (they are used
    for 96-byte copies into the instance tables)."""
    from tests.test_walk_generator import _ens
    e = _ens(lib, FullDynamicsProblem, "host")
    e2 = _ens(lib, FullDynamicsProblem, "device")
    cap = int(e2.dims.max_stage_doubles)
    from mpc_benchmark_amd import _capi as K
    for field, val in (("off_lf", cap - 11), ("toff_rf", cap - 3), ("toff_com", cap - 2), ("off_xref_z", cap)):
        cfg = K.MpcWalkConfig()
        cfg.T_ss, cfg.T_ds, cfg.frame_lf, cfg.frame_rf = 80, 30, 0, 1
        cfg.off_lf = cfg.off_rf = cfg.toff_lf = cfg.toff_rf = cfg.toff_com = cfg.off_xref_z = -1
        setattr(cfg, field, val)
        with pytest.raises(RuntimeError, match="out of range"):
            e2.native.walk_init(cfg)


def test_walk_init_rejects_offsets_outside_the_tables_oracle(oracle_lib):
    _bad_walk_offsets(oracle_lib)


@pytest.mark.gpu
def test_walk_init_rejects_offsets_outside_the_tables_hip(hip_lib):
    _bad_walk_offsets(hip_lib)
