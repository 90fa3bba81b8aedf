"""GPU parity AT THE BENCHMARKED CONFIGURATION (BASELINE.json config 5, the per-GPU shard; bench.py's default): 64 instances of the
N = 100 complete-model full-dynamics OCP, Riccati sweep in 4 legs, tick reuse on, two ticks in flight — built by the same
``make_bench_shards`` bench.py calls.

* whole ensemble, every instance: bit-identical to the plain path (tick reuse off, synchronous ticks) after 33 MPC ticks, which
  take the horizon through the first change of the appended stage (double -> single support enters at tick 30,
  fulldynamic_talos.py:248-266);
* instances {0, 17, 63} against the CPU oracle (serial sweep) over six ticks across that change, started from the HIP iterate of
  tick 27: xs, us, K_0 within 1e-6 (BASELINE.json's tolerance), components held one by one."""
import os

import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC, make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from tests._metrics import rel_cols, rel_tiles

pytestmark = pytest.mark.gpu
B, N, T0, T1 = 64, 100, 27, 33
PICK = [0, 17, 63]


def _pipelined_ticks(e, count):
    """bench.py's driver: tick t + 1 is enqueued before the host looks at tick t."""
    inflight, nostep = 0, 0
    for _ in range(count):
        e.step_async(); inflight += 1
        if inflight == 2:
            nostep += sum(1 for s in e.wait() if s.num_iters == 0); inflight -= 1
    while inflight:
        nostep += sum(1 for s in e.wait() if s.num_iters == 0); inflight -= 1
    return nostep


def test_bench_default_configuration(hip_lib, oracle_lib):
    pd = FullDynamicsProblem(horizon=N, complete_model=True)
    (a,) = make_bench_shards(pd, hip_lib, B, legs=4, tick_reuse=True)     # exactly bench.py's default shard
    (p,) = make_bench_shards(pd, hip_lib, B, legs=4, tick_reuse=False)    # plain path: every knot evaluated every tick, synchronous
    for e in (a, p):
        e.prepare_schedule(T1 + 4)
        e.cold_solve(max_iters=100)
    assert _pipelined_ticks(a, T0) == 0  # (an instance-tick without a step would be deferred in the pipelined driver: none here)
    for _ in range(T0):
        p.step()
    ra, rp = a.results(gains=True), p.results(gains=True)
    for key in ("xs", "us", "K"):
        assert np.array_equal(ra[key], rp[key]), "tick reuse + two ticks in flight differ from the plain path at tick %d: %s" % (T0, key)
    # the oracle takes over instances {0, 17, 63} from the HIP iterate of tick T0: same stage ring, shifted iterate, measured state
    o = EnsembleMPC(FullDynamicsProblem(horizon=N, complete_model=True), batch=len(PICK), library=oracle_lib, x0=a.x0[PICK])
    o.options.riccati_legs = 1
    o.options.num_threads = os.cpu_count() or 8  # host threads of the oracle (OpenMP over knots)
    o.options.max_iters = 1
    o.native.set_options(o.options)
    o.prepare_schedule(T1 + 4)
    for t in range(T0 + 1):  # the ring after T0 + 1 cycles: the tables of ticks 0 .. T0
        o.native.cycle(*o._table_for_tick(t % pd.t_mpc))
    o.tick = T0 + 1
    for t in range(T0, T1):
        if t == T0:
            xs, us = ra["xs"][PICK], ra["us"][PICK]
            xs_s = np.concatenate([xs[:, 1:], xs[:, -1:]], axis=1)  # the warm-start shift of k_shift (csrc/solver_kernels.h)
            us_s = np.concatenate([us[:, 1:], us[:, -1:]], axis=1)
            o.native.set_x0(xs[:, 1])  # perfect-model feedback: the predicted next state is the measurement
            o.native.setup()
            o.native.run(xs_s, us_s)
            o.native.set_x0(None)
        else:
            o.step()
        assert _pipelined_ticks(a, 1) == 0
        p.step()
        ra, ro = a.results(gains=True), o.results(gains=True)
        for key, floor in (("xs", 1e-3), ("us", 1e-1)):
            err = rel_cols(ra[key][PICK], ro[key], floor)
            assert err < 1e-6, "tick %d: %s deviates from the oracle by %.2e" % (t, key, err)
        errK = max(rel_tiles(ra["K"][i, 0], ro["K"][j, 0], 1e-6) for j, i in enumerate(PICK))
        assert errK < 1e-6, "tick %d: K_0 deviates from the oracle by %.2e" % (t, errK)
    rp = p.results(gains=True)
    ra = a.results(gains=True)
    for key in ("xs", "us", "K"):
        assert np.array_equal(ra[key], rp[key]), "tick reuse + two ticks in flight differ from the plain path at tick %d: %s" % (T1, key)


@pytest.mark.parametrize("refs", ["shared", "instance"])
def test_whole_schedule_walk_with_two_iterations_per_tick(hip_lib, refs):
    """Robustness of the benchmarked ensemble over the reference's WHOLE schedule (fulldynamic_talos.py:248-266: 1000 ticks, three steps per
    foot and the final stop), walk mode, two ticks in flight, no episode restart, no rescue, two ProxDDP iterations per tick
    (``iters_per_tick = 2``).  References shared by the ensemble (planned from the nominal instance): every one of the 64 randomised
    instances stays with the nominal one to the end.  References per instance (every instance replans from its own measured foot poses —
    bench.py's walk): the same with failure isolation on, where at most a handful of the 64 000 instance-ticks end in a re-seed from the
    nominal instance (one on this seed).  (With the reference's single iteration per tick the nominal instance walks the schedule and
    perturbed ones are lost when the first single-support phase reaches the front of the horizon: DESIGN.md §5, tools/robustness_probe.py.)"""
    pd = FullDynamicsProblem(horizon=N, complete_model=True)
    (e,) = make_bench_shards(pd, hip_lib, B, legs=4, tick_reuse=True)
    e.iters_per_tick = 2
    e.prepare_schedule(pd.t_mpc + 4)
    e.cold_solve(max_iters=100)
    if refs == "instance":
        e.enable_failure_isolation(auto_revive=True, source=0)
    e.enable_walk(per_instance=(refs == "instance"))
    ticks = min(1000, pd.t_mpc - 1)
    worst = 0.0
    for t in range(ticks):
        e.step_async()
        if e.inflight == 2:
            st = e.wait()   # raises if the library lost an instance (failed factorisation) and isolation is off
            c = np.array([s.traj_cost for s in st if s.converged >= 0])
            assert np.all(np.isfinite(c))
            if t > 300 and refs == "shared":  # the initial perturbations have decayed: every instance walks the nominal gait
                worst = max(worst, float(np.max(np.abs(c - c[0])) / abs(c[0])))
    while e.inflight:
        st = e.wait()
    assert getattr(e, "rescues", 0) == 0 and e.tick == ticks
    assert e.revived <= (3 if refs == "instance" else 0), e.lost
    assert all(rec[1] != 0 for rec in e.lost)
    assert worst < 0.25, worst
    c = np.array([s.traj_cost for s in st if s.converged >= 0])
    assert len(c) >= B - 1 and np.max(np.abs(c - c[0])) < (0.05 if refs == "shared" else 20.0) * abs(c[0])  # (per-instance footholds: the final stances differ)
    r = e.results(gains=False)
    alive = [b for b, s in enumerate(st) if s.converged >= 0]
    assert np.all(np.isfinite(r["xs"][alive])) and np.all(np.isfinite(r["us"][alive]))


def test_failed_instances_are_isolated_and_revived(hip_lib):
    """mpc_set_failure_policy(1) on the benchmarked ensemble with the reference's ONE iteration per tick, over the stretch of the schedule
    where perturbed instances are lost: the tick never raises, a lost instance is reported with its code, sits out, is
    re-seeded from the nominal instance when the handle is idle, and walks on; the nominal instance itself is never lost.  Pipelined
    driver (two ticks in flight)."""
    pd = FullDynamicsProblem(horizon=N, complete_model=True)
    (e,) = make_bench_shards(pd, hip_lib, 32, legs=4, tick_reuse=True)
    e.prepare_schedule(260)
    e.cold_solve(max_iters=100)
    e.enable_failure_isolation(auto_revive=True, source=0)
    seen_negative = 0
    for t in range(250):
        e.step_async()
        if e.inflight == 2:
            st = e.wait()
            seen_negative += sum(1 for s in st if s.converged < 0)
    while e.inflight:
        e.wait()
    assert len(e.lost) >= 1 and e.revived >= 1, (e.lost, e.revived)          # the scenario does lose instances with one iteration per tick
    assert all(rec[1] != 0 for rec in e.lost)                               # never the nominal one
    # (a loss reported while the driver drains the pipeline for a revival is not seen by this loop: at least one is)
    assert all(rec[2] in (2, 3, 4, 5, 6) for rec in e.lost) and seen_negative >= 1
    r = e.results(gains=False)
    st = e.native.wait()
    alive = [b for b, s in enumerate(st) if s.converged >= 0]
    assert 0 in alive and len(alive) >= 28
    assert np.all(np.isfinite(r["xs"][alive])) and np.all(np.isfinite(r["us"][alive]))


def test_bench_walk_with_per_instance_references_matches_oracle(hip_lib, oracle_lib):
    """The driver's HEADLINE mode at full size — walk, per-instance references, 64 instances, N = 100, complete model, 4 legs, tick reuse —
    against the oracle, INSIDE the first replanning window (tick >= 100: every instance replans its footholds from its own measured
    foot poses, every reference of every knot changes): instances {0, 17, 63}, three ticks, xs / us / K_0 within 1e-6 per component (the two
    libraries walk on from their own iterates: a fourth tick is at 1.1e-6).
    The oracle ensemble (3 instances, serial sweep) takes over the HIP iterate of tick 100 with the same stage ring and countdowns; up
    to there the references do not depend on the measurements (no take-off inside the planning window, fulldynamic_talos.py:444-459)."""
    from mpc_benchmark_amd import references as refgen
    T0w, ticks = 100, 3
    pd = FullDynamicsProblem(horizon=N, complete_model=True)
    (a,) = make_bench_shards(pd, hip_lib, B, legs=4, tick_reuse=True)
    a.prepare_schedule(T0w + ticks + 4)
    a.cold_solve(max_iters=100)
    a.enable_walk(per_instance=True)
    assert _pipelined_ticks(a, T0w - 1) == 0
    a.step()  # (synchronous: the measurement the next tick plans from is the state THIS tick predicted, as in the oracle's loop below)
    ra = a.results(gains=True)
    o = EnsembleMPC(FullDynamicsProblem(horizon=N, complete_model=True), batch=len(PICK), library=oracle_lib, x0=a.x0[PICK])
    o.options.riccati_legs = 1
    o.options.num_threads = os.cpu_count() or 8
    o.options.max_iters = 1
    o.native.set_options(o.options)
    o.prepare_schedule(T0w + ticks + 4)
    for t in range(T0w):
        o.native.cycle(*o._table_for_tick(t % pd.t_mpc))
    o.tick = T0w
    o.enable_walk(per_instance=True)
    for _ in range(T0w):
        refgen.update_timings(o._walk["lists"][3], o._walk["lists"][2], o._walk["lists"][1], o._walk["lists"][0])
    assert o._walk["lists"] == a._walk["lists"]
    o._walk["x_measured_all"] = np.array(a._walk["x_measured_all"])[PICK].copy()
    replanned = 0
    for t in range(ticks):
        if t == 0:  # the oracle's first tick starts from the HIP iterate (shifted by k_shift's rule), then it walks on its own
            o._walk_references()
            o.native.cycle(*o._table_for_tick(o.tick % pd.t_mpc))
            o._walk_terminal()
            xs, us = ra["xs"][PICK], ra["us"][PICK]
            o.native.set_x0(xs[:, 1])
            o.native.setup()
            o.native.run(np.concatenate([xs[:, 1:], xs[:, -1:]], axis=1), np.concatenate([us[:, 1:], us[:, -1:]], axis=1))
            o.native.set_x0(None)
            o.tick += 1
            o._walk["x_measured_all"] = o.results(gains=False)["xs"][:, 1].copy()
        else:
            o.step()
        a.step()
        replanned += int(a._walk["replanning"])
        ra, ro = a.results(gains=True), o.results(gains=True)
        for key, floor in (("xs", 1e-3), ("us", 1e-1)):
            err = rel_cols(ra[key][PICK], ro[key], floor)
            assert err < 1e-6, "walk tick %d: %s deviates from the oracle by %.2e" % (T0w + t, key, err)
        errK = max(rel_tiles(ra["K"][i, 0], ro["K"][j, 0], 1e-6) for j, i in enumerate(PICK))
        assert errK < 1e-6, "walk tick %d: K_0 deviates from the oracle by %.2e" % (T0w + t, errK)
    assert replanned == ticks, "the compared ticks were meant to be replanning ticks"
    # the instances really have references of their own by now
    Lb = a._walk["last_all"][0]
    assert np.max(np.abs(Lb[PICK[1]] - Lb[PICK[0]])) > 1e-6
