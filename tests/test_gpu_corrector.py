"""``mpc_options.corrector_prim_tol`` / ``corrector_window`` on the device (csrc/solver_kernels.h k_after_step, csrc/mpc_hip.hip
corrector_armed) against the oracle's rule (oracle/solver.hpp run_instance, tests/test_corrector.py):

  * tick by tick from the oracle's solver state across both kinds of pattern change (double -> single support at tick 30 of the schedule,
    single -> double at tick 110), reduced horizon: the same instances take the extra iteration on the same ticks and end at the same
    iterate, synchronous entry point;
  * the asynchronous entry point (two ticks in flight, the corrector pass enqueued ahead on the ticks of the window) equals the synchronous
    one bit for bit, with tick reuse;
  * the benchmarked ensemble (64 randomised instances, N = 100, complete model, two ticks in flight) over the reference's whole schedule
    with max_iters = 1: nobody is lost with bench.py's settings; with the scripts' plain warm start the nominal instance and all but a
    handful of the perturbed ones."""
import os

import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC, make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from tests._metrics import rel_cols

pytestmark = pytest.mark.gpu


def _handle(lib, horizon, tol, window, batch=2, **kw):
    e = EnsembleMPC(FullDynamicsProblem(horizon=horizon), batch=batch, library=lib, seed=3, sigma_q=0.004, sigma_v=0.01, **kw)
    e.options.num_threads = os.cpu_count() or 8
    e.options.riccati_legs = 1
    e.options.corrector_prim_tol = tol
    e.options.corrector_window = window
    e.native.set_options(e.options)
    e.prepare_schedule(130)
    return e


@pytest.mark.parametrize("window", [3])   # (window 0 — the rule on every tick — is held by test_gpu_free_running.py's (3, 20.0) walk; one parametrisation here keeps the suite in its budget)
def test_corrector_matches_oracle_tick_by_tick(hip_lib, oracle_lib, window):
    er, eh = _handle(oracle_lib, 30, 5.0, window), _handle(hip_lib, 30, 5.0, window)
    er.cold_solve(max_iters=100)
    eh.cold_solve(max_iters=100)
    worst, fired = 0.0, []
    for t in range(116):
        eh.native.set_state(er.native.get_state())  # every tick from the oracle's iterate (the state carries the window counter)
        sr, sh = er.step(), eh.step()
        assert [s.num_iters for s in sh] == [s.num_iters for s in sr], (t, [s.num_iters for s in sh], [s.num_iters for s in sr])
        assert [s.alpha for s in sh] == [s.alpha for s in sr], (t, [s.alpha for s in sh], [s.alpha for s in sr])
        a, b = eh.results(gains=True), er.results(gains=True)
        e = max(rel_cols(a["xs"], b["xs"], 1e-3), rel_cols(a["us"], b["us"], 1.0), rel_cols(a["K"][:, 0], b["K"][:, 0], 1.0))
        assert e < 1e-6, "tick %d: deviates from the oracle by %.3e" % (t, e)
        worst = max(worst, e)
        if any(s.num_iters == 2 for s in sr):
            fired.append(t)
    assert 30 in fired and 110 in fired, fired
    if window:
        assert all(30 <= t < 30 + window or 110 <= t < 110 + window for t in fired), fired
    print("corrector (window %d): extra iteration on ticks %s ; worst deviation from the oracle %.3e" % (window, fired, worst))


@pytest.mark.parametrize("window", [0, 3])
def test_async_ticks_equal_synchronous_ticks(hip_lib, window):
    """mpc_run_shifted_async enqueues the corrector pass ahead (every tick for window 0, the ticks of the window otherwise); instances that do
    not need it sit it out.  Same iterates as mpc_run_shifted, bit for bit, tick reuse on, four instances of which some take the extra
    iteration and some do not."""
    def run(asynchronous):
        e = _handle(hip_lib, 30, 8.0, window, batch=4, tick_reuse=True)
        e.cold_solve(max_iters=100)
        iters = []
        for t in range(40):
            if asynchronous:
                e.step_async()
                if e.inflight == 2:
                    iters.append([s.num_iters for s in e.wait()])
            else:
                iters.append([s.num_iters for s in e.step()])
        while e.inflight:
            iters.append([s.num_iters for s in e.wait()])
        return e.results(gains=True), iters
    ra, ia = run(True)
    rs, isy = run(False)
    assert ia == isy
    assert any(2 in row for row in isy) and any(1 in row for row in isy)
    for key in ("xs", "us", "K"):
        assert np.array_equal(ra[key], rs[key]), key


@pytest.mark.parametrize("refs,refine,window", [("frozen", 3, 8), ("device", 3, 8), ("device-floor", 3, 8), ("instance", 3, 0), ("instance", 0, 0)])
def test_whole_schedule_on_one_iteration_per_tick(hip_lib, refs, refine, window):
    """BASELINE.json's ensemble as bench.py runs it — 64 randomised instances, N = 100, complete model, 4 legs, tick reuse, two ticks in
    flight, max_iters = 1, corrector 20.0 — over the reference's whole 1000-tick schedule.
    refine_appended_knot = 3 (bench.py's setting, with the corrector on the 8 ticks after a pattern change; also on every tick): nobody is lost, with frozen references and with every instance replanning from its own
    measured feet (round 4, refinement alone: 5 of 64 lost with per-instance references; neither: 40 - 91, the nominal instance among them).
    refine_appended_knot = 0 (the scripts' own warm start, us[-1] duplicated, fulldynamic_talos.py:532-534): the nominal instance walks the
    whole schedule and at most a handful of the 63 perturbed ones are lost and re-seeded (0 - 3 from run to run of the settings explored in
    profiles/r05_robustness.txt; 46 / 40 without the corrector)."""
    pd = FullDynamicsProblem(horizon=100, complete_model=True)
    (e,) = make_bench_shards(pd, hip_lib, 64, legs=4, tick_reuse=True)
    e.options.refine_appended_knot = refine
    e.options.corrector_prim_tol = 20.0
    e.options.corrector_window = window
    e.native.set_options(e.options)
    e.iters_per_tick = 1
    e.prepare_schedule(pd.t_mpc + 4)
    e.cold_solve(max_iters=100)
    e.enable_failure_isolation(auto_revive=True, source=0)
    if refs != "frozen":   # "device": the reference generator in the library (mpc_walk_*), which is what bench.py runs ; "instance": the numpy generator
        e.enable_walk(per_instance=True, generator="device" if refs.startswith("device") else "host", floor=refs.endswith("floor"))   # "device-floor": bench.py's setting
    ticks = min(1000, pd.t_mpc - 1)
    extra, total, worst_prim = 0, 0, 0.0
    for t in range(ticks):
        e.step_async()
        if e.inflight == 2:
            st = e.wait()
            extra += sum(1 for s in st if s.num_iters > 1); total += len(st)
            worst_prim = max([worst_prim] + [s.prim_infeas for s in st if s.converged >= 0])
    while e.inflight:
        st = e.wait()
    print("refine_appended_knot %d + corrector (window %d), references %s, 64 instances, %d ticks: lost %s ; %d of %d instance-ticks took the extra iteration ; largest primal infeasibility seen %.2e"
          % (refine, window, refs, ticks, [r[:3] for r in e.lost], extra, total, worst_prim))
    assert getattr(e, "rescues", 0) == 0 and e.tick == ticks
    assert all(r[1] != 0 for r in e.lost), "the nominal instance failed"
    assert len(e.lost) <= (0 if refine else 4), e.lost
    assert extra < 0.10 * total
    r = e.results(gains=False)
    alive = [b for b, s in enumerate(st) if s.converged >= 0]
    assert np.all(np.isfinite(r["xs"][alive])) and np.all(np.isfinite(r["us"][alive]))
