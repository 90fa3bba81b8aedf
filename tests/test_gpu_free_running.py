"""HIP and the oracle FREE-RUNNING: both libraries start from the same cold solve and then walk the schedule on their own — the HIP handle is never
reset to the oracle's state — through the first take-off entering the horizon (tick 30 of the schedule).  The other walk tests compare one tick at a
time from the oracle's iterate; this one holds the TRAJECTORY of the loop: states, controls and K_0 of every tick within 1e-8 per component (measured:
6e-11 over 45 ticks, no growth), the same accepted step lengths and iteration counts.  Two settings: the reference loop's exact budget (one iteration,
plain warm start) and bench.py's (refinement of the appended knot + corrector)."""
import os

import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from tests._metrics import rel_cols

pytestmark = pytest.mark.gpu


def _handle(lib, refine, corrector, horizon=30, complete=False, sigma=(0.004, 0.01), schedule=60):
    e = EnsembleMPC(FullDynamicsProblem(horizon=horizon, complete_model=complete), batch=2, library=lib, seed=3, sigma_q=sigma[0], sigma_v=sigma[1])
    e.options.num_threads = os.cpu_count() or 8
    e.options.riccati_legs = 1
    e.options.refine_appended_knot = refine
    e.options.corrector_prim_tol = corrector
    e.native.set_options(e.options)
    e.prepare_schedule(schedule)
    e.cold_solve(max_iters=100)
    return e


@pytest.mark.parametrize("refine,corrector", [(0, 0.0), (3, 20.0)])
def test_free_running_walk_stays_with_the_oracle(hip_lib, oracle_lib, refine, corrector):
    er, eh = _handle(oracle_lib, refine, corrector), _handle(hip_lib, refine, corrector)
    worst = 0.0
    for t in range(45):
        sr, sh = er.step(), eh.step()
        assert [s.alpha for s in sh] == [s.alpha for s in sr] and [s.num_iters for s in sh] == [s.num_iters for s in sr], (t, [s.alpha for s in sh], [s.alpha for s in sr])
        a, b = eh.results(gains=True), er.results(gains=True)
        e = max(rel_cols(a["xs"], b["xs"], 1e-3), rel_cols(a["us"], b["us"], 1.0), rel_cols(a["K"][:, 0], b["K"][:, 0], 1.0))
        assert e < 1e-8, "tick %d of the free-running loops: HIP is %.3e away from the oracle" % (t, e)
        worst = max(worst, e)
    print("free-running, refine_appended_knot %d, corrector %g: worst deviation over 45 ticks %.3e" % (refine, corrector, worst))


def test_free_running_walk_at_full_size(hip_lib, oracle_lib):
    """BASELINE.json's problem itself — complete model nq = 39, N = 100, instances perturbed as the benchmark's (sigma_q 0.02, sigma_v 0.05), the
    reference loop's exact iteration budget, serial sweep in both libraries — free-running through the first take-off entering the horizon: within 1e-6
    per component on every one of 125 ticks, including the landing at tick 110 and its backtracking ticks (round 5 measured <= 6e-8 there,
    profiles/r05_free_running.txt, and ran 36 of them as a test) — the trajectory criterion of BASELINE.json held by the LOOP, not by one Newton step at a time."""
    TICKS = 125
    er, eh = (_handle(lib, 0, 0.0, horizon=100, complete=True, sigma=(0.02, 0.05), schedule=TICKS + 10) for lib in (oracle_lib, hip_lib))
    worst = 0.0
    for t in range(TICKS):
        sr, sh = er.step(), eh.step()
        assert [s.alpha for s in sh] == [s.alpha for s in sr], (t, [s.alpha for s in sh], [s.alpha for s in sr])
        a, b = eh.results(gains=True), er.results(gains=True)
        e = max(rel_cols(a["xs"], b["xs"], 1e-3), rel_cols(a["us"], b["us"], 1.0), rel_cols(a["K"][:, 0], b["K"][:, 0], 1.0))
        assert e < 1e-6, "tick %d of the free-running loops: HIP is %.3e away from the oracle" % (t, e)
        worst = max(worst, e)
    print("free-running at full size: worst deviation over %d ticks %.3e" % (TICKS, worst))
