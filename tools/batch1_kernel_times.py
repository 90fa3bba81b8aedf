"""Developer tool: per-kernel time of one MPC tick at batch 1 (HIP events around every kernel: the sum is above the untimed tick)."""
import sys, time, numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
legs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
one = EnsembleMPC(FullDynamicsProblem(horizon=100, complete_model=True), batch=1, library=lib, perturb=False, tick_reuse=True)
one.options.riccati_legs = legs
one.native.set_options(one.options)
one.prepare_schedule(60)
one.cold_solve(max_iters=100)
for _ in range(5):
    one.step()
one.native.profile(2); one.native.profile(1)
T = 20
t0 = time.perf_counter()
for _ in range(T):
    one.step()
one.results(gains=False)
wall = (time.perf_counter() - t0) / T * 1e3
one.native.profile(0)
tot = 0.0
for k, (cnt, ms) in sorted(one.native.profile_read().items(), key=lambda kv: -kv[1][1]):
    print("%-28s %6.3f ms per tick (%d launches)" % (k, ms / T, cnt))
    tot += ms / T
print("sum of kernels %.3f ms, wall per tick (with the event pairs) %.3f ms" % (tot, wall))
