"""Developer tool (GPU box): the benchmarked ensemble (64 instances, N = 100, complete model, 4 legs, tick reuse) with the whole solution of
every instance (xs, us, K_0: what the scripts read from `results` after a solve) brought to the host after EVERY tick, against the
resident loop of bench.py — the PCIe-inclusive rate of DESIGN.md section 5.  Synchronous ticks in both cases (one tick in flight)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=True)
e.options.refine_appended_knot = 3
e.native.set_options(e.options)
e.prepare_schedule(140)
e.cold_solve(max_iters=400)
for _ in range(5):
    e.step()
T = 40
t0 = time.perf_counter()
for _ in range(T):
    e.step()
t1 = time.perf_counter()
nbytes = 0
for _ in range(T):
    e.step()
    r = e.results(gains=False)
    K0, k0 = e.native.get_gain(0)
    nbytes = r["xs"].nbytes + r["us"].nbytes + K0.nbytes + k0.nbytes
t2 = time.perf_counter()
a, b = (t1 - t0) / T * 1e3, (t2 - t1) / T * 1e3
print("synchronous ticks of 64 instances: resident %.3f ms per tick (%.0f solves/s) ; with xs, us, K_0 of every instance downloaded after every tick (%.2f MB) %.3f ms (%.0f solves/s)" % (
    a, 64e3 / a, nbytes / 1e6, b, 64e3 / b))
