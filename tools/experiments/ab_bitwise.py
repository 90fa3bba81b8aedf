"""Developer tool: the same batch-1 / batch-64 MPC ticks with two builds of the HIP library — results compared bit for bit, tick times side by side.
usage: python tools/experiments/ab_bitwise.py libA.so libB.so"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
res = {}
for path in sys.argv[1:3]:
    lib = _capi.bind_library(path)
    out = []
    for batch, legs, ticks in ((1, 32, 60), (8, 4, 20)):
        pd = FullDynamicsProblem(horizon=100, complete_model=True)
        e = EnsembleMPC(pd, batch=batch, library=lib, tick_reuse=True, perturb=(batch > 1))
        e.options.riccati_legs = legs; e.native.set_options(e.options)
        e.prepare_schedule(ticks + 8)
        e.cold_solve(max_iters=100)
        lat = []
        for t in range(ticks):
            t0 = time.perf_counter(); e.step(); lat.append((time.perf_counter() - t0) * 1e3)
        r = e.results(gains=True)
        out.append((r["xs"].copy(), r["us"].copy(), r["K"].copy(), float(np.percentile(lat[5:], 50))))
    res[path] = out
a, b = [res[p] for p in sys.argv[1:3]]
for i, name in enumerate(("batch 1, 32 legs", "batch 8, 4 legs")):
    same = all(np.array_equal(x, y) for x, y in zip(a[i][:3], b[i][:3]))
    print("%s: bit-identical %s ; p50 per tick %.3f ms -> %.3f ms" % (name, same, a[i][3], b[i][3]))
