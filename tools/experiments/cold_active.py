import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=False)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 36
prev = None
for iters in (30, 31, 32, 33):
    e.cold_solve(max_iters=iters)
    rows = []
    for k in range(101):
        act = e.native.debug_get("act", k, b); cv = e.native.debug_get("cval", k, b)
        rows.append((act.copy(), cv.copy()))
    nact = [int(np.count_nonzero(a)) for a, _ in rows]
    if prev is not None:
        flips = [(k, np.nonzero((rows[k][0] != 0) != (prev[k][0] != 0))[0].tolist()) for k in range(101) if np.any((rows[k][0] != 0) != (prev[k][0] != 0))]
        print("after %d iterations: active rows per knot (first 12) %s ; knots whose active set changed since the previous count: %s" % (iters, nact[:12], flips[:10]))
        for k, idx in flips[:4]:
            print("   knot %d rows %s: cval now %s, before %s" % (k, idx, np.round(rows[k][1][idx], 8).tolist(), np.round(prev[k][1][idx], 8).tolist()))
    prev = rows
