import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=False)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 36
for iters in (13, 14, 15):
    e.cold_solve(max_iters=iters)
    print("=== after", iters, "iterations")
    r = e.results(gains=False)
    for k in range(0, 8):
        act = e.native.debug_get("act", k, b); cv = e.native.debug_get("cval", k, b); lo = e.native.debug_get("lo", k, b); hi = e.native.debug_get("hi", k, b); ct = e.native.debug_get("ctype", k, b)
        du = e.native.debug_get("du", k, b)
        idx = np.nonzero(act)[0]
        print(" knot %d: active rows %s ctype %s cval %s lo %s hi %s | du: max %.3e at %d ; u there %.3f" % (
            k, idx.tolist(), ct[idx].tolist(), np.round(cv[idx], 7).tolist(), np.round(lo[idx], 4).tolist(), np.round(hi[idx], 4).tolist(), np.max(np.abs(du)), int(np.argmax(np.abs(du))), r["us"][b, k, int(np.argmax(np.abs(du)))]))
