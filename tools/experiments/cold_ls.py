import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=int(os.environ.get("LEGS", "4")), tick_reuse=False)
b = int(sys.argv[1]) if len(sys.argv) > 1 else 36
for iters in (12, 13, 14, 15, 20, 31):
    st = e.cold_solve(max_iters=iters)
    ls = e.native.debug_get("ls", 0, b)
    print("after %2d iterations: phi0 %.9e dphi0 %.3e alpha %.4g | phi(alpha_i) - phi0 for alpha = 1, 1/2, ...: %s | ratio to alpha dphi0: %s" % (
        iters, ls[0], ls[1], ls[2], " ".join("%.2e" % (v - ls[0]) for v in ls[4:]), " ".join("%.2f" % ((v - ls[0]) / (ls[1] * 0.5 ** i)) for i, v in enumerate(ls[4:]))))
    du = [float(np.max(np.abs(e.native.debug_get("du", k, b)))) for k in range(100)]
    dx = [float(np.max(np.abs(e.native.debug_get("dx", k, b)))) for k in range(101)]
    print("     max|du| %.3e @%d   max|dx| %.3e @%d" % (max(du), int(np.argmax(du)), max(dx), int(np.argmax(dx))))
