import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=True)
st = e.cold_solve(max_iters=int(sys.argv[1]) if len(sys.argv) > 1 else 120)
print("converged", sum(bool(s.converged) for s in st))
