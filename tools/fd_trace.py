import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=4, library=lib, seed=20250304, sigma_q=0.02, sigma_v=0.05)
ens.prepare_schedule(4)
st = ens.cold_solve(max_iters=100)
print([(int(s.num_iters), bool(s.converged)) for s in st])
