import os, sys
sys.path.insert(0, '/root/repo')
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
sq, sv = float(sys.argv[1]), float(sys.argv[2])
ens = EnsembleMPC(pd, batch=64, library=lib, seed=20250304, sigma_q=sq, sigma_v=sv)
ens.prepare_schedule(30)
st = ens.cold_solve(max_iters=100)
print("sigma", sq, sv); print("fulldyn cold: converged %d/64, iters %s" % (sum(bool(s.converged) for s in st), sorted(set(int(s.num_iters) for s in st))))
worst = 0
for t in range(25):
    st = ens.step()
    worst = max(worst, max(s.prim_infeas for s in st))
print("after 25 ticks: max prim infeas over ticks %.3e, last dual %.3e, alpha set %s" % (worst, max(s.dual_infeas for s in st), sorted(set(round(s.alpha,4) for s in st))))
