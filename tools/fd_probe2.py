import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
sq, sv = float(sys.argv[1]), float(sys.argv[2])
ens = EnsembleMPC(pd, batch=64, library=lib, seed=20250304, sigma_q=sq, sigma_v=sv)
ens.prepare_schedule(200)
st = ens.cold_solve(max_iters=int(sys.argv[3]) if len(sys.argv) > 3 else 100)
def show(tag, st):
    pr = np.array([s.prim_infeas for s in st]); du = np.array([s.dual_infeas for s in st]); co = np.array([s.traj_cost for s in st])
    print("%-10s prim med %.2e max %.2e | dual med %.2e max %.2e | cost med %.3f max %.3f | conv %d" % (tag, np.median(pr), pr.max(), np.median(du), du.max(), np.median(co), co.max(), sum(bool(s.converged) for s in st)))
show("cold", st)
for t in range(1, 151):
    st = ens.step()
    if t % 25 == 0 or t in (1, 5, 10): show("tick %d" % t, st)
