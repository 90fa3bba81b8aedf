import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=4, library=lib, seed=20250304, sigma_q=float(sys.argv[1]), sigma_v=float(sys.argv[2]))
ens.prepare_schedule(400)
st = ens.cold_solve(max_iters=100)
print("cold", [(int(s.num_iters), bool(s.converged)) for s in st])
for t in range(1, 331):
    st = ens.step()
    if t % 20 == 0:
        r = ens.results(gains=False)
        zb = r["xs"][:, 0, 2]
        print("tick %3d phase %s cost %s prim %s base z %s" % (t, pd.contact_phases[t % pd.t_mpc], np.round([s.traj_cost for s in st], 2), np.round([s.prim_infeas for s in st], 3), np.round(zb, 3)))
