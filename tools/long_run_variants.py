import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
mode = sys.argv[1]
ens = EnsembleMPC(pd, batch=16, library=lib, seed=20250304, sigma_q=0.005, sigma_v=0.01)
ens.prepare_schedule(400)
ens.cold_solve(max_iters=100)
if mode == "iters2":
    ens.options.max_iters = 2; ens.native.set_options(ens.options)
try:
    for t in range(1, 331):
        desc, params = ens._table_for_tick(ens.tick % pd.t_mpc)
        ens.native.cycle(desc, params)
        if mode != "nosetup": ens.native.setup()
        st = ens.native.run_shifted()
        ens.tick += 1
        if t % 55 == 0:
            c = np.array([s.traj_cost for s in st]); pr = np.array([s.prim_infeas for s in st])
            print("  tick %3d cost med %.1f max %.1f prim med %.3f max %.3f" % (t, np.median(c), c.max(), np.median(pr), pr.max()))
    print(mode, "survived 330 ticks")
except RuntimeError as e:
    print(mode, "FAILED at tick", t, str(e)[-50:])
