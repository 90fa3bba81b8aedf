import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
sq, sv = float(sys.argv[1]), float(sys.argv[2])
ens = EnsembleMPC(pd, batch=64, library=lib, seed=20250304, sigma_q=sq, sigma_v=sv, perturb_dofs=(range(18, pd.nv) if len(sys.argv) > 3 and sys.argv[3] == 'upper' else None),
                  closed_loop=((10, pd.dt / 10) if 'closed' in sys.argv else None))
if os.environ.get("LEGS"):
    ens.options.riccati_legs = int(os.environ["LEGS"]); ens.native.set_options(ens.options)
ens.prepare_schedule(400)
st = ens.cold_solve(max_iters=100)
print("sigma", sq, sv, "cold converged", sum(bool(s.converged) for s in st), "cost med %.1f max %.1f" % (np.median([s.traj_cost for s in st]), max(s.traj_cost for s in st)))
try:
    for t in range(1, 331):
        st = ens.step()
        if t % 55 == 0:
            c = np.array([s.traj_cost for s in st]); pr = np.array([s.prim_infeas for s in st])
            print("  tick %3d cost med %.1f max %.1f prim med %.3f max %.3f alpha min %.3g" % (t, np.median(c), c.max(), np.median(pr), pr.max(), min(s.alpha for s in st)))
    print("  survived 330 ticks")
except RuntimeError as e:
    print("  FAILED at tick", t, str(e)[-60:])
