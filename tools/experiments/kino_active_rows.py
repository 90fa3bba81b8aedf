import sys; sys.path.insert(0, '/root/repo')
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
kp = KinodynamicProblem(horizon=150, complete_model=True)
ens = EnsembleMPC(kp, batch=4, library=_capi.load_hip_library(), seed=7, perturb_dofs=range(18, kp.nv))
ens.options.riccati_legs = 4; ens.native.set_options(ens.options)
ens.prepare_schedule(60); ens.cold_solve(100)
for t in range(40):
    ens.step()
    if t in (5, 20, 39):
        ca = [int(np.count_nonzero(ens.native.debug_get("act", k, 1))) for k in range(150)]
        print("tick", t, "active rows per knot: min %d median %d max %d; knots with > 16: %d, > 32: %d" % (min(ca), np.median(ca), max(ca), sum(c > 16 for c in ca), sum(c > 32 for c in ca)), ca[:30])
