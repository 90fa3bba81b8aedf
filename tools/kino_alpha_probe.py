"""Developer probe: accepted step lengths of the kinodynamic ensemble per tick (why does the backtracking kernel run?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.load_hip_library()
kp = KinodynamicProblem(horizon=150, complete_model=True)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ens = EnsembleMPC(kp, batch=B, library=lib, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=True)
ens.options.riccati_legs = 4
ens.native.set_options(ens.options)
ens.prepare_schedule(40)
st = ens.cold_solve(max_iters=100)
print("cold: iters", [int(s.num_iters) for s in st][:8], "converged", sum(bool(s.converged) for s in st), "/", B)
for t in range(20):
    st = ens.step()
    al = np.array([s.alpha for s in st]); ls = np.array([s.ls_steps for s in st])
    print("tick %2d  alpha: min %.4f  #(<1) %2d  ls_steps max %d | prim %.2e dual %.2e | merit[0] %.6e" % (t, al.min(), int((al < 1).sum()), ls.max(), max(s.prim_infeas for s in st), max(s.dual_infeas for s in st), st[0].merit))
