import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.load_hip_library()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kp = KinodynamicProblem(horizon=N, complete_model=True)
ens = EnsembleMPC(kp, batch=B, library=lib, seed=7, perturb_dofs=range(18, kp.nv))
print("dims n=%d m=%d nc_max=%d" % (ens.dims.ndx, ens.dims.nu, ens.dims.nc_max))
ens.prepare_schedule(4)
ens.native.profile(2); ens.native.profile(1)
t0 = time.perf_counter()
stats = ens.cold_solve(max_iters=60)
dt = time.perf_counter() - t0
ens.native.profile(0)
print("cold solve %.2f s; iters: %s" % (dt, sorted(set(int(s.num_iters) for s in stats))))
print("converged %d / %d; prim max %.2e dual max %.2e" % (sum(bool(s.converged) for s in stats), B, max(s.prim_infeas for s in stats), max(s.dual_infeas for s in stats)))
for k, (c, ms) in sorted(ens.native.profile_read().items(), key=lambda kv: -kv[1][1])[:8]:
    print("%-28s launches %4d total %9.2f ms avg %8.3f ms" % (k, c, ms, ms / max(c, 1)))
bad = [i for i, s in enumerate(stats) if not s.converged][:5]
for i in bad:
    s = stats[i]; print("instance", i, "iters", s.num_iters, "prim", s.prim_infeas, "dual", s.dual_infeas, "mu", s.mu, "al", s.al_iters)
