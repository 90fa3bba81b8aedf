import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.load_hip_library()
kp = KinodynamicProblem(horizon=150, complete_model=True)
ens = EnsembleMPC(kp, batch=64, library=lib, seed=7, perturb_dofs=range(18, kp.nv))
ens.prepare_schedule(30)
ens.cold_solve(max_iters=100)
for _ in range(3): ens.step()
ens.native.profile(2); ens.native.profile(1)
T=15
t0=time.perf_counter()
for _ in range(T): ens.step()
dt=(time.perf_counter()-t0)/T
ens.native.profile(0)
print("kino N=150 B=64: %.2f ms per tick" % (dt*1e3))
for k,(c,ms) in sorted(ens.native.profile_read().items(), key=lambda kv:-kv[1][1])[:9]:
    print("  %-26s launches/tick %.2f  ms/tick %.3f" % (k, c/T, ms/T))
