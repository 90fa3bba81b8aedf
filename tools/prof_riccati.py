import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=4, library=lib)
ens.prepare_schedule(10)
ens.cold_solve(100)
ens.native.debug_get('ric_prof', 0)
for _ in range(3): ens.step()
p = ens.native.debug_get('ric_prof', 0)
names = ['T6inv+actscan','T6 transform','LP build+vv','chol n','LI inv','trsm','sym,w,store Pt','AB load+gh','panels','KKT prep','chol m','KKT solve','value+store']
tot = p.sum()
for n_, v in zip(names, p): print('%-18s %8.1f us/knot  %5.1f%%' % (n_, v/300/2400.0*1.0, 100*v/tot))
print('total us/knot', tot/300/2400)
