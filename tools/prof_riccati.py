import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.bind_library(sys.argv[1]) if len(sys.argv) > 1 else _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=4, library=lib)
ens.prepare_schedule(10)
ens.cold_solve(100)
ens.native.profile(3)  # in-kernel phase timers on
ens.native.debug_get('ric_prof', 0)
for _ in range(3): ens.step()
p = ens.native.debug_get('ric_prof', 0)
names = ['T6inv+actscan','T6 transform','LP build+vv','chol n','LI inv','trsm','sym,w,store Pt','AB load+gh','panels','KKT prep','chol m','KKT solve','value+store']
tot = p[:32].sum()
for n_, v in zip(names, p): print('%-18s %8.1f us/knot  %5.1f%%' % (n_, v/300/2400.0*1.0, 100*v/tot))
print('total us/knot', p[:32].sum()/300/2400)
ev = p[32:45]
enames = ['load,FK,J','vel,inertia,composites,U','M,bias,contacts,Jc','chol M','Minv (potrs)','X,S,Kinv','solve,forces','deriv blocks','dr rows','dsol gemm','integrator,AB','terms','merit']
for n_, v in zip(enames, ev): print('EVAL %-26s %8.1f us  %5.1f%%' % (n_, v/3/2400.0, 100*v/ev.sum()))
print('EVAL total us per workgroup', ev.sum()/3/2400)
tn = ['', 'state_error', 'control_error', 'frame_placement', 'frame_translation', 'frame_velocity', 'com_translation', 'centroidal_momentum',
      'contact_force', 'mb_wrench_cone', 'centroidal_wrench_cone', 'lin_acc', 'ang_acc', 'centroidal_momentum_der']
for t in range(1, 14):
    if p[32 + 13 + t] > 0: print('EVAL term %-24s %8.1f us' % (tn[t], p[32 + 13 + t] / 3 / 2400.0))
print('EVAL flush_stack %8.1f us' % (p[32 + 27] / 3 / 2400.0))
print('RIC series: max rho %.3e, mean terms %.2f, chol fallbacks/knot %.3f' % (p[20], p[21] / 300, p[22] / 300))
for i_, n_ in ((13, 'norm+abr issue'), (14, 'series mma (wave 0)'), (15, 'series barrier wait'), (16, 'series store+barrier')):
    print('RIC fine %-24s %8.1f us/knot' % (n_, p[i_] / 300 / 2400.0))
