import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=1, library=lib, perturb=False)
ens.prepare_schedule(400)
ens.cold_solve(max_iters=100)
lf, rf = pd.robot.foot_placements
print("feet y: L %.3f R %.3f ; com0 %s ; foot half width %.3f" % (lf.translation[1], rf.translation[1], np.round(pd.robot.com0, 3), 0.0))
for t in range(1, 131):
    st = ens.step()
    if t in (40, 80, 110, 129):
        r = ens.results(gains=False)
        xs = r["xs"][0]
        ks = [0, 20, 40, 60, 80, 99]
        W = [ens.native.get_stage_data(k)[1][0] for k in ks]
        print("tick %d cost %.1f prim %.2f | base y over horizon %s | base roll(qx) %s" % (t, st[0].traj_cost, st[0].prim_infeas, np.round(xs[ks, 1], 4), np.round(xs[ks, 3], 4)))
        print("    left foot: fz %s  tau_x/fz (CoP y) %s  tau_y/fz %s" % (np.round([w[0][2] for w in W], 0), np.round([w[0][3] / max(w[0][2], 1e-9) for w in W], 3), np.round([-w[0][4] / max(w[0][2], 1e-9) for w in W], 3)))
