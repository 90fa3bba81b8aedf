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
umax = pd.umax
for t in range(1, 150):
    st = ens.step()
    if t in (20, 60, 100, 125, 140):
        r = ens.results(gains=False)
        us = r["us"][0]; xs = r["xs"][0]
        fmax = max(np.max(np.abs(ens.native.debug_get("f", k))) for k in range(0, 100))
        kworst = int(np.argmax([np.max(np.abs(ens.native.debug_get("f", k))) for k in range(100)]))
        print("tick", t, "phase0", pd.contact_phases[t % pd.t_mpc], "cost %.1f prim %.3f" % (st[0].traj_cost, st[0].prim_infeas),
              "| max|f| %.3f at knot %d" % (fmax, kworst), "| max |u|/umax %.2f" % np.max(np.abs(us) / umax[None, :]),
              "| max joint vel %.2f" % np.max(np.abs(xs[:, pd.robot.model.nq + 6:])), "| base z range %.3f..%.3f" % (xs[:, 2].min(), xs[:, 2].max()))
        w = [ens.native.get_stage_data(k)[1][0] for k in (0, 50, 99)]
        print("   wrenches fz (L,R) at knots 0,50,99:", [(round(x[0][2], 1), round(x[1][2], 1)) for x in w], " mass*g = %.1f" % (pd.robot.mass * 9.81))
