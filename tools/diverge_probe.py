import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib=_capi.load_hip_library()
pd=FullDynamicsProblem(horizon=100, complete_model=True)
ens=EnsembleMPC(pd, batch=64, library=lib, seed=20250304)
ens.prepare_schedule(260)
ens.cold_solve(100)
hist=[]
for t in range(250):
    try:
        st=ens.step()
    except RuntimeError as e:
        print("tick", t, "error:", e)
        bad=int(str(e).split("instance ")[1].split()[0])
        for tt,h in enumerate(hist[-25:]):
            s=h[bad]; print(len(hist)-25+tt, "cost %.4e merit %.4e prim %.2e dual %.2e alpha %.3g ls %d mu %.1e" % s)
        break
    hist.append([(s.traj_cost, s.merit, s.prim_infeas, s.dual_infeas, s.alpha, s.ls_steps, s.mu) for s in st])
else:
    print("no failure in 250 ticks")
# how many instances have alpha < 1 over time
al=np.array([[h[b][4] for b in range(64)] for h in hist])
print("ticks with any alpha<1:", [(t, int((al[t]<1).sum())) for t in range(len(al)) if (al[t]<1).any()][:40])
