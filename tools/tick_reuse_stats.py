"""Developer tool: how often tick reuse applies on the default bench workload — per tick, the instances whose full step was not
accepted (their knots are all re-evaluated by the next tick's launch of the current point) and the time of that launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

B = int(os.environ.get("BATCH", "64"))
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=B, library=_capi.load_hip_library(), tick_reuse=True)
ens.options.riccati_legs = 4
ens.native.set_options(ens.options)
ens.prepare_schedule(60)
ens.cold_solve(100)
for t in range(40):
    ens.native.profile(2); ens.native.profile(1)
    st = ens.step()
    ens.native.profile(0)
    pr = ens.native.profile_read()
    alphas = np.array([s.alpha for s in st])
    print("tick %2d  alpha<1: %2d  no-step: %2d  k_eval_stage %.3f ms  trial %.3f ms  backtrack %.3f ms" % (
        t, int((alphas < 1).sum()), sum(1 for s in st if s.num_iters == 0), pr.get("k_eval_stage", (0, 0))[1],
        pr.get("k_eval_stage_trial", (0, 0))[1], pr.get("k_eval_stage_backtrack", (0, 0))[1]))
