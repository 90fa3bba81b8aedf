"""Developer probe (GPU box): BASELINE config 4 (kinodynamic stairs, N = 150, 64 instances) over the whole schedule on one iteration per tick under
refine_appended_knot / corrector settings: args = list of "refine,corrector" pairs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.load_hip_library()
for pair in sys.argv[1:]:
    refine, corr = pair.split(",")
    kp = KinodynamicProblem(horizon=150, complete_model=True)
    ens = EnsembleMPC(kp, batch=64, library=lib, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=True)
    ens.options.riccati_legs = 4
    ens.options.refine_appended_knot = int(refine)
    ens.options.corrector_prim_tol = float(corr)
    ens.native.set_options(ens.options)
    ticks = kp.t_mpc - 1
    ens.prepare_schedule(ticks + 4)
    ens.cold_solve(max_iters=100)
    ens.enable_walk(z_height=float(os.environ.get("Z", "0.10")))
    ens.enable_failure_isolation(auto_revive=True, source=0)
    t0 = time.time(); extra = 0; back = 0; first_nominal = None
    for t in range(ticks):
        st = ens.step()
        extra += sum(1 for s in st if s.num_iters > 1); back += sum(1 for s in st if s.alpha < 1 and s.converged >= 0)
        if st[0].converged < 0 and first_nominal is None:
            first_nominal = t
    print("stairs z %s refine %s corrector %s: %d losses (nominal lost at %s), first %s ; corrector instance-ticks %d, backtracking %d ; %.1f s" % (
        os.environ.get("Z", "0.10"), refine, corr, len(ens.lost), first_nominal, ens.lost[:3], extra, back, time.time() - t0), flush=True)
    del ens
