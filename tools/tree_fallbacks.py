"""Developer tool (GPU box): how many compositions of the cut tree (k_leg_compose) take the blocked elimination on the matrix cores and how
many fall back to the pivoted Gauss-Jordan — cold solve and MPC ticks, batch 1 at 32 legs and the benchmarked ensemble (64 x 4 legs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.load_hip_library()
for name, pd, B, legs, kw in (("full dynamics, batch 1, 32 legs", FullDynamicsProblem(horizon=100, complete_model=True), 1, 32, {"perturb": False}),
                              ("full dynamics, 64 instances, 4 legs", FullDynamicsProblem(horizon=100, complete_model=True), 64, 4, {}),
                              ("kinodynamic N = 150, 64 instances, 4 legs", KinodynamicProblem(horizon=150, complete_model=True), 64, 4, {"seed": 7, "perturb_dofs": range(18, 38)})):
    ens = EnsembleMPC(pd, batch=B, library=lib, tick_reuse=True, **kw)
    ens.options.riccati_legs = legs
    ens.native.set_options(ens.options)
    ens.prepare_schedule(140)
    ens.native.profile(3)
    def read():
        tot = np.zeros(2)
        for b in range(B):
            p = ens.native.debug_get("ric_prof", 0, b)
            tot += p[29:31]
        return tot
    read()
    ens.cold_solve(max_iters=100)
    c = read()
    for _ in range(120):
        ens.step()
    t = read()
    print("%-45s cold solve: %6d blocked, %5d pivoted ; 120 ticks: %6d blocked, %5d pivoted" % (name, c[0], c[1], t[0], t[1]))
