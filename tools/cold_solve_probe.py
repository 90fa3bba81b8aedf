"""Developer tool (GPU box): the cold solve of the bench ensemble — which randomised instances do not reach tol within the budget, and why."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
pd = FullDynamicsProblem(horizon=100, complete_model=True)
for iters in [int(v) for v in (sys.argv[1:] or ["100", "300"])]:
    (e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=True)
    st = e.cold_solve(max_iters=iters)
    bad = [b for b, s in enumerate(st) if not s.converged]
    print("budget %d: converged %d/64; iterations of the converged: median %d max %d" % (
        iters, 64 - len(bad), np.median([s.num_iters for s in st if s.converged]), max(s.num_iters for s in st if s.converged)))
    mdl = e.problem.stages[0].xspace.model
    for b in bad:
        s = st[b]
        q = e.x0[b][7:mdl.nq]
        at_lo = int(np.sum(q <= mdl.lowerPositionLimit[7:] + 1e-12)); at_hi = int(np.sum(q >= mdl.upperPositionLimit[7:] - 1e-12))
        print("  instance %2d: iters %3d al %2d prim %.2e dual %.2e mu %.1e cost %.4e | joints of x0 clipped to a limit: %d lower %d upper" % (
            b, s.num_iters, s.al_iters, s.prim_infeas, s.dual_infeas, s.mu, s.traj_cost, at_lo, at_hi))
