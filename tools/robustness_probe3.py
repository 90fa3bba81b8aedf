"""Developer tool (GPU box): the per-instance-reference walk with refine_appended_knot = 3, one iteration per tick — per-tick status of the
instances that are lost (WATCH=3,22,...), from tick FROM on, and per-knot defects / constraint values at the ticks before the loss."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 480
watch = [int(v) for v in os.environ.get("WATCH", "0,3").split(",")]
frm = int(os.environ.get("FROM", "425"))
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=bool(int(os.environ.get("REUSE", "1"))))
e.options.refine_appended_knot = int(os.environ.get("REFINE", "3")); e.native.set_options(e.options)
e.prepare_schedule(ticks + 4); e.cold_solve(max_iters=100)
e.enable_failure_isolation(auto_revive=False)
e.enable_walk(per_instance=True)
N = 100
for t in range(ticks):
    st = e.step()
    if t >= frm:
        w = e._walk
        for b in watch:
            s = st[b]
            f = np.array([np.max(np.abs(e.native.debug_get("f", k, b))) for k in range(N)])
            c = np.array([np.max(np.abs(np.concatenate((e.native.debug_get("cval", k, b), [0.0])))) for k in range(N + 1)])
            Lb = w["last_all"][0][b][9:], w["last_all"][1][b][9:]
            print("tick %3d inst %2d conv %2d alpha %-8.3g prim %.2e dual %.2e cost %.3e | max|f| %.2e @%d | last refs L %s R %s | timings %s" % (
                t, b, s.converged, s.alpha, s.prim_infeas, s.dual_infeas, s.traj_cost, f.max(), f.argmax(), np.round(Lb[0], 3), np.round(Lb[1], 3), [l[:1] for l in w["lists"]]), flush=True)
