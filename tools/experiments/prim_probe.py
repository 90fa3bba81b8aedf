"""Developer probe (GPU box): WHICH residual carries the primal infeasibility of an instance at a given tick — dynamics gap or constraint row, which knot —
beside the nominal instance.  args: seed instance tick"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
seed, inst, tick = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=False, seed=seed)
e.options.refine_appended_knot = 3; e.options.corrector_prim_tol = 20.0; e.options.corrector_window = 8
e.native.set_options(e.options)
e.prepare_schedule(pd.t_mpc + 4)
e.cold_solve(max_iters=100)
e.enable_failure_isolation(auto_revive=False)
e.enable_walk(per_instance=True, generator="device")
for t in range(tick + 1):
    st = e.step()
print("tick %d: inst %d prim %.3e alpha %g | nominal prim %.3e" % (tick, inst, st[inst].prim_infeas, st[inst].alpha, st[0].prim_infeas))
x0 = e.x0s if hasattr(e, "x0s") else None
r = e.results(gains=False)
print("perturbation of the instance's initial configuration (joints, rad):", np.round(np.asarray(e.x0)[inst][7:pd.robot.nq] - np.asarray(e.x0)[0][7:pd.robot.nq], 3).tolist())
for b in (inst, 0):
    rows = []
    for k in range(101):
        f = np.asarray(e.native.debug_get("f", k, b)).ravel()
        cv = np.asarray(e.native.debug_get("cval", k, b)).ravel(); lo = np.asarray(e.native.debug_get("lo", k, b)).ravel(); hi = np.asarray(e.native.debug_get("hi", k, b)).ravel()
        ct = np.asarray(e.native.debug_get("ctype", k, b)).ravel()[:cv.size]
        viol = np.where(ct == 1, np.abs(cv), np.where(ct == 2, np.maximum(cv, 0.0), np.maximum(np.maximum(lo - cv, cv - hi), 0.0))) if cv.size else np.zeros(1)  # equality / negative orthant / box
        rows.append((k, float(np.max(np.abs(f))) if f.size else 0.0, int(np.argmax(np.abs(f))) if f.size else -1, float(viol.max()), int(viol.argmax())))
    rows.sort(key=lambda r_: -max(r_[1], r_[3]))
    print("instance %d: the five knots with the largest residual (knot, |dynamics gap| max, its row, constraint violation max, its row):" % b)
    for r_ in rows[:5]:
        print("   knot %3d  gap %.3e (row %d)  constraint %.3e (row %d)" % r_)
    k = rows[0][0]
    cv = np.asarray(e.native.debug_get("cval", k, b)).ravel(); lo = np.asarray(e.native.debug_get("lo", k, b)).ravel(); hi = np.asarray(e.native.debug_get("hi", k, b)).ravel()
    i = rows[0][4]
    print("   at knot %d: row %d value %.4f bounds [%.4g, %.4g] ; rows by block: %s" % (k, i, cv[i], lo[i], hi[i], [int(v) for v in np.asarray(e.native.debug_get("ctype", k, b)).ravel()[:cv.size]][:80]))
    print("   us[%d] joint torques of the violated row's neighbourhood: %s" % (k if k < 100 else 99, np.round(r["us"][b, min(k, 99)], 1).tolist()))
