"""Developer probe (GPU box): what happens to the instances that are lost late in the schedule (ticks 780 - 800) with per-instance references: foothold plan
(mpc_walk_get_state: start / final pose of both feet), base position and solver statistics of one instance beside the nominal one.  args: seed instance"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
seed, inst = int(sys.argv[1]), int(sys.argv[2])
t_from = int(sys.argv[3]) if len(sys.argv) > 3 else 600
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), 64, legs=4, tick_reuse=True, seed=seed)
e.options.refine_appended_knot = int(os.environ.get("REFINE", "3")); e.options.corrector_prim_tol = float(os.environ.get("CORRECTOR", "20")); e.options.corrector_window = int(os.environ.get("WINDOW", "8"))
e.native.set_options(e.options)
e.prepare_schedule(pd.t_mpc + 4)
e.cold_solve(max_iters=100)
e.enable_failure_isolation(auto_revive=False)
e.enable_walk(per_instance=True, generator="device", floor=bool(os.environ.get("FLOOR")))
for t in range(999):
    st = e.step()
    if t >= t_from and (t % int(os.environ.get("EVERY", "10")) == 0 or st[inst].alpha < 1.0 or st[inst].num_iters > 1 or st[inst].converged < 0):
        plan = np.asarray(e.native.walk_get_state()).reshape(64, 48)
        r = e.results(gains=False)
        def yx(p): return "(%.3f %.3f %.3f)" % (p[9], p[10], p[11])
        row = lambda b: "x0 base (%.3f %.3f %.3f) | L start %s final %s | R start %s final %s" % (r["xs"][b, 0, 0], r["xs"][b, 0, 1], r["xs"][b, 0, 2], yx(plan[b, 0:12]), yx(plan[b, 12:24]), yx(plan[b, 24:36]), yx(plan[b, 36:48]))
        cs = pd.contact_phases[max(0, t + 1 - 100) % pd.t_mpc], pd.contact_phases[(t + 1) % pd.t_mpc]
        print("tick %3d knot0 %s appended %s | inst %d: alpha %.4g iters %d prim %.2e cost %.1f conv %d | %s" % (t, list(cs[0]), list(cs[1]), inst, st[inst].alpha, st[inst].num_iters, st[inst].prim_infeas, st[inst].traj_cost, st[inst].converged, row(inst)))
        print("                                     nominal: alpha %.4g iters %d prim %.2e cost %.1f | %s" % (st[0].alpha, st[0].num_iters, st[0].prim_infeas, st[0].traj_cost, row(0)), flush=True)
    if st[inst].converged < 0:
        break
