"""Developer tool (GPU box): the bench ensemble over a long stretch of the schedule WITHOUT episode restarts, frozen references or
walk mode — cost / infeasibility statistics every 20 ticks, failures.  usage: python tools/long_walk.py [walk|frozen] [ticks] [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

mode = sys.argv[1] if len(sys.argv) > 1 else "walk"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
pd = FullDynamicsProblem(horizon=100, complete_model=True)
closed = (10, pd.dt / 10) if os.environ.get("CLOSED") else None  # CLOSED=1: measured states from the simulation stand-in (N2)
(e,) = make_bench_shards(pd, _capi.load_hip_library(), batch, legs=4, tick_reuse=True, closed_loop=closed)
e.prepare_schedule(ticks + 4)
st = e.cold_solve(max_iters=100)
print("cold: converged %d/%d, iterations %s" % (sum(bool(s.converged) for s in st), batch, sorted(int(s.num_iters) for s in st)[-8:]))
if mode == "walk":
    e.enable_walk()
if os.environ.get("ITERS"):  # ITERS=2: two ProxDDP iterations per tick
    e.options.max_iters = int(os.environ["ITERS"]); e.native.set_options(e.options)
t0 = time.perf_counter()
try:
    for t in range(1, ticks + 1):
        st = e.step()
        if t % 20 == 0:
            c = np.array([s.traj_cost for s in st]); pr = np.array([s.prim_infeas for s in st]); al = np.array([s.alpha for s in st])
            print("tick %4d cost med %9.2f max %10.2f | prim med %.2e max %.2e | alpha<1: %2d no-step: %2d | replanning so far %d" % (
                t, np.median(c), c.max(), np.median(pr), pr.max(), int((al < 1).sum()), sum(1 for s in st if s.num_iters == 0), getattr(e, "replanning_ticks", 0)))
    print(mode, "survived", ticks, "ticks, %.1f ms per tick (synchronous)" % ((time.perf_counter() - t0) / ticks * 1e3))
except RuntimeError as ex:
    print(mode, "FAILED at tick", t, str(ex)[-80:])
