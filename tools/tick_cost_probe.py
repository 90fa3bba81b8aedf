"""Developer tool (GPU box): what a tick of the benchmarked ensemble costs over the schedule, tick by tick — interval between completions with two
ticks in flight, instances that backtracked / took the corrector iteration, replanning ticks — and, with PROFILE=1, synchronous ticks with the
per-kernel times of chosen ticks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 300
refs = os.environ.get("REFS", "instance")
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
(e,) = make_bench_shards(pd, lib, 64, legs=4, tick_reuse=True)
e.options.corrector_prim_tol = float(os.environ.get("CORRECTOR", "20"))
e.options.refine_appended_knot = int(os.environ.get("REFINE", "0"))
e.native.set_options(e.options)
e.prepare_schedule(pd.t_mpc + 4)
e.cold_solve(max_iters=400)
e.enable_failure_isolation(auto_revive=True, source=0)
if refs != "frozen":
    e.enable_walk(per_instance=(refs == "instance"))
e.results(gains=False)
if os.environ.get("PROFILE"):
    for t in range(ticks):
        e.native.profile(2); e.native.profile(1)
        t0 = time.perf_counter()
        st = e.step()
        e.results(gains=False)
        dt = (time.perf_counter() - t0) * 1e3
        e.native.profile(0)
        nc = sum(1 for s in st if s.num_iters > 1); nb = sum(1 for s in st if s.alpha < 1.0)
        pr = e.native.profile_read()
        top = sorted(pr.items(), key=lambda kv: -kv[1][1])[:9]
        if nc or nb or t % 25 == 0:
            print("tick %3d %.2f ms corrector %2d backtrack %2d | %s" % (t, dt, nc, nb, " ".join("%s %dx%.2f" % (k.replace("k_", ""), c, m) for k, (c, m) in top)), flush=True)
    sys.exit(0)
t_prev = time.perf_counter(); rows = []
for t in range(ticks):
    rp0 = getattr(e, "replanning_ticks", 0)
    e.step_async()
    rep = getattr(e, "replanning_ticks", 0) - rp0
    if e.inflight == 2:
        st = e.wait(); now = time.perf_counter()
        rows.append(((now - t_prev) * 1e3, sum(1 for s in st if s.num_iters > 1), sum(1 for s in st if s.alpha < 1.0), rep)); t_prev = now
while e.inflight:
    e.wait()
rows = np.array(rows[2:])
for name, mask in (("plain ticks", (rows[:, 1] == 0) & (rows[:, 2] == 0)), ("ticks with a corrector iteration", rows[:, 1] > 0), ("ticks with backtracking only", (rows[:, 1] == 0) & (rows[:, 2] > 0))):
    v = rows[mask, 0]
    if v.size:
        print("%-34s %4d ticks: mean %.2f ms p50 %.2f p90 %.2f max %.2f" % (name, v.size, v.mean(), np.percentile(v, 50), np.percentile(v, 90), v.max()))
for t in range(0, len(rows), 1):
    if os.environ.get("VERBOSE"):
        print("%3d %.2f ms c %d b %d replanning %d" % (t + 2, *rows[t]))
