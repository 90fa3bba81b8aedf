"""Developer tool: where the host time of a walk-mode tick goes (per-instance references, one shard of BATCH instances).
usage: python tools/walk_host_profile.py [BATCH]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
pd = FullDynamicsProblem(horizon=100, complete_model=True)
e = EnsembleMPC(pd, batch=B, library=_capi.load_hip_library(), tick_reuse=True, perturb=False)  # (nominal instances: the host work is the same)
e.options.riccati_legs = 4
e.native.set_options(e.options)
e.prepare_schedule(400)
e.cold_solve(100)
e.enable_walk(per_instance=True)
for _ in range(110):  # into the first replanning window
    e.step_async()
    if e.inflight == 2:
        e.wait()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
T = 60
for _ in range(T):
    e.step_async()
    if e.inflight == 2:
        e.wait()
pr.disable()
dt = time.perf_counter() - t0
while e.inflight:
    e.wait()
print("%.3f ms per tick wall (batch %d)" % (dt / T * 1e3, B))
st = pstats.Stats(pr)
st.sort_stats("cumulative")
rows = []
for (fn, ln, name), (cc, nc, tt, ct, _) in st.stats.items():
    rows.append((ct, tt, nc, "%s:%d %s" % (os.path.basename(fn), ln, name)))
for ct, tt, nc, nm in sorted(rows, reverse=True)[:28]:
    print("%8.3f ms/tick cumulative %8.3f own %6d calls/tick  %s" % (ct / T * 1e3, tt / T * 1e3, nc // T, nm))
