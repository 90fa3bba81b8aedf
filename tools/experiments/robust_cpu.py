"""Developer probe (build container, no GPU): the benchmarked ensemble's walk on ONE ProxDDP iteration per tick on the CPU port
(oracle/cpu_port, the oracle's solver with closed-form stage evaluation) — which instances are lost, and when, under a given
globalisation setting.  Usage: robust_cpu.py [ticks] [batch] ; env: MODEL=complete|reduced, REFS=frozen|shared|instance, REFINE, THREADS,
WATCH=<instance> (per-tick line for it), N (horizon)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests import _cpu_port
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = _cpu_port.load()
refs = os.environ.get("REFS", "frozen")
pd = FullDynamicsProblem(horizon=int(os.environ.get("N", "100")), complete_model=os.environ.get("MODEL", "complete") == "complete")
(e,) = make_bench_shards(pd, lib, batch, legs=1, tick_reuse=False)
e.options.refine_appended_knot = int(os.environ.get("REFINE", "0"))
e.options.num_threads = int(os.environ.get("THREADS", "8"))
for kv in os.environ.get("OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("=")
        setattr(e.options, k, type(getattr(e.options, k))(float(v)))
e.native.set_options(e.options)
e.prepare_schedule(pd.t_mpc + 4)
t0 = time.time()
st = e.cold_solve(max_iters=int(os.environ.get("COLD", "100")))
print("cold solve: iters %s (%.1f s)" % ([s.num_iters for s in st], time.time() - t0), flush=True)
e.enable_failure_isolation(auto_revive=False)
if refs != "frozen":
    e.enable_walk(per_instance=(refs == "instance"))
watch = int(os.environ.get("WATCH", "-1"))
lost = {}
nback = 0
niter = 0
for t in range(ticks):
    st = e.step()
    for b, s in enumerate(st):
        if s.converged < 0 and b not in lost:
            lost[b] = t
            print("   instance %d lost at tick %d" % (b, t), flush=True)
    alive = [s for b, s in enumerate(st) if b not in lost]
    nback += sum(1 for s in alive if s.alpha < 1.0)
    niter += sum(s.num_iters for s in alive)
    if watch >= 0 and watch not in lost:
        s = st[watch]
        print("tick %3d inst %d cost %.4e prim %.2e dual %.2e alpha %.4g ls %d" % (t, watch, s.traj_cost, s.prim_infeas, s.dual_infeas, s.alpha, s.ls_steps), flush=True)
    elif t % 50 == 49:
        print("tick %d: %d alive, worst prim %.2e, min alpha %.3g (%.0f s)" % (t, len(alive), max([s.prim_infeas for s in alive] + [0]), min([s.alpha for s in alive] + [1]), time.time() - t0), flush=True)
    if len(lost) == batch:
        break
print("%s refs, refine %d, %s: %d of %d lost over %d ticks: %s ; backtracking instance-ticks %d ; iterations %d ; %.0f s" % (
    refs, e.options.refine_appended_knot, os.environ.get("OPTS", ""), len(lost), batch, ticks, lost, nback, niter, time.time() - t0))
