"""Developer tool (GPU box): which settings let the benchmarked ensemble (64 randomised instances, N = 100, complete model, 4 legs, tick reuse,
two ticks in flight) walk the reference's whole 1000-tick schedule on ONE ProxDDP iteration per tick — instances lost (isolated and re-seeded
from the nominal one) per combination of {references: frozen | walk shared | walk per instance} x {refine_appended_knot: 0 | 3}."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import make_bench_shards
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 999
for refs in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("frozen", "shared", "instance")):
    for R in [int(v) for v in os.environ.get("REFINES", "0,3").split(",")]:
        pd = FullDynamicsProblem(horizon=100, complete_model=True)
        closed = (10, pd.dt / 10) if os.environ.get("CLOSED") else None  # CLOSED=1: measured states from the simulation stand-in (mpc_simulate: 10 x 1 ms of knot 0's contact dynamics under u = us[0] - K_0 difference(x, xs[0])) instead of perfect-model feedback
        (e,) = make_bench_shards(pd, lib, 64, legs=4, tick_reuse=True, seed=int(os.environ.get("SEED", "20250304")), closed_loop=closed)
        e.options.refine_appended_knot = R
        e.options.corrector_prim_tol = float(os.environ.get("CORRECTOR", "20"))
        e.options.corrector_window = int(os.environ.get("WINDOW", "0"))
        e.native.set_options(e.options)
        e.prepare_schedule(pd.t_mpc + 4)
        e.cold_solve(max_iters=100)
        e.enable_failure_isolation(auto_revive=True, source=0)
        if refs != "frozen":
            e.enable_walk(per_instance=(refs == "instance"), generator=("device" if refs == "instance" and os.environ.get("GENERATOR", "host") == "device" else "host"), floor=(refs == "instance" and bool(os.environ.get("FLOOR"))))  # GENERATOR=device: mpc_walk_* (bench.py's path) ; FLOOR=1: the measured soles stay on the floor
        t0 = time.time(); worst = 0.0; nominal_lost = False; extra = 0; extra_ticks = 0; back = 0
        for t in range(ticks):
            e.step_async()
            if e.inflight == 2:
                st = e.wait()
                worst = max([worst] + [s.prim_infeas for s in st if s.converged >= 0])
                nx = sum(1 for s in st if s.num_iters > 1); extra += nx; extra_ticks += 1 if nx else 0
                back += sum(1 for s in st if s.alpha < 1.0 and s.converged >= 0)
                if st[0].converged < 0 and not nominal_lost:
                    nominal_lost = True; print("   (the nominal instance failed at tick %d)" % t, flush=True)
        while e.inflight:
            e.wait()
        print(("closed loop (simulated measurements), " if closed else "") + "references %-8s refine_appended_knot %d: %3d instance losses in %d ticks (first at tick %s; instances %s) ; largest primal infeasibility before a step %.2e ; corrector %g: %d instance-ticks on %d ticks, %d backtracking instance-ticks ; %.1f s" % (
            refs, R, len(e.lost), ticks, e.lost[0][0] if e.lost else "-", sorted(set(r[1] for r in e.lost))[:12], worst, e.options.corrector_prim_tol, extra, extra_ticks, back, time.time() - t0), flush=True)
        del e
