"""Developer tool (GPU box): tick times of BASELINE config 4 (kinodynamic STAIRS, N = 150, 64 instances, complete model, 4 legs, tick
reuse) — p50 / p90, how many ticks take more than one pass (a BCL update without a step, then the step), per-kernel time.
The walk of the script (0.3 m steps, kinodynamic_talos.py:257) with 0.10 m gained per step, references replanned every tick
(EnsembleMPC.enable_walk); STAIRS=0: flat ground, WALK=0: frozen references (the round-3 form of this measurement), PERINST=1:
every instance plans from its own foot poses, REFINE=R: mpc_options.refine_appended_knot (negative: after every cycle).  START=n: n untimed ticks first (120: the first swing is at knot 0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.bind_library(os.environ["LIB"]) if os.environ.get("LIB") else _capi.load_hip_library()
kp = KinodynamicProblem(horizon=150, complete_model=True)
ens = EnsembleMPC(kp, batch=64, library=lib, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=not os.environ.get("NO_REUSE"))
ens.options.riccati_legs = int(os.environ.get("LEGS", "4"))
ens.options.refine_appended_knot = int(os.environ.get("REFINE", "0"))  # (mpc_options.refine_appended_knot: Newton steps on the control of the appended knot when the contact pattern changes)
ens.native.set_options(ens.options)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ens.prepare_schedule(T + 10 + int(os.environ.get("START", "0")))
ens.cold_solve(max_iters=100)
if int(os.environ.get("WALK", "1")):
    ens.enable_walk(z_height=(0.10 if int(os.environ.get("STAIRS", "1")) else 0.0), per_instance=bool(int(os.environ.get("PERINST", "0"))))
for _ in range(3 + int(os.environ.get("START", "0"))): ens.step()
ens.native.profile(2); ens.native.profile(1)
lat, al, ls = [], [], []
for _ in range(T):
    t0 = time.perf_counter()
    st = ens.step()
    lat.append((time.perf_counter() - t0) * 1e3)
    al.append(sum(1 for s in st if s.al_iters > 0))
    ls.append(np.bincount([int(s.ls_steps) for s in st], minlength=8))
ens.native.profile(0)
lat = np.array(lat)
print("kinodynamic N=150 B=64 %s legs %d: p50 %.2f ms  p90 %.2f ms  mean %.2f ms | ticks in which some instance took a BCL update: %d of %d (instances per such tick: mean %.1f)" % (
    ("frozen references" if ens._walk is None else "walk, z_height %.2f%s, replanning ticks %d" % (ens._walk_args["z_height"], ", per-instance references" if ens._walk_args["per_instance"] else "", getattr(ens, "replanning_ticks", 0))),
    ens.options.riccati_legs, np.percentile(lat, 50), np.percentile(lat, 90), lat.mean(), sum(1 for a in al if a), T, np.mean([a for a in al if a] or [0])))
for t, (l, h) in enumerate(zip(lat, ls)):
    if h[1:].sum():
        print("  tick %2d %.1f ms: instances by accepted backtracking step (0 = full step .. 7): %s" % (t, l, h.tolist()))
print("  tick times (ms):", " ".join("%.1f" % v for v in lat))
for k, (c, ms) in sorted(ens.native.profile_read().items(), key=lambda kv: -kv[1][1])[:9]:
    print("  %-26s launches/tick %.2f  ms/tick %.3f" % (k, c / T, ms / T))
