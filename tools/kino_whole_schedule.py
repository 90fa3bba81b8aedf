"""Developer tool (GPU box): BASELINE configuration 4 (kinodynamic stairs, N = 150, 64 instances, complete model) over the script's whole
schedule with ONE iteration per tick — instances lost / revived, backtracking ticks, time — for a value of mpc_options.refine_appended_knot
(REFINE, default -1: the control of the appended knot refined after every cycle).  usage: python tools/kino_whole_schedule.py [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.load_hip_library()
kp = KinodynamicProblem(horizon=150, complete_model=True)
ens = EnsembleMPC(kp, batch=64, library=lib, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=True)
ens.options.riccati_legs = 4
ens.options.refine_appended_knot = int(os.environ.get("REFINE", "-1"))
ens.native.set_options(ens.options)
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else kp.t_mpc - 1
ens.prepare_schedule(ticks + 4)
st = ens.cold_solve(max_iters=100)
ens.enable_walk(z_height=float(os.environ.get("Z", "0.10")))
ens.enable_failure_isolation(auto_revive=True, source=0)
back, t0, lat = [], time.time(), []
for t in range(ticks):
    t1 = time.perf_counter()
    st = ens.step()
    lat.append((time.perf_counter() - t1) * 1e3)
    nb = sum(1 for s in st if s.ls_steps > 0)
    if nb: back.append((t, nb, max(int(s.ls_steps) for s in st)))
r = ens.results(gains=False)
base = r["xs"][:, 0, :3] - kp.robot.x0[:3]
lat = np.array(lat)
print("refine_appended_knot %d, z_height %s, %d ticks, one iteration per tick: lost and revived %s ; ticks with backtracking %d (instances x halvings, first ten: %s) ; tick p50 %.2f p90 %.2f max %.2f ms ; base x %.3f .. %.3f z %.3f .. %.3f ; %.1f s" % (
    ens.options.refine_appended_knot, os.environ.get("Z", "0.10"), ticks, [(t, b, c) for t, b, c, _ in ens.lost], len(back), back[:10], np.percentile(lat, 50), np.percentile(lat, 90), lat.max(),
    base[:, 0].min(), base[:, 0].max(), base[:, 2].min(), base[:, 2].max(), time.time() - t0))
