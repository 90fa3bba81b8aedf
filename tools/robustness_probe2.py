"""Developer tool (GPU box): per-knot defects / constraint values of a failing instance around the tick where alpha drops."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi, ensemble as E
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 170
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
watch = [int(v) for v in os.environ.get("WATCH", "0,30,5").split(",")]
pd = FullDynamicsProblem(horizon=100, complete_model=True)
prob = pd.build()
nv = prob.stages[0].xspace.model.nv
x0s = E.ensemble_initial_states(prob.x0_init, prob.stages[0].xspace, batch, perturb_dofs=np.arange(18, nv))
e = E.EnsembleMPC(pd, batch=batch, library=_capi.load_hip_library(), x0=x0s, tick_reuse=bool(int(os.environ.get("REUSE", "1"))))
e.options.riccati_legs = int(os.environ.get("LEGS", "4")); e.native.set_options(e.options)
e.prepare_schedule(ticks + 4)
e.cold_solve(max_iters=100)
N = 100
for t in range(1, ticks + 1):
    try:
        st = e.step()
    except RuntimeError as ex:
        print("tick", t, "error:", str(ex)[-90:]); break
    if t >= int(os.environ.get("FROM", "130")) and t % int(os.environ.get("EVERY", "2")) == 0:
        for b in watch:
            f = np.array([np.max(np.abs(e.native.debug_get("f", k, b))) for k in range(N)])
            c = np.array([np.max(np.abs(np.concatenate((e.native.debug_get("cval", k, b), [0.0])))) for k in range(N + 1)])
            du = np.array([np.max(np.abs(e.native.debug_get("du", k, b))) for k in range(N)])
            s = st[b]
            ls = e.native.debug_get("ls", 0, b)
            print("   ls: phi0 %.6e dphi0 %.3e alpha %.3g | phi(alpha_i) - phi0: %s" % (ls[0], ls[1], ls[2], " ".join("%.3e" % (v - ls[0]) for v in ls[4:])))
            print("tick %3d inst %2d cost %.3e prim %.2e dual %.2e alpha %.3g | max|f| %.2e @%d | max|c| %.2e @%d | max|du| %.2e @%d" % (
                t, b, s.traj_cost, s.prim_infeas, s.dual_infeas, s.alpha, f.max(), f.argmax(), c.max(), c.argmax(), du.max(), du.argmax()))
