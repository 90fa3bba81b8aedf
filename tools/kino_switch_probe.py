"""Developer probe (GPU box): the ticks of the kinodynamic walk in which EVERY instance backtracks (profiles/r04_kino_tick.txt: ticks 15 - 20
after the first take-off reaches the front of the horizon).  One nominal instance, N = 150, complete model: accepted step per tick, the
merit of the candidates, and the knots whose merit grows along the full step.  usage: python tools/kino_switch_probe.py [ticks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
lib = _capi.bind_library(os.environ["LIB"]) if os.environ.get("LIB") else _capi.load_hip_library()
N = int(os.environ.get("HORIZON", "150"))
kp = KinodynamicProblem(horizon=N, complete_model=bool(int(os.environ.get("COMPLETE", "1"))))
ens = EnsembleMPC(kp, batch=1, library=lib, perturb=False, tick_reuse=not os.environ.get("NO_REUSE"))
ens.options.riccati_legs = int(os.environ.get("LEGS", "4"))
ens.options.refine_appended_knot = int(os.environ.get("REFINE", "0"))
ens.native.set_options(ens.options)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ens.prepare_schedule(T + 8)
st = ens.cold_solve(max_iters=100)
print("cold solve: %d iterations, converged %s" % (st[0].num_iters, bool(st[0].converged)))
if int(os.environ.get("WALK", "1")):
    ens.enable_walk(z_height=0.0)
for t in range(T):
    st = ens.step()[0]
    ls = ens.native.debug_get("ls", 0)
    phi0, dphi0, alpha, nstep = ls[:4]
    cand = ls[4:]
    print("tick %2d: alpha %.5f (%d halvings)  phi0 %.6e  dphi0 %.3e  phi(1) - phi0 %+.3e  phi(1/2) - phi0 %+.3e | prim %.2e dual %.2e" % (
        t, alpha, int(nstep), phi0, dphi0, cand[0] - phi0, (cand[1] - phi0) if nstep >= 1 else float("nan"), st.prim_infeas, st.dual_infeas))
    if nstep >= 1:
        rows = []
        for k in range(N + 1):
            v = ens.native.debug_get("ls_knot", k)
            rows.append((v[0] - (v[-2] + v[-1]), k, v[0], v[-2], v[-1]))
        rows.sort(reverse=True)
        print("   knots whose merit grows most along the full step (trial - current: knot, trial, cost, penalty):")
        for d, k, tr, c, pn in rows[:6]:
            print("      knot %3d  %+.3e   trial %.4e  cost %.4e  penalty %.4e" % (k, d, tr, c, pn))
        dxn = [float(np.max(np.abs(ens.native.debug_get("dx", k)))) for k in range(N + 1)]
        dun = [float(np.max(np.abs(ens.native.debug_get("du", k)))) for k in range(N)]
        print("   max|dx_k| knots 0..5: %s ... 145..150: %s" % (" ".join("%.1e" % v for v in dxn[:6]), " ".join("%.1e" % v for v in dxn[-6:])))
        print("   max|du_k| knots 0..5: %s ... 144..149: %s" % (" ".join("%.1e" % v for v in dun[:6]), " ".join("%.1e" % v for v in dun[-6:])))
        kb = int(np.argmax(dun)); du = ens.native.debug_get("du", kb)
        top = np.argsort(-np.abs(du))[:6]
        print("   largest control step at knot %d: components %s = %s  (0..11: wrenches LF, RF ; 12..: joint accelerations)" % (kb, top.tolist(), " ".join("%+.2e" % du[i] for i in top)))
        cv = ens.native.debug_get("cval", N - 1); act = ens.native.debug_get("act", N - 1)
        print("   knot %d at the current point: %d constraint rows, %d active, largest |value| among the active %.2e" % (N - 1, cv.size, int(np.count_nonzero(act)), float(np.max(np.abs(cv[act != 0]))) if np.any(act != 0) else 0.0))
