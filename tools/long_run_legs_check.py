"""Developer check (GPU): a long run of the bench ensemble with the parallel-in-time sweep (tree over the cuts, value-function guesses
carried from tick to tick) against the serial sweep: same costs / feasibility tick by tick, trajectories equal to round-off growth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 110
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))
MARKS = sorted({t for t in (1, 5, 20, 55) if t < T} | {T})
runs = {}
for legs in (1, 4, 32):
    pd = FullDynamicsProblem(horizon=100, complete_model=True)
    ens = EnsembleMPC(pd, batch=B, library=lib, seed=20250304, tick_reuse=True)
    ens.options.riccati_legs = legs
    ens.native.set_options(ens.options)
    ens.prepare_schedule(T + 5)
    st = ens.cold_solve(max_iters=100)
    rec = [(ens.results(gains=False)["xs"].copy(), [int(s.num_iters) for s in st])]
    for t in range(1, T + 1):
        st = ens.step()
        if t in MARKS:
            rec.append((ens.results(gains=False)["xs"].copy(), np.array([s.traj_cost for s in st]), np.array([s.prim_infeas for s in st]), min(s.alpha for s in st)))
    runs[legs] = rec
same = [i for i in range(B) if runs[1][0][1][i] == runs[4][0][1][i] == runs[32][0][1][i]]
print("instances whose cold solves took the same number of iterations in all three runs: %d of %d" % (len(same), B))
for legs in (4, 32):
    print("legs %2d vs serial:" % legs)
    for (ta, a), b in zip(zip(MARKS, runs[legs][1:]), runs[1][1:]):
        print("   tick %3d: xs rel err %.2e | cost med %.3f vs %.3f | prim max %.2e vs %.2e | alpha min %.3g vs %.3g" % (
            ta, rel(a[0][same], b[0][same]), np.median(a[1]), np.median(b[1]), a[2].max(), b[2].max(), a[3], b[3]))
