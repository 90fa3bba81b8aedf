"""Developer probe (GPU): the tree over the cuts (riccati_legs > 8, csrc/legs_tree.h) against the serial sweep and the chain consensus."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem

lib = _capi.load_hip_library()
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))


def run(pd, legs, batch=2, ticks=6, **kw):
    ens = EnsembleMPC(pd, batch=batch, library=lib, tick_reuse=True, **kw)
    ens.options.riccati_legs = legs
    ens.native.set_options(ens.options)
    ens.prepare_schedule(ticks + 4)
    st = ens.cold_solve(max_iters=100)
    r = ens.results()
    out = [(r["xs"].copy(), r["us"].copy(), r["K"][:, 0].copy(), [int(s.num_iters) for s in st])]
    lat = []
    for _ in range(ticks):
        t0 = time.perf_counter(); ens.step(); lat.append(time.perf_counter() - t0)
        r = ens.results()
        out.append((r["xs"].copy(), r["us"].copy(), r["K"][:, 0].copy(), None))
    return out, float(np.median(lat) * 1e3)


for name, mk, kw in (("centroidal N=100", lambda: CentroidalProblem(horizon=100), dict(perturb=False)),
                     ("full dynamics N=100 complete", lambda: FullDynamicsProblem(horizon=100, complete_model=True), {})):
    ref, t1 = run(mk(), 1, **kw)
    print("%s: serial %.3f ms per tick, cold iters %s" % (name, t1, ref[0][3]))
    for legs in (8, 12, 16):
        res, tl = run(mk(), legs, **kw)
        ex = max(rel(r[0], q[0]) for r, q in zip(res, ref)); eu = max(rel(r[1], q[1]) for r, q in zip(res, ref))
        ek = max(rel(r[2], q[2]) for r, q in zip(res, ref))
        print("   legs %2d: %.3f ms per tick | max rel err over cold solve + ticks: xs %.2e us %.2e K0 %.2e | cold iters %s" % (legs, tl, ex, eu, ek, res[0][3]))
