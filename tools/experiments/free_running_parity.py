"""Developer probe (GPU box): HIP and the oracle FREE-RUNNING from the same cold solve — no reset of the HIP handle to the oracle's state between ticks —
how the deviation of xs / us / K_0 grows tick over tick.  args: ticks horizon ; env CORRECTOR, REFINE, MODEL=reduced|complete"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from tests import _oracle
from tests._metrics import rel_cols
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
def handle(lib):
    e = EnsembleMPC(FullDynamicsProblem(horizon=N, complete_model=os.environ.get("MODEL", "reduced") == "complete"), batch=2, library=lib, seed=3, sigma_q=float(os.environ.get("SIGMA_Q", "0.004")), sigma_v=float(os.environ.get("SIGMA_V", "0.01")))
    e.options.num_threads = os.cpu_count() or 8
    e.options.riccati_legs = int(os.environ.get("LEGS", "1")) if lib is not _ORACLE else 1
    e.options.corrector_prim_tol = float(os.environ.get("CORRECTOR", "0"))
    e.options.refine_appended_knot = int(os.environ.get("REFINE", "0"))
    e.native.set_options(e.options)
    e.prepare_schedule(ticks + 8)
    e.cold_solve(max_iters=100)
    return e
_ORACLE = _oracle.load()
er, eh = handle(_ORACLE), handle(_capi.load_hip_library())
for t in range(ticks):
    sr, sh = er.step(), eh.step()
    a, b = eh.results(gains=True), er.results(gains=True)
    e = max(rel_cols(a["xs"], b["xs"], 1e-3), rel_cols(a["us"], b["us"], 1.0), rel_cols(a["K"][:, 0], b["K"][:, 0], 1.0))
    print("tick %3d deviation %.3e | alpha hip %s oracle %s iters %s prim %.2e" % (t, e, [s.alpha for s in sh], [s.alpha for s in sr], [s.num_iters for s in sh], max(s.prim_infeas for s in sr)), flush=True)
