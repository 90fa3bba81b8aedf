import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
lib = _capi.load_hip_library()
for name, pdf in (("full dynamics nq=39", lambda: FullDynamicsProblem(horizon=100, complete_model=True)), ("centroidal", lambda: CentroidalProblem(horizon=100))):
    for legs in (4, 8, 12, 16, 24, 32):
        pd = pdf()
        one = EnsembleMPC(pd, batch=1, library=lib, perturb=False, tick_reuse=True)
        one.options.riccati_legs = legs
        one.native.set_options(one.options)
        one.prepare_schedule(60)
        one.cold_solve(max_iters=100)
        lat = []
        for i in range(45):
            one.results(gains=False)
            ts = time.perf_counter()
            one.step()
            one.results(gains=False)
            if i >= 5:
                lat.append((time.perf_counter() - ts) * 1e3)
        print("%s legs %2d: p50 %.3f ms" % (name, legs, np.percentile(lat, 50)))
