"""Developer tool: per-kernel time of the centroidal OCP (BASELINE.json config 1: N = 100, batch 1) and of a 64-instance ensemble."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
lib = _capi.load_hip_library()
for B in (1, 64):
    pd = CentroidalProblem(horizon=100)
    ens = EnsembleMPC(pd, batch=B, library=lib, perturb=False)
    ens.prepare_schedule(60)
    st = ens.cold_solve(max_iters=100)
    for _ in range(5): ens.step()
    ens.native.profile(2); ens.native.profile(1)
    ens.results(gains=False)
    t0 = time.perf_counter()
    for _ in range(40): ens.step()
    ens.results(gains=False)
    dt = (time.perf_counter() - t0) / 40
    ens.native.profile(0)
    print("centroidal N=100 B=%d: cold iters %d conv %s | %.3f ms per tick (%.0f solves/s)" % (B, st[0].num_iters, bool(st[0].converged), dt * 1e3, B / dt))
    for k, (c, ms) in sorted(ens.native.profile_read().items(), key=lambda kv: -kv[1][1])[:6]:
        print("    %-24s %8.4f ms per tick" % (k, ms / 40))
