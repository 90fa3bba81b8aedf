"""Developer tool: where the host time of one ensemble tick goes (cycle upload, setup, launch + wait)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=64, library=lib)
ens.prepare_schedule(60)
ens.cold_solve(100)
for _ in range(5): ens.step()
acc = {"table": 0.0, "cycle": 0.0, "setup": 0.0, "run": 0.0}
T = 30
for _ in range(T):
    t0 = time.perf_counter(); d, p = ens._table_for_tick(ens.tick % pd.t_mpc)
    t1 = time.perf_counter(); ens.native.cycle(d, p)
    t2 = time.perf_counter(); ens.native.setup()
    t3 = time.perf_counter(); ens.native.run_shifted(); ens.tick += 1
    t4 = time.perf_counter()
    acc["table"] += t1 - t0; acc["cycle"] += t2 - t1; acc["setup"] += t3 - t2; acc["run"] += t4 - t3
print({k: round(v / T * 1e3, 3) for k, v in acc.items()}, "ms per tick")
