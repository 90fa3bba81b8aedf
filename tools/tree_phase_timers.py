"""Developer tool (GPU box): phase timers of k_leg_compose (role 0 of the first composition of the first level, batch 1 at 32 legs):
loads, Mt = I - Sg D, elimination, the products after it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
ens = EnsembleMPC(FullDynamicsProblem(horizon=100, complete_model=True), batch=1, library=_capi.load_hip_library(), perturb=False, tick_reuse=True)
ens.options.riccati_legs = 32
ens.native.set_options(ens.options)
ens.prepare_schedule(80)
ens.cold_solve(100)
ens.native.profile(3)
ens.native.debug_get("ric_prof", 0)
T = 50
for _ in range(T):
    ens.step()
p = ens.native.debug_get("ric_prof", 0)
names = {24: "loads (Sg_a, P_b, guess), D", 25: "rv, Mt = I - Sg_a D, Lm_a^T in", 26: "elimination", 27: "Lm_b in, products, node record out"}
tot = sum(p[i] for i in names)
for i, nm in names.items():
    print("COMPOSE %-36s %6.1f us %5.1f%%" % (nm, p[i] / T / 2.4e3, 100 * p[i] / tot))
print("COMPOSE total %.1f us per composition (role 0) ; of the elimination, the owners' serial pieces (panel and inverse of the pivot block out): %.1f us" % (tot / T / 2.4e3, p[28] / T / 2.4e3))
