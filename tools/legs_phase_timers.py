"""Developer tool: in-kernel phase timers of k_leg_consensus (shader clock, 2.4 GHz assumed; ratios are what counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

legs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lib = _capi.load_hip_library()
pd = FullDynamicsProblem(horizon=100, complete_model=True)
ens = EnsembleMPC(pd, batch=int(sys.argv[2]) if len(sys.argv) > 2 else 4, library=lib)
ens.options.riccati_legs = legs
ens.native.set_options(ens.options)
ens.prepare_schedule(10)
ens.cold_solve(100)
ens.native.profile(3)
ens.native.debug_get('ric_prof', 0)
TICKS = 3
for _ in range(TICKS):
    ens.step()
p = ens.native.debug_get('ric_prof', 0)
pk = ens.native.debug_get('ric_prof', 0, b=1)[32:]
pc = ens.native.debug_get('ric_prof', 0, b=2)[32:]
GHZ = 2.4
names = {23: 'loads (Sg, Lm^T, old calP), dP', 24: 'rv, Mt = I - Sg dP', 25: 'Gauss-Jordan', 26: 'reorder, Zx / zc out', 27: 'D = dP Zx, ev',
         28: 'calP_j = P_j + Lm_j D (or K0)', 29: 'forward over the cuts'}
cuts = (legs - 1) * TICKS
tot = sum(p[i] for i in names)
for i, nm in names.items():
    print('CONS %-40s %7.1f us/cut %5.1f%%' % (nm, p[i] / cuts / (GHZ * 1e3), 100 * p[i] / tot))
print('CONS total %.1f us per cut' % (tot / cuts / (GHZ * 1e3)))

q = ens.native.debug_get('ric_prof', 0, b=1)[32:] if False else None

kn = {0: 'loads: [A B], K to LDS, Pt to registers', 1: 'A + B K, y0, Pt to LDS', 2: '(I - mu_d Pt)[Acl | B], in place', 3: 'T6 rows, Phi / phi out, Pt T^T',
      4: 'Mu, Znu in, T Pt T^T', 5: 'U1 = Mu Bc^T, Knup, Ku out', 6: 'Gamma'}
if os.environ.get("LK_SUB"):  # a library built with -DLK_SUBPROF: stage 4 in pieces (their time comes off slot 5)
    kn.update({7: '  4a: U1 tiles (MFMA)', 8: '  4b: Knup tiles', 9: '  4c: U1 -> LDS, Ku out'})
tk = sum(pk[i] for i in kn)
for i, nm in kn.items():
    print('KNOT %-44s %7.1f us %5.1f%%' % (nm, pk[i] / TICKS / (GHZ * 1e3), 100 * pk[i] / tk))
print('KNOT total %.1f us per workgroup (knot 1, parametric)' % (tk / TICKS / (GHZ * 1e3)))
cn = {0: 'Phi, Gamma registers -> LDS, barrier', 1: 'Lm = Phi^T Lm, Z = Gamma Lm (tiles)', 2: 'sg += Lm^T phi', 3: 'barrier', 4: 'Z -> LDS, barrier',
      5: 'Sg += Lm^T Z (tiles)', 6: 'barrier', 7: 'Lm -> LDS and gain record, barrier'}
knots = (100 // legs) * TICKS
tcn = sum(pc[i] for i in cn)
for i, nm in cn.items():
    print('COND %-44s %7.1f us/knot %5.1f%%' % (nm, pc[i] / knots / (GHZ * 1e3), 100 * pc[i] / tcn))
print('COND total %.1f us per knot' % (tcn / knots / (GHZ * 1e3)))
