"""Developer tool: in-kernel phase timers (shader clock, ~2.4 GHz assumed) of the Riccati sweep and of the whole-body
stage kernel on the default bench workload.  usage: python tools/phase_timers.py [path/to/libmpc_hip.so] [centroidal]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

lib = _capi.bind_library(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else _capi.load_hip_library()
if "kino" in sys.argv[1:]:
    from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
    pd = KinodynamicProblem(horizon=100, complete_model=True)
elif "centroidal" in sys.argv[1:]:
    from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
    pd = CentroidalProblem(horizon=100)
else:
    pd = FullDynamicsProblem(horizon=100, complete_model=True)
LEGS = int(os.environ.get("PHASE_LEGS", "1"))  # > 1: the workgroup of leg 0 is timed (a parametric leg: N / legs knots per sweep)
BATCH = int(os.environ.get("PHASE_BATCH", "4"))
ens = EnsembleMPC(pd, batch=BATCH, library=lib, **({'perturb_dofs': range(18, pd.nv), 'seed': 7} if 'kino' in sys.argv[1:] else {}))
ens.options.riccati_legs = LEGS  # 1: the serial sweep, one workgroup walks all knots (tools/legs_phase_timers.py times the other leg kernels)
ens.native.set_options(ens.options)
ens.prepare_schedule(10)
ens.cold_solve(100)
ens.native.profile(3)  # in-kernel phase timers on
ens.native.debug_get('ric_prof', 0)  # read + clear
TICKS = 3
for _ in range(TICKS):
    ens.step()
p = ens.native.debug_get('ric_prof', 0)
GHZ = 2.4
ric = {0: 'T6 inverse, active-row scan', 1: 'T6 similarity transform of P', 2: 'Ph copy, Frobenius norm, vv', 17: 'series length', 18: '[A B] prefetch issue', 13: 'small-vector prefetch issue',
       14: 'series: tile products', 15: 'series: barrier wait', 16: 'series: in-place update', 3: 'chol n (fallback only)',
       5: 'triangular solves (fallback only)', 6: 'AB prefetch issue, w, store Pt', 7: 'AB to LDS, gh', 8: 'G = Pt [A B], Hh = H + [A B]^T G',
       9: 'KKT operands', 10: 'chol m', 11: 'KKT solve (active rows), gains', 12: 'value function, store gain record'}
if os.environ.get("PHASE_SUB"):  # a build with -DRIC_SUBPROF (and -DRIC_PROF_TID=64 for the view of wavefront 1): the time up to each mark comes off the phase that contains it
    ric.update({23: '  0a: prefetched values to LDS, barrier', 24: '  6a: w rows, Pt record', 26: '  8a: G tiles (MFMA)', 27: '  8b: barrier, G to LDS, barrier',
                28: '  8c: Ruu tiles, two barriers', 29: '  8d: chol Ruu (wave 0) | Sh^T and x tiles of Hh', 30: '  11a: forward solves', 31: '  11b: active rows (Schur complement)',
                4: '  11c: backward solves', 19: '  12a: gains out, p', 25: '  12b: P tiles'})
knots = (100 // LEGS) * TICKS
tot = sum(p[i] for i in ric)
for i, name in ric.items():
    print('RIC  %-40s %7.1f us/knot %5.1f%%' % (name, p[i] / knots / (GHZ * 1e3), 100 * p[i] / tot))
print('RIC  total %.1f us/knot ; series: max rho %.3e, mean terms %.2f, Cholesky fallbacks per knot %.3f' % (
    tot / knots / (GHZ * 1e3), p[20], p[21] / knots, p[22] / knots))
ev = {0: 'load, FK, joint columns', 1: 'velocities, inertias, composites, U', 7: 'derivative pre-pass (Psd, Phi, B_i, Bc, Bt, Tv)',
      2: 'M tiles, bias, contact frames, Y16', 5: 'chol M, Y, S, multipliers, accelerations (one wave)', 6: 'xdot / wrench record, contact wrenches',
      3: 'forces at the solution, Psdd, Tq', 8: 'contact rows R2', 9: 'R1 in registers + implicit differentiation',
      10: 'SE(3) pre-pass, integrator, [A B]', 29: 'term table: classification, accumulator reset',
      11: 'dense-weight terms, cost sum', 12: 'merit, projections'}
terms = {1: 'state_error', 2: 'control_error', 3: 'frame_placement', 4: 'frame_translation', 5: 'frame_velocity', 6: 'com_translation',
         7: 'centroidal_momentum', 8: 'contact_force', 9: 'mb_wrench_cone', 10: 'centroidal_wrench_cone', 13: 'centroidal_momentum_der'}
e = p[32:]
etot = sum(e[i] for i in ev) + sum(e[13 + t] for t in terms) + e[27] + e[28]
for i, name in ev.items():
    print('EVAL %-40s %7.1f us %5.1f%%' % (name, e[i] / TICKS / (GHZ * 1e3), 100 * e[i] / etot))
for t, name in terms.items():
    if e[13 + t] > 0:
        print('EVAL term %-35s %7.1f us %5.1f%%' % (name, e[13 + t] / TICKS / (GHZ * 1e3), 100 * e[13 + t] / etot))
if os.environ.get("PHASE_SUB"):  # a build with -DEV_SUBPROF: the one-wavefront solve in pieces (slots of terms this problem does not have)
    for i, name in ((3, 'sub: chol M'), (4, 'sub: Y = L^-1 [Jc^T | r1]'), (16, 'sub: S = Y^T Y, chol S'), (17, 'sub: multipliers'), (18, 'sub: V16 build'), (19, 'sub: accelerations')):
        print('EVAL %-40s %7.1f us' % (name, e[i] / TICKS / (GHZ * 1e3)))
if os.environ.get("PHASE_P11"):  # a build with -DEV_P11PROF: P11 per wavefront
    for i, name in ((16, 'P11 wave 0: its column blocks'), (17, 'P11 wave 1'), (18, 'P11 wave 2'), (19, 'P11 wave 3 (one block less)'), (3, 'P11 wave 3: SE(3) work of the integrator'), (4, 'P11 wave 0: R1 of its first block')):
        print('EVAL %-40s %7.1f us' % (name, e[i] / TICKS / (GHZ * 1e3)))
print('EVAL %-40s %7.1f us %5.1f%%' % ('stacked cost terms (one per wavefront)', e[28] / TICKS / (GHZ * 1e3), 100 * e[28] / etot))
print('EVAL %-40s %7.1f us %5.1f%%' % ('Gauss-Newton Hessian flush (MFMA)', e[27] / TICKS / (GHZ * 1e3), 100 * e[27] / etot))
print('EVAL total %.1f us per workgroup (knot 1 of instance 0)' % (etot / TICKS / (GHZ * 1e3)))
