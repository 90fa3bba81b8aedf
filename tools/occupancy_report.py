"""LDS / wavefront occupancy study of BASELINE.json's kinodynamic configuration (config 4: Talos kinodynamic, N = 150, 64 MPC
instances on one MI355X) — and, with --problem full, of the headline full-dynamics configuration.

Part 1  the residency of every kernel of a tick: threads per workgroup, VGPRs, LDS per workgroup, workgroups a CU holds and what
        limits them (LDS: 160 KB per CU; registers: 512 per SIMD lane; 8 wavefront slots per SIMD), wavefronts per SIMD, workgroups
        per launch against the 256 CUs (mpc_kernel_info: hipFuncGetAttributes + the runtime's occupancy calculator).
Part 2  chip-level occupancy: instances per GPU x Riccati legs -> ms per tick, solves/s, per-kernel time.
Part 3  workgroup-size variants of the two big kernels (libraries built with -DEVAL_THREADS=256 / 1024, -DRIC_THREADS=256 under
        mpc_benchmark_amd/csrc/variants/): same table, same ticks.
Part 4  the LDS footprint of the reduced model (nq = 29, what the reference scripts run): same tables.

usage (GPU box): python tools/occupancy_report.py [--problem kino|full] [--batches 16,32,64,128] > profiles/rNN_occupancy_kinodynamic.txt"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem

ap = argparse.ArgumentParser()
ap.add_argument("--problem", default="kino")
ap.add_argument("--batches", default="16,32,64,128")
ap.add_argument("--legs", default="1,4")
ap.add_argument("--ticks", type=int, default=7)  # the synthetic kinodynamic scenario leaves its first double-support phase after ~18 ticks (3 + ticks + 4 are run)
ap.add_argument("--variants", default="eval256,eval1024,ric256")
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LDS_CU, VGPR_SIMD, SLOTS = 160 * 1024, 512, 8


def problem(complete=True):
    if args.problem == "full":
        return FullDynamicsProblem(horizon=100, complete_model=complete), {}
    kp = KinodynamicProblem(horizon=150, complete_model=complete)
    return kp, dict(seed=7, perturb_dofs=range(18, kp.nv))


def ensemble(lib, batch, legs, complete=True):
    pd, kw = problem(complete)
    ens = EnsembleMPC(pd, batch=batch, library=lib, tick_reuse=True, **kw)
    ens.options.riccati_legs = legs
    ens.native.set_options(ens.options)
    ens.prepare_schedule(args.ticks + 8)
    ens.cold_solve(max_iters=100)
    return ens


def residency_table(ens):
    print("%-58s %7s %5s %8s %9s %7s %-9s %6s %10s %7s" % ("kernel", "threads", "VGPRs", "scratch", "LDS/WG", "WG/CU", "limiter", "w/SIMD", "WGs/launch", "rounds"))
    for name, k in ens.native.kernel_info():
        waves = (k["threads"] + 63) // 64
        lds = k["static_lds"] + k["dynamic_lds"]
        by_lds = LDS_CU // lds if lds else 99
        vg = (k["vgprs"] + 7) // 8 * 8
        by_reg = (VGPR_SIMD // max(vg, 8)) * 4 // waves  # wavefronts per SIMD the registers allow, over the 4 SIMDs
        by_slot = SLOTS * 4 // waves
        lim = min((by_lds, "LDS"), (by_reg, "registers"), (by_slot, "wave slots"))
        wg_cu = k["workgroups_per_cu"]
        print("%-58s %7d %5d %8d %9d %7d %-9s %6.1f %10d %7.2f" % (name, k["threads"], k["vgprs"], k["scratch_bytes"], lds, wg_cu, lim[1], wg_cu * waves / 4.0,
                                                          k["workgroups_per_launch"], k["workgroups_per_launch"] / (256.0 * max(1, wg_cu))))


def tick_times(ens, ticks):
    for _ in range(3):
        ens.step()
    lat = []
    for _ in range(ticks):
        t0 = time.perf_counter()
        ens.step()
        lat.append(time.perf_counter() - t0)
    ens.native.profile(2)
    ens.native.profile(1)
    for _ in range(4):
        ens.step()
    ens.native.profile(0)
    per = {k: v[1] / 4 for k, v in ens.native.profile_read().items()}
    return float(np.percentile(np.array(lat) * 1e3, 50)), per


def show(label, ens, ticks):
    p50, per = tick_times(ens, ticks)
    top = sorted(per.items(), key=lambda kv: -kv[1])[:7]
    print("%-34s tick p50 %7.3f ms  %8.1f solves/s | %s" % (label, p50, ens.batch / p50 * 1e3, "  ".join("%s %.2f" % (k.replace("k_", ""), v) for k, v in top)))
    sys.stdout.flush()


lib = _capi.load_hip_library()
title = "kinodynamic N = 150 (nq = 39: n = 76, m = 44)" if args.problem != "full" else "full dynamics N = 100 (nq = 39: n = 76, m = 32)"
print("== Part 1: residency of the kernels of one tick, %s, 64 instances, 4 legs ==" % title)
print("   (one CU: 160 KB LDS, 4 SIMDs x 512 VGPRs per lane x 8 wavefront slots; rounds = workgroups per launch / (256 CUs x WG/CU))")
ens = ensemble(lib, 64, 4)
residency_table(ens)
print("== the same with the serial sweep (legs = 1) ==")
ens1 = ensemble(lib, 64, 1)
residency_table(ens1)
del ens1
print()
print("== Part 2: instances per GPU x legs (ms per tick of the ensemble, kernel times in ms per tick; names as in mpc_profile: closed_loop = k_leg_knot) ==")
for b in [int(x) for x in args.batches.split(",")]:
    for legs in [int(x) for x in args.legs.split(",")]:
        e = ens if (b == 64 and legs == 4) else ensemble(lib, b, legs)
        show("batch %3d, legs %d" % (b, legs), e, args.ticks)
        if e is not ens:
            del e
del ens
print()
print("== Part 3: workgroup-size variants (64 instances, 4 legs) ==")
for v in [x for x in args.variants.split(",") if x]:
    path = os.path.join(ROOT, "mpc_benchmark_amd", "csrc", "variants", "libmpc_hip_%s.so" % v)
    if not os.path.exists(path):
        print("variant %s: %s not built" % (v, path))
        continue
    vl = _capi.bind_library(path)
    try:
        ev = ensemble(vl, 64, 4)
    except Exception as ex:  # a variant that does not fit (LDS, registers) says so
        print("variant %-10s failed: %s" % (v, str(ex)[:160]))
        continue
    print("variant %s:" % v)
    for name, k in ev.native.kernel_info():
        if ("eval_multibody<3>" in name and v.startswith("eval")) or ("riccati" in name and v.startswith("ric")):
            print("   %-58s threads %d VGPRs %d scratch %d LDS %d WG/CU %d" % (name, k["threads"], k["vgprs"], k["scratch_bytes"], k["static_lds"] + k["dynamic_lds"], k["workgroups_per_cu"]))
    show("   %s, batch 64, legs 4" % v, ev, args.ticks)
    del ev
print()
print("== Part 4: reduced model (nq = 29), 64 instances, 4 legs: smaller LDS footprint, same residency ==")
er = ensemble(lib, 64, 4, complete=False)
residency_table(er)
show("reduced model, batch 64, legs 4", er, args.ticks)
