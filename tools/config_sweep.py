"""The other configurations of BASELINE.json beside the bench line: solves/s and p50 ms per solve of one
warm-started MPC tick (one ProxDDP iteration, perfect-model feedback), HIP library only.
  config 2  centroidal walk, N = 100, batch 1
  config 3  full dynamics, N = 100, batch 1, complete (nq = 39) and reduced (nq = 29) model
  config 4  kinodynamic, N = 150, batch 64
usage: python tools/config_sweep.py  > profiles/rNN_other_configs.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem

lib = _capi.load_hip_library()


def run(name, pd, batch, ticks, warm, legs=1, walk=None, **kw):
    """``walk``: keyword arguments of EnsembleMPC.enable_walk (the loop body's reference updates every tick; {} = the script's own step length) or
    None (frozen references)."""
    ens = EnsembleMPC(pd, batch=batch, library=lib, tick_reuse=True, **kw)  # (whole-body problems only; ignored for the centroidal one)
    ens.options.riccati_legs = legs  # parallel-in-time Riccati (csrc/legs.h); 1 = serial sweep
    ens.native.set_options(ens.options)
    name = "%s, legs %d" % (name, legs)
    ens.prepare_schedule(ticks + warm + 4)
    st = ens.cold_solve(max_iters=100)
    if walk is not None:
        ens.enable_walk(**walk)
        name += ", walk" + ("".join(" %s=%s" % kv for kv in walk.items()))
    for _ in range(warm):
        ens.step()
    lat = []
    for _ in range(ticks):
        t0 = time.perf_counter()
        ens.step()
        lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e3
    d = ens.dims
    print("%-56s n=%2d m=%2d c<=%3d | cold %3d it, %3d/%d converged | tick p50 %7.3f ms  p90 %7.3f ms | %8.1f solves/s" % (
        name, d.ndx, d.nu, d.nc_max, max(int(s.num_iters) for s in st), sum(bool(s.converged) for s in st), batch,
        np.percentile(lat, 50), np.percentile(lat, 90), batch / np.mean(lat) * 1e3))
    sys.stdout.flush()


for legs in (1, 4, 8, 16, 32):
    run("config 2: centroidal N=100 batch 1", CentroidalProblem(horizon=100), 1, 180, 130, legs=legs, perturb=False, walk={})  # timed ticks 130 - 310: the first swing
run("config 2': centroidal N=100 batch 64", CentroidalProblem(horizon=100), 64, 60, 10, legs=4, perturb=False)
for legs in (1, 4, 8, 16, 32):
    run("config 3: full dynamics N=100 batch 1 (nq=39)", FullDynamicsProblem(horizon=100, complete_model=True), 1, 100, 20, legs=legs, perturb=False)
for legs in (1, 4, 8, 16, 32):
    run("config 3: full dynamics N=100 batch 1 (nq=29)", FullDynamicsProblem(horizon=100, complete_model=False), 1, 100, 20, legs=legs, perturb=False)
for legs in (1, 4):
    run("full dynamics N=100 batch 64 (nq=29)", FullDynamicsProblem(horizon=100, complete_model=False), 64, 40, 5, legs=legs)
for legs in (1, 4):
    kp = KinodynamicProblem(horizon=150, complete_model=True)
    run("config 4: kinodynamic N=150 batch 64 (nq=39)", kp, 64, 20, 3, legs=legs, seed=7, perturb_dofs=range(18, kp.nv))
    run("config 4: kinodynamic STAIRS N=150 batch 64 (nq=39)", KinodynamicProblem(horizon=150, complete_model=True), 64, 40, 165, legs=legs, seed=7,
        perturb_dofs=range(18, kp.nv), walk={"z_height": 0.10})  # timed ticks 165 - 205: the first swing at knot 0
for legs in (1, 4):
    kr = KinodynamicProblem(horizon=150, complete_model=False)
    run("config 4: kinodynamic N=150 batch 64 (nq=29)", kr, 64, 20, 3, legs=legs, seed=7, perturb_dofs=range(18, kr.nv))
