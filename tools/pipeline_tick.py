"""Developer tool (GPU box): one MPC period of the kinodynamic pipeline (mpc_benchmark_amd/pipeline.py) for an ensemble of robots —
MPC tick + 10 x (ID QP on the device + simulator step): wall time per period and its split.  usage: python tools/pipeline_tick.py [B] [N] [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd.pipeline import KinodynamicPipeline
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = int(sys.argv[3]) if len(sys.argv) > 3 else 40
complete = bool(int(os.environ.get("COMPLETE", "0")))
kp = KinodynamicProblem(horizon=N, complete_model=complete)
p = KinodynamicPipeline(kp, batch=B, walk={}, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=True)
p.mpc.options.riccati_legs = max(1, min(32, 256 // B)) if B < 64 else 4
p.mpc.native.set_options(p.mpc.options)
p.mpc.prepare_schedule(T + 8)
st = p.cold_solve()
for _ in range(3):
    p.tick()
lat, low = [], []
for _ in range(T):
    t0 = time.perf_counter()
    cs = p.contact_state(); p._set_sim_contacts(cs)
    t1 = time.perf_counter()
    p.tick()
    lat.append((time.perf_counter() - t0) * 1e3)
lat = np.array(lat)
# the low-level part alone: inside the library (mpc_qp_low_level_steps: what tick() runs) and with the glue on the host (the round-4 form)
cs = p.contact_state(); p._set_sim_contacts(cs)
p.low_level_loop(cs)
t0 = time.perf_counter()
for _ in range(10):
    p.low_level_loop(cs)
ll = (time.perf_counter() - t0) / (10 * p.substeps) * 1e3
t0 = time.perf_counter()
for _ in range(50):
    p.low_level_step(cs)
lh = (time.perf_counter() - t0) / 50 * 1e3
print("kinodynamic pipeline, %s model, N = %d, %d robots: MPC period p50 %.2f ms p90 %.2f ms ; one low-level step (feedback terms, ID QP assembled + solved, clamp, simulator step: all on the device, one synchronisation per %d steps) %.3f ms ; with the glue on the host %.3f ms ; base heights %.4f .. %.4f" % (
    "complete" if complete else "reduced", N, B, np.percentile(lat, 50), np.percentile(lat, 90), p.substeps, ll, lh, p.x[:, 2].min(), p.x[:, 2].max()))
