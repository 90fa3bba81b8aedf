"""Developer tool (GPU box): the kinodynamic control pipeline (mpc_benchmark_amd/pipeline.py) over the script's WHOLE schedule for an ensemble of
robots — every MPC period is one kinodynamic solve + 10 low-level periods (feedback terms, ID QP, clamp, simulator step: mpc_qp_low_level_steps) with the
contact set of the simulator following the schedule.  Reports how far the robots walked, who fell, time per period.
usage: python tools/pipeline_walk.py [B] [N] [ticks]   (HOST_GLUE=1: the low-level periods with the glue on the host)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpc_benchmark_amd.pipeline import KinodynamicPipeline
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
complete = bool(int(os.environ.get("COMPLETE", "0")))
kp = KinodynamicProblem(horizon=N, complete_model=complete)
T = int(sys.argv[3]) if len(sys.argv) > 3 else kp.t_mpc - 1
host = bool(int(os.environ.get("HOST_GLUE", "0")))
p = KinodynamicPipeline(kp, batch=B, walk={"per_instance": bool(int(os.environ.get("PERINST", "0")))}, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=True,
                        **({"sigma_q": float(os.environ["SIGMA_Q"])} if os.environ.get("SIGMA_Q") else {}))
p.mpc.options.riccati_legs = max(1, min(32, 256 // B)) if B < 64 else 4
p.mpc.options.corrector_prim_tol = float(os.environ.get("CORRECTOR", "0"))
p.mpc.native.set_options(p.mpc.options)
p.mpc.prepare_schedule(T + 8)
p.cold_solve()
x_start = p.x.copy()
lat, fallen_at = [], {}
for t in range(T):
    t0 = time.perf_counter()
    st = p.tick(host_glue=host)
    lat.append((time.perf_counter() - t0) * 1e3)
    z = p.x[:, 2]
    for b in range(B):
        if b not in fallen_at and (not np.isfinite(z[b]) or abs(z[b] - x_start[b, 2]) > 0.25 or st[b].converged < 0):
            fallen_at[b] = t
    if len(fallen_at) == B:
        break
    if t % 100 == 0:
        print("  tick %4d: base x %.3f .. %.3f  z %.3f .. %.3f  contact state %s  max |tau| / limit %.2f" % (
            t, p.x[:, 0].min(), p.x[:, 0].max(), z.min(), z.max(), list(p.contact_state()), float(np.max(np.abs(p.torques) / p.umax))), flush=True)
lat = np.array(lat)
alive = [b for b in range(B) if b not in fallen_at]
dx = (p.x[alive, 0] - x_start[alive, 0]) if alive else np.zeros(1)
zz = p.x[alive, 2] if alive else np.zeros(1)
print("kinodynamic pipeline over the schedule, %s model, N = %d, %d robots, %d MPC periods (%s glue): period p50 %.2f ms p90 %.2f ms ; fallen %d %s ; "
      "walked (base x of the standing ones) %.3f .. %.3f m ; base height %.3f .. %.3f" % (
          "complete" if complete else "reduced", N, B, len(lat), "host" if host else "device", np.percentile(lat, 50), np.percentile(lat, 90), len(fallen_at),
          sorted(fallen_at.items())[:8], dx.min(), dx.max(), zz.min(), zz.max()))
