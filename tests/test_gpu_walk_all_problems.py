"""GPU parity of the walking loops of the kinodynamic and centroidal scripts (kinodynamic_talos.py:361-497: 0.3 m steps;
centroidal_talos.py:353-468: 0.2 m steps) and of the "stairs" variant of BASELINE.json's kinodynamic configuration (0.10 m gained per
step: the ``z_height`` argument of ``footTrajectory``, talos_utils.py:188-192), through ``EnsembleMPC.enable_walk``: HIP against the oracle
over a stretch that contains a take-off and a landing at knot 0 (reduced horizon), and the full-size schedules through size-independent
properties."""
import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
from tests._metrics import rel_cols

pytestmark = pytest.mark.gpu


def _handle(lib, make_pd, ticks, z_height):
    pd = make_pd()
    e = EnsembleMPC(pd, batch=1, library=lib, perturb=False)
    e.options.num_threads = 8
    e.options.riccati_legs = 1  # the serial sweep in both libraries: a parallel-in-time sweep starts from guesses of the cut Hessians that each
    e.native.set_options(e.options)  # library keeps from its own previous pass (tests/test_gpu_legs.py holds the legs to the serial sweep)
    e.prepare_schedule(ticks + 2)
    return e


@pytest.mark.parametrize("name,z_height", [("kinodynamic", 0.0), ("kinodynamic", 0.10), ("centroidal", 0.0)])
def test_walk_take_off_and_landing_match_the_oracle(hip_lib, oracle_lib, name, z_height):
    """One iteration per tick, perfect-model feedback, references replanned every tick from the predicted foot poses, until the right foot
    has taken off AND landed at knot 0 (T_ds + N + T_ss ticks) — every tick's xs / us / K_0 within 1e-6 per component of the oracle's.
    The two libraries walk in lock-step and every tick starts from the ORACLE's solver state (the portable checkpoint of
    include/mpc_abi.h: iterate, multipliers, penalties, stage tables): each tick is one Newton step of a 1 / mu = 1e8 penalty problem, and
    around the contact switches — where the walk backtracks to alpha = 1/16 — a difference of 1e-9 in the iterate is amplified to 1e-5
    within five ticks in EITHER library; held tick by tick, the comparison is that of the step itself.
    (Horizon 40 for the kinodynamic problem: a 0.3 m step needs more than a few ticks of preview — at N <= 20 the loop itself diverges,
    in both libraries.)"""
    if name == "kinodynamic":
        make, ticks = (lambda: KinodynamicProblem(horizon=40)), 20 + 40 + 80 + 6
    else:
        make, ticks = (lambda: CentroidalProblem(horizon=10)), 20 + 10 + 80 + 6
    er, eh = _handle(oracle_lib, make, ticks, z_height), _handle(hip_lib, make, ticks, z_height)
    assert er.cold_solve(max_iters=100)[0].converged
    eh.cold_solve(max_iters=100)
    eh.native.set_state(er.native.get_state())
    er.enable_walk(z_height=z_height)
    eh.enable_walk(z_height=z_height)
    worst, alphas, log, bad, planned_rf = 0.0, [], [], [], None
    n_land0 = len(er._walk["lists"][2])
    for t in range(ticks):
        sr = er.step()
        sh = eh.step()
        a, b = eh.results(gains=True), er.results(gains=True)
        ex, eu, ek = rel_cols(a["xs"][0], b["xs"][0], 1e-3), rel_cols(a["us"][0], b["us"][0], 1.0), rel_cols(a["K"][0, 0], b["K"][0, 0], 1.0)
        e = max(ex, eu, ek)
        log.append("tick %3d alpha hip %-8g oracle %-8g  xs %.2e us %.2e K0 %.2e  prim %.2e dual %.2e" % (t, sh[0].alpha, sr[0].alpha, ex, eu, ek, sr[0].prim_infeas, sr[0].dual_infeas))
        if sh[0].alpha != sr[0].alpha or not e < 1e-6:
            bad.append(log[-1])
        worst = max(worst, e)
        alphas.append(sr[0].alpha)
        if len(er._walk["lists"][2]) == n_land0:  # the FIRST landing of the right foot is still pending: the foothold planned for it
            planned_rf = np.array(er._walk["traj"].final_pose_right.translation)
        eh.native.set_state(er.native.get_state())
        eh._walk["x_measured"] = er._walk["x_measured"].copy()
        if eh._walk["feet"] is not None:
            eh._walk["feet"] = er._walk["feet"]
    import os
    if os.path.isdir("gpurun_out"):
        with open("gpurun_out/r04_walk_parity_%s_%.2f.txt" % (name, z_height), "w") as f:
            f.write("\n".join(log) + "\n")
    assert not bad, "%s z_height %.2f: %d of %d ticks deviate from the oracle: %s" % (name, z_height, len(bad), ticks, bad[:6])
    # the walk really happened: the right foot's reference left the ground and came down one step further (and higher, on stairs)
    rf0 = np.asarray(er.pd.robot.foot_placements[1].translation)
    assert planned_rf[0] - rf0[0] > 0.15 and abs((planned_rf[2] - rf0[2]) - z_height) < 1e-3, planned_rf
    print("%s z_height %.2f: worst deviation over %d ticks %.3e; ticks that backtracked: %d" % (name, z_height, ticks, worst, sum(1 for x in alphas if x < 1)))


@pytest.mark.parametrize("iters_per_tick,refine,corrector", [(1, 0, 0.0), (2, 0, 0.0), (1, 0, 20.0)])
def test_config4_stairs_whole_schedule(hip_lib, iters_per_tick, refine, corrector):
    """BASELINE.json configuration 4 AS STATED: kinodynamic, N = 150, 64 instances (upper body perturbed), complete model, STAIRS — every
    step 0.3 m forward and 0.10 m up (kinodynamic_talos.py:257 with z_height = 0.10), references replanned every tick, over the
    script's schedule.  Two iterations per tick: the whole 820 ticks, nobody lost, the ensemble ends six footholds further and higher,
    standing.  The script's ONE iteration per tick: the first 640 ticks (five climbing steps) — the replanning of the closing step at
    tick 650 (``updateForward(0, 0, ...)``, kinodynamic_talos.py:368-370: the foothold planned 0.3 m ahead and one stair up comes back
    beside the stance foot within one tick) is more than one Newton step of the penalty problem absorbs on stairs, also for the
    nominal instance (the flat walk passes it) — perturbed instances that are lost on the way are isolated and
    re-seeded from the nominal one, which must not fail.  With the corrector (``corrector_prim_tol = 20``, include/mpc_abi.h: a second
    iteration on the ~2 % of the instance-ticks whose warm start is infeasible by more than that or whose step backtracks) ONE iteration
    per tick walks the whole 820 ticks, closing step included, and nobody is lost (round 5; ``refine_appended_knot = -1``, round 4's
    attempt, does not survive a one-ulp change of the references: profiles/r05_kino_stairs_corrector.txt)."""
    kp = KinodynamicProblem(horizon=150, complete_model=True)
    ens = EnsembleMPC(kp, batch=64, library=hip_lib, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=True)
    ens.options.riccati_legs = 4
    ens.options.refine_appended_knot = refine
    ens.options.corrector_prim_tol = corrector
    ens.native.set_options(ens.options)
    ens.iters_per_tick = iters_per_tick
    ticks = kp.t_mpc - 1 if (iters_per_tick == 2 or corrector > 0) else 640
    ens.prepare_schedule(ticks + 4)
    st = ens.cold_solve(max_iters=100)
    assert all(s.converged for s in st)
    ens.enable_walk(z_height=0.10)
    ens.enable_failure_isolation(auto_revive=True, source=0)
    for _ in range(ticks):
        ens.step()
    r = ens.results(gains=False)
    base = r["xs"][:, 0, :3]
    x0 = kp.robot.x0[:3]
    print("stairs, %d iteration(s) per tick, corrector %g, refine_appended_knot %d: base displacement of the ensemble after %d ticks: x %.3f .. %.3f  z %.3f .. %.3f ; instances lost and revived: %s" % (
        iters_per_tick, corrector, refine, ticks, (base[:, 0] - x0[0]).min(), (base[:, 0] - x0[0]).max(), (base[:, 2] - x0[2]).min(), (base[:, 2] - x0[2]).max(),
        [(t, b, c) for t, b, c, _ in ens.lost]))
    assert np.all(np.isfinite(r["xs"]))
    assert all(b != 0 for _, b, _, _ in ens.lost), "the nominal instance failed"
    assert len(ens.lost) <= (6 if (iters_per_tick == 1 and corrector == 0) else 0), ens.lost
    # steps of 0.3 m / 0.10 m (talos_utils.py:224-246): five of them by tick 640, the closing one after that
    assert np.all(base[:, 0] - x0[0] > 1.1) and np.all(base[:, 2] - x0[2] > 0.35), (base[:, 0].min(), base[:, 2].min())


def test_config2_centroidal_walk_whole_schedule(hip_lib):
    """BASELINE.json configuration 2: centroidal walk (0.2 m steps, centroidal_talos.py:175), N = 100, one instance, the script's
    420-tick schedule, one iteration per tick."""
    cp = CentroidalProblem(horizon=100)
    ens = EnsembleMPC(cp, batch=1, library=hip_lib, perturb=False)
    ticks = cp.t_mpc - 1
    ens.prepare_schedule(ticks + 4)
    assert ens.cold_solve(max_iters=100)[0].converged
    ens.enable_walk()
    for _ in range(ticks):
        ens.step()
    r = ens.results(gains=False)
    com = r["xs"][0, 0, :3]
    print("centroidal walk: CoM displacement after %d ticks: %s" % (ticks, com - cp.x0[:3]))
    assert np.all(np.isfinite(r["xs"])) and com[0] - cp.x0[0] > 0.15
