"""The kinodynamic control pipeline (mpc_benchmark_amd/pipeline.py: MPC tick -> K_0 feedback -> inverse-dynamics QP assembled on the
library -> torque-driven simulator step, kinodynamic_talos.py:361-497) — on the oracle here (CPU), HIP against the oracle in the GPU test."""
import numpy as np
import pytest

from mpc_benchmark_amd.pipeline import KinodynamicPipeline
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
from tests._metrics import rel_cols


def _pipeline(lib, batch=2, horizon=40, walk=None):
    p = KinodynamicPipeline(KinodynamicProblem(horizon=horizon), batch=batch, library=lib, walk=walk, perturb=True, sigma_q=0.005, sigma_v=0.01)
    p.mpc.options.num_threads = 8
    p.mpc.native.set_options(p.mpc.options)
    p.mpc.prepare_schedule(80)
    assert all(s.converged >= 0 for s in p.cold_solve())  # (a perturbed kinodynamic cold solve may stop at the iteration limit: the loop below is what is checked)
    return p


def test_pipeline_keeps_the_robots_standing_on_the_oracle(oracle_lib):
    p = _pipeline(oracle_lib, walk={})
    z0 = p.x[:, 2].copy()
    for _ in range(25):
        st = p.tick()
        assert all(s.converged >= 0 for s in st)
    assert np.all(np.abs(p.x[:, 2] - z0) < 5e-3), p.x[:, 2] - z0          # nobody falls in 250 simulated milliseconds
    assert np.all(np.abs(p.torques) <= p.umax + 1e-12)                    # kinodynamic_talos.py:448-456
    assert np.all(p.forces[:, 2] > 100.0) and np.all(p.forces[:, 8] > 100.0)  # both feet carry weight in double support


def _glue_vs_host(lib, ticks, tol):
    """mpc_qp_low_level_steps (the ten low-level periods inside the library) against the same periods one library call at a time with the feedback terms,
    the clamp and the bookkeeping of x_measured in numpy (KinodynamicPipeline.low_level_step, kinodynamic_talos.py:411-462)."""
    pl, ph = _pipeline(lib, walk={}), _pipeline(lib, walk={})
    worst = 0.0
    for t in range(ticks):
        sl, sh = pl.tick(), ph.tick(host_glue=True)
        ex = rel_cols(pl.x, ph.x, 1e-3)
        ep = rel_cols(pl.x_prev, ph.x_prev, 1e-3)
        et = rel_cols(pl.torques, ph.torques, 1.0)
        ef = rel_cols(pl.forces, ph.forces, 1.0)
        assert max(ex, ep, et, ef) < tol, "tick %d: states %.2e (before the last period %.2e) torques %.2e forces %.2e" % (t, ex, ep, et, ef)
        assert [s.num_iters for s in sl] == [s.num_iters for s in sh]
        worst = max(worst, ex, ep, et, ef)
    return worst


def test_library_low_level_loop_equals_host_glue_on_the_oracle(oracle_lib):
    print("oracle: library loop against host glue over 6 periods: %.3e" % _glue_vs_host(oracle_lib, 6, 1e-9))


def test_low_level_loop_rejects_mismatched_handles(oracle_lib):
    p = _pipeline(oracle_lib, walk={})
    other = KinodynamicPipeline(KinodynamicProblem(horizon=40), batch=3, library=oracle_lib)
    with pytest.raises(RuntimeError, match="same batch size"):
        p.qp.qp.low_level_steps(p.mpc.native, other.sim, p.qp._frame_idx, p.qp._weights, p.qp.Cmin, 50.0, np.ones((2, 2), dtype=np.int32), p.umax, 1, 1e-3, x=p.x)
    with pytest.raises(RuntimeError, match="positive"):
        p.qp.qp.low_level_steps(p.mpc.native, p.sim, p.qp._frame_idx, p.qp._weights, p.qp.Cmin, 50.0, np.ones((2, 2), dtype=np.int32), p.umax, 0, 1e-3, x=p.x)


@pytest.mark.gpu
def test_library_low_level_loop_equals_host_glue_on_the_device(hip_lib):
    """The two glue kernels (csrc/pipeline_glue.h) between plan, QP and simulator against the numpy glue around the same library calls."""
    print("HIP: device glue against host glue over 8 periods: %.3e" % _glue_vs_host(hip_lib, 8, 1e-9))


@pytest.mark.gpu
def test_pipeline_hip_matches_oracle(hip_lib, oracle_lib):
    """Eight MPC periods (80 low-level steps: QP + simulator step each) of two perturbed robots: measured states, QP torques and contact
    forces of the HIP pipeline against the oracle pipeline, tick by tick.  The QP is solved to eps_abs = 1e-3 in <= 10 iterations
    (QP_utils.py:500-507) by the same algorithm in both libraries: the torques agree far below that tolerance."""
    ph, po = _pipeline(hip_lib, walk={}), _pipeline(oracle_lib, walk={})
    ph.mpc.native.set_state(po.mpc.native.get_state())  # same cold-solved start
    ph._fetch()
    worst = 0.0
    for t in range(8):
        ph.tick(), po.tick()
        ex = rel_cols(ph.x, po.x, 1e-3)
        et = rel_cols(ph.torques, po.torques, 1.0)
        ef = rel_cols(ph.forces, po.forces, 1.0)
        assert ex < 1e-6 and et < 1e-5 and ef < 1e-5, "tick %d: states %.2e torques %.2e forces %.2e" % (t, ex, et, ef)
        worst = max(worst, ex, et, ef)
    print("pipeline: worst deviation over 8 ticks %.3e" % worst)


def _short_horizon_pipeline(lib):
    p = KinodynamicPipeline(KinodynamicProblem(horizon=20), batch=2, library=lib, walk={}, perturb=True, sigma_q=0.005, sigma_v=0.01)
    p.mpc.options.num_threads = 8
    p.mpc.native.set_options(p.mpc.options)
    p.mpc.prepare_schedule(80)
    p.cold_solve()
    return p


@pytest.mark.gpu
def test_pipeline_hip_matches_oracle_into_single_support(hip_lib, oracle_lib):
    """Every MPC period of the oracle's walk through the first take-off, repeated by the HIP pipeline FROM THE SAME STATE: a short horizon (N = 20) puts the contact
    switch of the low-level loop — the simulator's contact set, the QP's contact states — at period 38; 43 periods = 430 QP + simulator steps, the last 50 of them
    on one foot.  Measured states, QP torques and contact forces of each period within 1e-6 (measured: 1e-10 while both feet stand, 6e-8 in the period of the
    take-off), the accepted step lengths equal (instance 0 backtracks to 1/64 on the way: the linesearch is inside)."""
    po, ph = _short_horizon_pipeline(oracle_lib), _short_horizon_pipeline(hip_lib)
    worst, single = 0.0, 0
    for t in range(43):
        ph.mpc.native.set_state(po.mpc.native.get_state())
        ph.x, ph.x_prev, ph._plan_stale = po.x.copy(), po.x_prev.copy(), True
        sh, so = ph.tick(), po.tick()
        assert list(ph.contact_state()) == list(po.contact_state())
        assert [a.alpha for a in sh] == [b.alpha for b in so] and [a.num_iters for a in sh] == [b.num_iters for b in so], "period %d" % t
        single += int(not all(po.contact_state()))
        e = max(rel_cols(ph.x, po.x, 1e-3), rel_cols(ph.torques, po.torques, 1.0), rel_cols(ph.forces, po.forces, 1.0))
        assert e < 1e-6, "period %d (contact state %s): %.2e" % (t, list(po.contact_state()), e)
        worst = max(worst, e)
    assert single >= 5
    print("pipeline, period by period along the oracle's walk into single support: worst deviation over 43 periods %.3e (%d periods on one foot)" % (worst, single))


@pytest.mark.gpu
def test_pipeline_free_running_beside_the_oracle(hip_lib, oracle_lib):
    """The same two pipelines FREE-RUNNING (never re-synchronised) from one cold solve.  While the solves are well determined the loops stay together at 1e-9
    (the first 20 periods, asserted at 1e-6).  From period 22 on single solves of the struggling instance 0 (it backtracks to alpha = 1/8) return controls
    that differ by up to 5e-4 between the libraries FROM IDENTICAL STATES in directions the stage KKT systems hardly determine (rounds 5 and 6 alike:
    profiles/r06_pipeline_one_step.txt) — the low-level loop feeds that back, and how far the two walks are apart afterwards depends on the round-off of the
    build (round 5: 1e-7 at period 43 ; round 6: 4e-4 at period 30, 3e-5 at 38).  What is asserted there is that they remain two copies of the same walk: equal
    contact states and step lengths, states within 1e-2, through the take-off at period 38."""
    po, ph = _short_horizon_pipeline(oracle_lib), _short_horizon_pipeline(hip_lib)
    ph.mpc.native.set_state(po.mpc.native.get_state())
    ph._fetch()
    early = late = 0.0
    for t in range(43):
        sh, so = ph.tick(), po.tick()
        assert list(ph.contact_state()) == list(po.contact_state())
        e = max(rel_cols(ph.x, po.x, 1e-3), rel_cols(ph.torques, po.torques, 1.0), rel_cols(ph.forces, po.forces, 1.0))
        if t < 20:
            assert e < 1e-6, "period %d: %.2e" % (t, e)
            early = max(early, e)
        else:
            assert e < 1e-2 and [a.alpha for a in sh] == [b.alpha for b in so], "period %d: %.2e, alpha %s against %s" % (t, e, [a.alpha for a in sh], [b.alpha for b in so])
            late = max(late, e)
    print("pipeline free-running: %.3e over the first 20 periods, %.3e over the 23 after them" % (early, late))


@pytest.mark.gpu
def test_pipeline_walks_the_whole_schedule(hip_lib):
    """kinodynamic_talos.py:361-497 from the first tick to the last for eight perturbed robots, every instance replanning its steps from its own
    feet: 819 MPC periods = 8 190 low-level periods (feedback terms, inverse-dynamics QP, clamp, simulator step — mpc_qp_low_level_steps), the
    simulator's contact set following the schedule through six steps.  Nobody falls, every robot ends 1.5 m further (six steps of 0.3 m, the
    last one closing), the torques stay inside the effort limits.  (profiles/r05_pipeline_walk.txt: the same with 64 robots, 8.9 ms per period.)"""
    kp = KinodynamicProblem(horizon=100)
    p = KinodynamicPipeline(kp, batch=8, library=hip_lib, walk={"per_instance": True}, seed=7, perturb_dofs=range(18, kp.nv), tick_reuse=True)
    p.mpc.options.riccati_legs = 32
    p.mpc.native.set_options(p.mpc.options)
    T = kp.t_mpc - 1
    p.mpc.prepare_schedule(T + 8)
    p.cold_solve()
    x0 = p.x.copy()
    worst_tau, single_support = 0.0, 0
    for t in range(T):
        st = p.tick()
        assert all(s.converged >= 0 for s in st), (t, [s.converged for s in st])
        assert np.all(np.isfinite(p.x)) and np.all(np.abs(p.x[:, 2] - x0[:, 2]) < 0.05), (t, p.x[:, 2] - x0[:, 2])
        worst_tau = max(worst_tau, float(np.max(np.abs(p.torques) / p.umax)))
        single_support += int(not all(p.contact_state()))
    walked = p.x[:, 0] - x0[:, 0]
    print("pipeline over %d MPC periods (%d of them in single support): walked %.3f .. %.3f m, largest |tau| / limit %.2f" % (T, single_support, walked.min(), walked.max(), worst_tau))
    assert single_support > 300
    assert np.all(walked > 1.4) and np.all(walked < 1.6), walked
    assert worst_tau <= 1.0 + 1e-12
