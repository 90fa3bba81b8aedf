"""Loop body of fulldynamic_talos.py:438-550 (reference generators + stage cycling + terminal-constraint rebuild) on the
oracle backend: the swing foot leaves the ground when its countdown expires and tracks the Bezier reference."""
import numpy as np

from tests import _oracle
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.walking_loop import WalkingMPCLoop


def test_walking_loop_swings_the_right_foot():
    fp = FullDynamicsProblem(horizon=10)
    solver = fp.make_solver(_native_library=_oracle.load())
    loop = WalkingMPCLoop(fp, solver, start_tick=26)  # the right foot takes off 30 + 10 - 26 = 14 ticks from now
    rf0 = fp.robot.foot_placements[1].translation.copy()
    recs = [loop.tick() for _ in range(34)]
    # countdown semantics
    assert recs[0]["takeoff_RF"] == 13 and recs[13]["takeoff_RF"] == 0 and recs[13]["land_RF"] == 80
    # on the ground until take-off, then the reference and the foot rise together
    assert all(abs(r["RF_ref"][2] - rf0[2]) < 1e-4 for r in recs[:14])   # pinned at the measured pose
    assert recs[-1]["RF_ref"][2] > rf0[2] + 0.005
    assert recs[-1]["RF"][2] > rf0[2] + 0.002                    # the MPC follows (one-iteration real-time scheme)
    assert abs(recs[-1]["LF"][2] - fp.robot.foot_placements[0].translation[2]) < 2e-3  # stance foot stays put
    # the horizon now starts in left-only support: one contact model in stage 0
    assert len(loop.problem.stages[0].dynamics.differential_dynamics.constraint_models) == 1
    assert np.isfinite(solver.results.traj_cost) and all(np.all(np.isfinite(x)) for x in loop.xs)
