"""The ensemble driver's walk mode (``EnsembleMPC.enable_walk``: the loop bodies of the three scripts as parameter patches on the flat
stage tables) against the same loop bodies written through the ``aligator`` mirror (problems/walking_loop.py: ``setReference``,
``contact_poses[i] = ...``, ``term_constraints.funcs[0].setReference``), both on the oracle: two routes to the same uploads, so the
trajectories must agree to round-off.  The GPU counterpart (HIP vs oracle over a take-off and a landing, stairs included) is
tests/test_gpu_walk_all_problems.py."""
import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems import walking_loop
from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
from tests._metrics import rel_cols


def run_pair(make_pd, lib, ticks, z_height=0.0, x_forward=None):
    pd = make_pd()
    e = EnsembleMPC(pd, batch=1, library=lib, perturb=False)
    e.prepare_schedule(ticks + 2)
    e.cold_solve(max_iters=100)
    e.enable_walk(z_height=z_height, x_forward=x_forward)
    pd2 = make_pd()
    solver = pd2.make_solver(_native_library=lib)
    kw = {} if x_forward is None else {"x_forward": x_forward}
    loop = walking_loop.make_loop(pd2, solver, z_height=z_height, **kw)
    worst = 0.0
    for t in range(ticks):
        e.step()
        loop.tick(x_fk=(pd2.robot.x0 if t == 0 else None))
        r = e.results(gains=False)
        err = max(rel_cols(r["xs"][0], np.array(loop.xs), 1e-3), rel_cols(r["us"][0], np.array(loop.us), 1e-3))
        assert err < 1e-8, "tick %d: ensemble walk and mirror loop differ by %.3e" % (t, err)
        worst = max(worst, err)
    return worst, loop


@pytest.mark.parametrize("name", ["fulldynamic", "kinodynamic", "kinodynamic_stairs", "centroidal"])
def test_ensemble_walk_equals_mirror_loop_on_the_oracle(oracle_lib, name):
    make = {"fulldynamic": lambda: FullDynamicsProblem(horizon=6), "kinodynamic": lambda: KinodynamicProblem(horizon=6),
            "kinodynamic_stairs": lambda: KinodynamicProblem(horizon=6), "centroidal": lambda: CentroidalProblem(horizon=10)}[name]
    # long enough for the first take-off to reach knot 0 (T_ds + N ticks) and the swing to get going
    ticks = {"fulldynamic": 50, "kinodynamic": 40, "kinodynamic_stairs": 40, "centroidal": 60}[name]
    worst, loop = run_pair(make, oracle_lib, ticks, z_height=(0.10 if name.endswith("stairs") else 0.0))
    moved = [h for h in loop.history if np.linalg.norm(h["RF_ref"] - loop.history[0]["RF_ref"]) > 1e-5]
    assert moved, "the right foot's reference never left the ground in %d ticks" % ticks
