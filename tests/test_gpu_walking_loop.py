"""GPU parity of the complete tick loop (reference generators, setReference fast path + batched parameter upload, stage
cycling across a contact switch, terminal-constraint rebuild): HIP library vs oracle, same loop, 1e-6 on trajectories."""
import os

import numpy as np
import pytest

from tests._metrics import rel_cols, traj_err

from tests import _oracle
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.walking_loop import WalkingMPCLoop

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def test_walking_loop_matches_oracle():
    traj = {}
    for name, lib in (("hip", _capi.load_hip_library()), ("ref", _oracle.load())):
        fp = FullDynamicsProblem(horizon=10)
        solver = fp.make_solver(_native_library=lib)
        loop = WalkingMPCLoop(fp, solver, start_tick=26, x_forward=0.05)
        hist = []
        for _ in range(30):
            loop.tick()
            hist.append((np.array(loop.xs), np.array(loop.us)))
        traj[name] = hist
    for t, (a, b) in enumerate(zip(traj["hip"], traj["ref"])):
        e = traj_err(a[0], a[1], b[0], b[1])
        assert e < 1e-6, "tick %d: %.3e" % (t, e)


@pytest.mark.parametrize("mode", ["fixed_iterations", "converged"])
def test_closed_loop_simulation_matches_oracle(mode):
    """N2 on the GPU: simulated measured state (10 x 1 ms under the feedback law) and three closed-loop ticks, HIP vs oracle, EVERY
    instance compared.  "fixed_iterations": the cold solve runs exactly 8 iterations in both libraries (tolerance 0: no convergence
    exit), so the closed loop starts from the same iterate by construction.  "converged": the cold solve runs to convergence; the
    seed is one for which both libraries take the same number of iterations for every instance (asserted: a cold solve that stops
    one iteration apart — the inner criterion met within round-off of its tolerance — starts the loop from a different point)."""
    from mpc_benchmark_amd.ensemble import EnsembleMPC
    out, iters = {}, {}
    for name, lib in (("hip", _capi.load_hip_library()), ("ref", _oracle.load())):
        pd = FullDynamicsProblem(horizon=10)
        # ("converged": small perturbations — cold solves of ~10 iterations, whose stopping iteration is not decided at round-off level
        # the way it is for the 30 - 40 iteration solves of strongly perturbed instances)
        sq, sv = (0.005, 0.01) if mode == "fixed_iterations" else (0.001, 0.002)
        ens = EnsembleMPC(pd, batch=3, library=lib, seed=7, sigma_q=sq, sigma_v=sv)
        ens.options.num_threads = os.cpu_count() or 8
        if mode == "fixed_iterations":
            ens.options.tol = 0.0
        ens.prepare_schedule(6)
        iters[name] = [(int(x.num_iters), bool(x.converged)) for x in ens.cold_solve(max_iters=8 if mode == "fixed_iterations" else 40)]
        hist = []
        for _ in range(3):
            ens.native.simulate(10, pd.dt / 10)
            hist.append(ens.native.get_x0().copy())
            ens.step()
            hist.append(ens.results(gains=False)["xs"][:, :3].reshape(9, -1).copy())  # rows: (instance, knot) ; columns: state components
        out[name] = hist
    assert iters["hip"] == iters["ref"], iters
    if mode == "fixed_iterations":
        assert all(it == (8, False) for it in iters["hip"]), iters
    for a, b in zip(out["hip"], out["ref"]):
        assert rel_cols(a, b, 1e-3) < 1e-6  # every state component against its own range
