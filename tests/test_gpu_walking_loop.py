"""GPU parity of the complete tick loop (reference generators, setReference fast path + batched parameter upload, stage
cycling across a contact switch, terminal-constraint rebuild): HIP library vs oracle, same loop, 1e-6 on trajectories."""
import numpy as np
import pytest

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
            hist.append(np.concatenate([np.ravel(loop.xs), np.ravel(loop.us)]))
        traj[name] = np.array(hist)
    assert _rel(traj["hip"], traj["ref"]) < 1e-6


def test_closed_loop_simulation_matches_oracle():
    """N2 on the GPU: simulated measured state (10 x 1 ms under the feedback law) and three closed-loop ticks, HIP vs oracle."""
    from mpc_benchmark_amd.ensemble import EnsembleMPC
    out, iters = {}, {}
    for name, lib in (("hip", _capi.load_hip_library()), ("ref", _oracle.load())):
        pd = FullDynamicsProblem(horizon=10)
        ens = EnsembleMPC(pd, batch=3, library=lib, seed=5, sigma_q=0.005, sigma_v=0.01)
        ens.prepare_schedule(6)
        iters[name] = [int(x.num_iters) for x in ens.cold_solve(max_iters=40)]
        hist = []
        for _ in range(3):
            ens.native.simulate(10, pd.dt / 10)
            hist.append(ens.native.get_x0().copy())
            ens.step()
            hist.append(ens.results(gains=False)["xs"][:, :3].reshape(3, -1).copy())
        out[name] = hist
    # an instance whose cold solve stops after a different number of iterations in the two libraries (the inner criterion met within
    # round-off of its tolerance) starts the closed loop from a different point: only the others are comparable at 1e-6
    same = [i for i in range(3) if iters["hip"][i] == iters["ref"][i]]
    assert len(same) >= 2, iters
    for a, b in zip(out["hip"], out["ref"]):
        assert _rel(a[same], b[same]) < 1e-6
