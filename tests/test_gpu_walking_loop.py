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
