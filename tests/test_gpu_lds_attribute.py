"""Two handles of one process whose LDS carve-outs differ (another number of constraint rows for the same state / control dimensions): the
dynamic-LDS limit of a kernel is a setting of the (device, kernel) pair — the handle created FIRST, with the larger carve-out, must still
launch after the second one was created."""
import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem

pytestmark = pytest.mark.gpu


def test_handles_with_different_carve_outs_coexist(hip_lib):
    big = EnsembleMPC(FullDynamicsProblem(horizon=12, complete_model=True), batch=2, library=hip_lib, seed=1)
    big.options.riccati_legs = 4
    big.native.set_options(big.options)
    big.prepare_schedule(8)
    big.cold_solve(max_iters=20)
    ref = big.results(gains=False)
    # a second handle on the same kernels with a smaller carve-out: the generic kernels serve it (reduced model: other dimensions) and a
    # complete-model handle without the terminal constraint rows differs in its row count only
    small = EnsembleMPC(FullDynamicsProblem(horizon=6, complete_model=False), batch=1, library=hip_lib, seed=2)
    small.prepare_schedule(4)
    small.cold_solve(max_iters=5)
    for _ in range(3):
        big.step()   # launches of the first handle after the second one set its attributes
    r = big.results(gains=False)
    assert np.all(np.isfinite(r["xs"])) and np.all(np.isfinite(r["us"]))
    assert r["xs"].shape == ref["xs"].shape
