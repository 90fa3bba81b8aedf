"""The fixed-dimension instantiations of the hot kernels (dimensions as compile-time constants — the sweep, the
leg kernels and the stage kernel for the complete Talos model) against the generic ones: same source, same arithmetic, so the
iterates must agree to round-off (bit for bit unless the compiler contracts a multiply-add in one instantiation and not in the other).
``MPC_HIP_GENERIC_DIMS=1`` selects the generic kernels (read when a handle is created and at every launch of the stage kernel), so
the two runs are made one after the other."""
import os

import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem

pytestmark = pytest.mark.gpu


def _run(hip_lib, make, legs, ticks, generic):
    if generic:
        os.environ["MPC_HIP_GENERIC_DIMS"] = "1"
    else:
        os.environ.pop("MPC_HIP_GENERIC_DIMS", None)
    try:
        e = make(hip_lib)
        e.options.riccati_legs = legs
        e.native.set_options(e.options)
        e.prepare_schedule(ticks + 4)
        st = e.cold_solve(max_iters=100)
        for _ in range(ticks):
            e.step()
        r = e.results(gains=True)
        return st, {k: np.array(r[k]) for k in ("xs", "us", "K")}, int(e.native.debug_get("fixed_dims", 0)[0])
    finally:
        os.environ.pop("MPC_HIP_GENERIC_DIMS", None)


@pytest.mark.parametrize("name,complete,legs,model_id", [("fulldynamic", True, 4, 1), ("fulldynamic", True, 32, 1), ("kinodynamic", True, 4, 2),
                                                        ("fulldynamic", False, 4, 3), ("kinodynamic", False, 4, 4)])
def test_fixed_dimension_kernels_equal_the_generic_ones(hip_lib, name, complete, legs, model_id):
    if name == "fulldynamic":
        make = lambda lib: EnsembleMPC(FullDynamicsProblem(horizon=40, complete_model=complete), batch=3, library=lib, seed=5)
    else:
        kp = KinodynamicProblem(horizon=40, complete_model=complete)
        make = lambda lib: EnsembleMPC(kp, batch=3, library=lib, seed=5, perturb_dofs=range(18, kp.nv))
    sf, fixed, idf = _run(hip_lib, make, legs, 6, generic=False)
    sg, gen, idg = _run(hip_lib, make, legs, 6, generic=True)
    assert idf == model_id and idg == 0, "expected the fixed-dimension kernels of model %d, then the generic ones: got %d, %d" % (model_id, idf, idg)
    assert [s.num_iters for s in sf] == [s.num_iters for s in sg]
    same = all(np.array_equal(fixed[k], gen[k]) for k in fixed)
    print("%s, %d legs: fixed-dimension and generic kernels %s" % (name, legs, "agree bit for bit" if same else "agree to round-off only"))
    for k in fixed:
        scale = np.maximum(1.0, np.abs(gen[k]))
        assert np.max(np.abs(fixed[k] - gen[k]) / scale) < 1e-9, k


@pytest.mark.parametrize("name,legs", [("fulldynamic", 32), ("fulldynamic", 4), ("kinodynamic", 8)])
def test_blocked_tree_elimination_equals_the_pivoted_one(hip_lib, name, legs):
    """k_leg_compose eliminates on the matrix cores by panels of four columns with the pivots on the diagonal (bounded multipliers, the pivoted
    Gauss-Jordan as fallback: DESIGN.md section 4); ``MPC_HIP_TREE_PIVOTED=1`` (read at every launch) keeps the pivoted form.  Same linear systems:
    the iterates agree to round-off over a cold solve and MPC ticks."""
    def run(pivoted):
        if pivoted:
            os.environ["MPC_HIP_TREE_PIVOTED"] = "1"
        else:
            os.environ.pop("MPC_HIP_TREE_PIVOTED", None)
        try:
            if name == "fulldynamic":
                e = EnsembleMPC(FullDynamicsProblem(horizon=96, complete_model=True), batch=2, library=hip_lib, seed=11)
            else:
                kp = KinodynamicProblem(horizon=64, complete_model=True)
                e = EnsembleMPC(kp, batch=2, library=hip_lib, seed=11, perturb_dofs=range(18, kp.nv))
            e.options.riccati_legs = legs
            e.native.set_options(e.options)
            e.prepare_schedule(16)
            st = e.cold_solve(max_iters=100)
            for _ in range(10):
                e.step()
            r = e.results(gains=True)
            return [s.num_iters for s in st], {k: np.array(r[k]) for k in ("xs", "us", "K")}
        finally:
            os.environ.pop("MPC_HIP_TREE_PIVOTED", None)

    it_b, blocked = run(False)
    it_p, pivoted = run(True)
    assert it_b == it_p
    for k in blocked:
        scale = np.maximum(1.0, np.abs(pivoted[k]))
        err = float(np.max(np.abs(blocked[k] - pivoted[k]) / scale))
        assert err < 1e-6, (k, err)  # (BASELINE.json's tolerance, entry by entry, after a cold solve and ten ticks of two different eliminations)
