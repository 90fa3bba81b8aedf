"""Generates tests/golden/oracle_regression.npz — a REGRESSION PIN OF THE IN-REPO ORACLE, not a reference vector.

The reference ships no golden vectors and its hot path (Aligator / Pinocchio) cannot be run here, so the
oracle stays "parity unpinned" (DESIGN.md §6).  What this fixture does: freeze the oracle's own outputs on three small
problems so that an accidental change of the oracle (which every GPU parity test leans on) is caught on CPU.
Regenerate deliberately with:  python tests/golden/make_oracle_regression.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def cases():
    from tests import _oracle
    from mpc_benchmark_amd import aligator
    from mpc_benchmark_amd.problems.centroidal import CentroidalProblem
    from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
    from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem
    lib = _oracle.load()
    out = {}
    for name, pd, iters in (("centroidal_n10", CentroidalProblem(horizon=10), 100),
                            ("fulldynamic_n4", FullDynamicsProblem(horizon=4), 3),
                            ("kinodynamic_n3", KinodynamicProblem(horizon=3), 2)):
        prob = pd.build()
        solver = pd.make_solver(_native_library=lib)
        solver.max_iters = iters
        solver.linear_solver_choice = aligator.LQ_SOLVER_SERIAL  # the pin was taken on the serial sweep; legs are tested against it
        solver.setup(prob)
        xs, us = pd.initial_guess()
        solver.run(prob, xs, us)
        r = solver.results
        out[name + "_xs"] = np.array(r.xs)
        out[name + "_us"] = np.array(r.us)
        out[name + "_K0"] = np.array(r.controlFeedbacks()[0])
        out[name + "_scalars"] = np.array([r.traj_cost, r.prim_infeas, r.dual_infeas, float(r.num_iters)])
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.npz")
    np.savez_compressed(path, **cases())
    print("wrote", path, os.path.getsize(path), "bytes")
