"""The CPU port that bench.py times (oracle/cpu_port: closed-form derivatives, -O3) against the checker (oracle: forward-mode AD) on
the CPU: per-phase dumps of one ProxDDP iteration at 1e-9 (block-wise norms), eight cold iterations + MPC ticks at 1e-7.  The port reuses
the oracle's solver (Riccati, linesearch, BCL), so the gains are compared as well — they only differ through the stage evaluation."""
import numpy as np
import pytest

from tests import _cpu_port, test_gpu_fulldynamic as T
from tests._phase_parity import compare


@pytest.fixture(scope="module")
def port_lib():
    return _cpu_port.load()


def test_backend_name(port_lib):
    assert port_lib.mpc_backend_name() == b"cpu-port"


def test_one_iteration_phase_parity(port_lib, oracle_lib):
    fp, sp = T._run_one_iteration(port_lib, complete_model=False)
    _, sr = T._run_one_iteration(oracle_lib, complete_model=False)
    N = len(T.PATTERN)
    worst = compare(sp._native, sr._native, T.PHASES + T.GAINS + T.STEPS, range(N + 1), fp.space.ndx, fp.nu, N,
                    skip_terminal=("AB", "f", "E6", "xdot", "wrench", "xnext", "K", "kff", "Mx", "mx", "du"))
    tol = {q: 1e-9 for q in T.PHASES}
    tol.update({q: 1e-8 for q in T.GAINS + T.STEPS})
    tol.update({q + "/dependent": 1.0 for q in ("Knu", "knu", "dvs")})
    tol.update({q + "/dependent_combined": 1e-6 for q in ("Knu", "knu", "dvs")})  # D_dep^T nu_dep: what the regularisation does pin
    bad = {q: e for q, e in worst.items() if not e <= tol[q]}
    assert not bad, "cpu port deviates from the oracle: %s (all: %s)" % (bad, worst)


def test_cold_solve_and_ticks(port_lib, oracle_lib):
    from mpc_benchmark_amd.ensemble import EnsembleMPC
    from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
    out = {}
    for name, lib in (("port", port_lib), ("ref", oracle_lib)):
        e = EnsembleMPC(FullDynamicsProblem(horizon=8), batch=2, library=lib, seed=3, sigma_q=0.005, sigma_v=0.01)
        e.options.num_threads = 4
        e.options.tol = 0.0  # a fixed number of iterations in both libraries
        e.prepare_schedule(8)
        st = e.cold_solve(max_iters=8)
        for _ in range(3):
            e.step()
        r = e.results(gains=True)
        out[name] = (r, [int(s.num_iters) for s in st])
    assert out["port"][1] == out["ref"][1]
    for key in ("xs", "us", "K"):
        a, b = out["port"][0][key], out["ref"][0][key]
        assert float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) < 1e-7, key
