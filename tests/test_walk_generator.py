"""Reference generation in the library (mpc_walk_init / mpc_walk_update, include/mpc_abi.h; csrc/walk_generator.h) against the host generator
(``references.FootTrajectoryBatch`` + ``minipin.frame_placements_batch``, which the drop-in fixtures hold to the reference's own talos_utils.py):
an ensemble with per-instance references walked with ``generator="host"`` and with ``generator="device"`` must carry the same parameter tables —
every knot's two placement references and the terminal targets of every instance — after every tick of a schedule that contains planning windows,
take-offs and landings, and the generator's plan (start / final poses) must equal the host generator's.  CPU: the oracle's implementation."""
import numpy as np
import pytest

from mpc_benchmark_amd.ensemble import EnsembleMPC
from mpc_benchmark_amd.problems.fulldynamic import FullDynamicsProblem
from mpc_benchmark_amd.problems.kinodynamic import KinodynamicProblem


def _ens(lib, problem, generator, **walk):
    pd = problem(horizon=8)
    e = EnsembleMPC(pd, batch=3, library=lib, seed=5, sigma_q=0.01, sigma_v=0.02)
    e.options.riccati_legs = 1
    e.options.num_threads = 8
    e.native.set_options(e.options)
    e.prepare_schedule(60)
    e.cold_solve(max_iters=20)
    e.enable_walk(per_instance=True, generator=generator, **walk)
    return e


def compare_generators(lib, problem, ticks, tol, lockstep=True, **walk):
    eh, ed = _ens(lib, problem, "host", **walk), _ens(lib, problem, "device", **walk)
    N, B = eh.dims.horizon, eh.batch
    worst = 0.0
    for t in range(ticks):
        if lockstep:
            # both from the same solver state: what is compared is the generator (the loops themselves amplify a 1e-16 difference of a reference).
            # (The oracle's mpc_set_state leaves the per-instance tables alone; the HIP library's resets them to the shared ones, so its test lets the
            # two ensembles run on their own instead.)
            ed.native.set_state(eh.native.get_state())
        eh.step(); ed.step()
        for b in range(B):
            for k in range(N + 1):
                ph, pdv = eh.native.debug_get("inst_params", k, b), ed.native.debug_get("inst_params", k, b)
                assert ph.shape == pdv.shape
                d = float(np.max(np.abs(ph - pdv))) if ph.size else 0.0
                assert d < tol, "tick %d instance %d knot %d: tables differ by %.3e at %s" % (t, b, k, d, np.flatnonzero(np.abs(ph - pdv) > tol)[:6])
                worst = max(worst, d)
        g = eh._walk["batch"]
        plan = ed.native.walk_get_state()
        for i, (R, p) in enumerate((g.sL, g.fL, g.sR, g.fR)):
            assert np.max(np.abs(plan[:, i, :9] - R.reshape(B, 9))) < tol and np.max(np.abs(plan[:, i, 9:] - p)) < tol, (t, i)
    return worst


@pytest.mark.parametrize("problem,walk", [(FullDynamicsProblem, {}), (KinodynamicProblem, {}), (KinodynamicProblem, {"z_height": 0.10})])
def test_library_generator_equals_the_host_generator_on_the_oracle(oracle_lib, problem, walk):
    T = 45 if problem is FullDynamicsProblem else 40  # through the first planning window and take-off (T_ds = 30 / 20) into the swing
    worst = compare_generators(oracle_lib, problem, T, 1e-12, **walk)
    print("%s %s: parameter tables of host and library generator within %.1e over %d ticks" % (problem.__name__, walk, worst, T))


def test_floor_is_the_same_rule_in_both_generators(oracle_lib):
    """``enable_walk(floor=...)`` / ``mpc_walk_config.floor_z``: no foothold is planned below the floor.  The full-dynamics walk aims its left foot 1 cm
    below the right one's height (fulldynamic_talos.py:449); with the floor at the initial footholds' height — and, to make every target hit it, 3 mm above —
    the final poses of both generators stop there, their tables stay equal at 1e-12, and the start poses remain what was measured."""
    worst = compare_generators(oracle_lib, FullDynamicsProblem, 45, 1e-12, floor=0.003)
    with_floor, without = _ens(oracle_lib, FullDynamicsProblem, "device", floor=0.003), _ens(oracle_lib, FullDynamicsProblem, "device")
    for _ in range(10):   # (the plan is made inside the double-support window before a take-off: from tick 1 on)
        with_floor.step(); without.step()
    pf, p0 = with_floor.native.walk_get_state(), without.native.walk_get_state()
    assert np.all(pf[:, [1, 3], 11] == 0.003) and np.all(p0[:, [1, 3], 11] < 0.003)   # final poses of both feet: z at the floor / where the rules put them
    assert np.all(pf[:, [0, 2], 11] < 0.003)                                          # start poses: as measured
    with pytest.raises(ValueError):
        _ens(oracle_lib, KinodynamicProblem, "device", z_height=0.10, floor=True)   # stairs have no flat floor
    print("floor: tables of host and library generator within %.1e over 45 ticks" % worst)


@pytest.mark.gpu
def test_floor_rule_on_the_device(hip_lib):
    worst = compare_generators(hip_lib, FullDynamicsProblem, 45, 1e-9, lockstep=False, floor=0.003)
    print("floor (HIP): parameter tables of host and device generator within %.1e" % worst)


@pytest.mark.gpu
@pytest.mark.parametrize("problem,walk", [(FullDynamicsProblem, {}), (KinodynamicProblem, {"z_height": 0.10})])
def test_library_generator_equals_the_host_generator_on_the_device(hip_lib, problem, walk):
    # two ensembles of the same library, one with the numpy generator and one with k_walk_refs, each on its own from the same cold solve: the
    # references differ by round-off (1e-16), which the loops carry along — the tables stay within 1e-9 of each other over the 45 ticks
    worst = compare_generators(hip_lib, problem, 45, 1e-9, lockstep=False, **walk)
    print("%s %s (HIP): parameter tables of host and device generator within %.1e" % (problem.__name__, walk, worst))
