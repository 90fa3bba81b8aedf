"""N3 on the GPU: the batched dense QP kernel through the C-ABI against the oracle on the same problems (inverse-dynamics
structure of QP_utils.py:437-575, with and without the torque box) and against the KKT conditions directly."""
import numpy as np
import pytest

from tests import _oracle, _qp_cases as cases
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd._qp_capi import BatchedQP

pytestmark = pytest.mark.gpu


def _solve(lib, qs, box, eps, max_iter=60, max_iter_in=40):
    n, neq, nin = qs[0]["H"].shape[0], qs[0]["A"].shape[0], qs[0]["C"].shape[0]
    qp = BatchedQP(len(qs), n, neq, nin, box=box, library=lib)
    qp.settings.eps_abs, qp.settings.max_iter, qp.settings.max_iter_in = eps, max_iter, max_iter_in
    st = lambda k: np.stack([q[k] for q in qs])
    args = [st(k) for k in ("H", "g", "A", "b", "C", "l", "u")] + ([st("l_box"), st("u_box")] if box else [])
    return qp.solve(*args)


@pytest.mark.parametrize("box", [False, True])
def test_qp_matches_oracle_and_kkt(box):
    rng = np.random.default_rng(21)
    contacts = [(True, True), (True, False), (False, True)]
    qs = [cases.id_qp(rng, contact=contacts[i % 3], torque_limit=(45.0 if box else None)) for i in range(12)]
    xh, yh, zh, zbh, ih = _solve(_capi.load_hip_library(), qs, box, 1e-6)
    xr, yr, zr, zbr, ir = _solve(_oracle.load(), qs, box, 1e-6)
    for i, q in enumerate(qs):
        assert ih[i].status == 0 and ir[i].status == 0
        stat, prim, comp = cases.kkt_residuals(q, xh[i], yh[i], zh[i], zbh[i] if box else None)
        assert stat < 2e-6 and prim < 2e-6
        scale = max(1.0, float(np.max(np.abs(xr[i]))))
        assert np.max(np.abs(xh[i] - xr[i])) / scale < 1e-6  # same algorithm, same path: agreement far below eps_abs
        assert ih[i].n_active == ir[i].n_active  # (iteration counts may differ by round-off when a pass ends at the tolerance)


def test_complete_model_size_and_reference_settings():
    """nv = 38 (n = 82, neq = 50) with eps_abs = 1e-3, max_iter = 10, max_iter_in = 10 (QP_utils.py:502-507)."""
    rng = np.random.default_rng(4)
    qs = [cases.id_qp(rng, nv=38) for _ in range(8)]
    xh, yh, zh, _, ih = _solve(_capi.load_hip_library(), qs, False, 1e-3, 10, 10)
    for i, q in enumerate(qs):
        assert ih[i].status == 0 and ih[i].iters <= 10
        stat, prim, comp = cases.kkt_residuals(q, xh[i], yh[i], zh[i])
        assert stat < 2e-3 and prim < 2e-3


def test_id_solver_mirror_hip_equals_oracle():
    """The IDSolver_ulim mirror (QP_utils.py:437-575) on the synthetic Talos: HIP and oracle give the same accelerations,
    forces and torques; a batch of robots in one launch."""
    from mpc_benchmark_amd import qp_utils
    from mpc_benchmark_amd.robot import dynamics as dyn, minipin as pin
    from mpc_benchmark_amd.robot.talos_synth import load_talos
    _, model, _, q0 = load_talos()
    rng = np.random.default_rng(6)
    ids = [model.getFrameId("left_sole_link"), model.getFrameId("right_sole_link")]
    w = 9.81 * pin.computeTotalMass(model)
    items = []
    for i in range(4):
        v = rng.normal(size=model.nv) * 0.05
        q = pin.integrate(model, q0, np.concatenate((np.zeros(6), rng.normal(size=model.nv - 6) * 0.02)))
        data = dyn.compute_all_terms(model, model.createData(), q, v)
        a = rng.normal(size=model.nv) * 0.2
        forces = np.array([5, -3, 0.55 * w, 1, -2, 0, -4, 2, 0.45 * w, 0, 1, 0], dtype=float)
        items.append((data, [True, True], v, a, forces, data.M))
    out = {}
    for name, lib in (("hip", _capi.load_hip_library()), ("ref", _oracle.load())):
        solver = qp_utils.IDSolver_ulim(model, [1.0, 1e-3], 2, 0.8, 0.1, 0.075, ids, 6, False, library=lib, batch=4)
        solver.qp.settings.eps_abs, solver.qp.settings.max_iter, solver.qp.settings.max_iter_in = 1e-6, 60, 40
        out[name] = solver.solve_batch(items)
        assert all(i.status == 0 for i in solver.last_info)
    for h, r in zip(out["hip"], out["ref"]):
        for xh, xr in zip(h, r):
            assert np.max(np.abs(xh - xr)) / max(1.0, np.max(np.abs(xr))) < 1e-6


def test_ikid_solver_mirror_hip_equals_oracle():
    """The IKIDSolver_f6 mirror (QP_utils.py:584-762: inverse kinematics + inverse dynamics in one QP with the torque box, used at
    centroidal_talos.py:326, 435) on the synthetic Talos in double and single support: HIP and oracle give the same accelerations,
    forces and torques, and the solution satisfies dynamics, contact, cone and box conditions."""
    from mpc_benchmark_amd import qp_utils
    from mpc_benchmark_amd.robot import dynamics as dyn, minipin as pin
    from mpc_benchmark_amd.robot.talos_synth import load_talos
    _, model, _, q0 = load_talos()
    rng = np.random.default_rng(5)
    nv = model.nv
    ids = [model.getFrameId("left_sole_link"), model.getFrameId("right_sole_link")]
    Kp, Kd = 100.0, 20.0
    gains = [(np.eye(nv) * Kp, np.eye(nv) * Kd), (np.eye(6) * Kp, np.eye(6) * Kd), None, (np.eye(3) * Kp, np.eye(3) * Kd)]
    w = 9.81 * pin.computeTotalMass(model)
    z3, z6, zn = np.zeros(3), np.zeros(6), np.zeros(nv)
    for cs, forces in (([True, True], np.array([0, 0, 0.5 * w, 0, 0, 0, 0, 0, 0.5 * w, 0, 0, 0], dtype=float)),
                       ([True, False], np.array([0, 0, w, 0, 0, 0, 0, 0, 0, 0, 0, 0], dtype=float))):
        v = rng.normal(size=nv) * 0.05
        q = pin.integrate(model, q0, np.concatenate((np.zeros(6), rng.normal(size=nv - 6) * 0.02)))
        data = dyn.compute_all_terms(model, model.createData(), q, v)
        q_diff = np.concatenate((np.zeros(6), rng.normal(size=nv - 6) * 0.01))
        out = {}
        for name, lib in (("hip", _capi.load_hip_library()), ("ref", _oracle.load())):
            solver = qp_utils.IKIDSolver_f6(model, [1.0, 100.0, 1.0, 10.0, 1e-3], gains, 2, 0.8, 0.1, 0.075, ids, model.getFrameId("base_link"),
                                            model.getFrameId("torso_2_link"), 6, False, library=lib)
            solver.qp.settings.eps_abs = 1e-6
            out[name] = solver.solve(data, cs, v, q_diff, zn, z6, z6, z6, z6, z3, z3, z3, z3, forces, np.zeros(6), data.M)
            assert solver.last_info[0].status == 0
        for xh, xr in zip(out["hip"], out["ref"]):
            assert np.max(np.abs(xh - xr)) / max(1.0, np.max(np.abs(xr))) < 1e-6
        a, f, tau = out["hip"]
        Jc = np.vstack([dyn.frame_jacobian_local(model, data, i) for i, on in zip(ids, cs) if on])
        fa = np.concatenate([f[6 * i:6 * i + 6] for i, on in enumerate(cs) if on])
        S = np.zeros((nv, nv - 6)); S[6:] = np.eye(nv - 6)
        assert np.max(np.abs(data.M @ a + data.nle - S @ tau - Jc.T @ fa)) < 1e-5
        assert np.all(np.abs(tau) <= np.asarray(model.effortLimit)[6:] + 1e-5)


@pytest.mark.parametrize("complete", [False, True])
def test_id_qp_assembled_on_the_device_equals_the_host_mirror(complete):
    """mpc_qp_set_model / mpc_qp_solve_id: the model is uploaded once and one kernel per batch builds A, b, C, l of the
    inverse-dynamics QP (QP_utils.py:120-158) in HBM from (x, a, forces, contact states).  The matrices equal the numpy
    mirror's (double and single support), the solution equals the host-assembled one and the checker's."""
    from tests.test_qp_utils import _id_cases
    from mpc_benchmark_amd import qp_utils
    from mpc_benchmark_amd.robot.talos_synth import load_talos
    cm, rm, qc, qr = load_talos()
    model, q0 = (cm, qc) if complete else (rm, qr)  # complete model: n = 82, the tile-packed matrix-core path of the QP kernel
    rng = np.random.default_rng(12)
    ids = [model.getFrameId("left_sole_link"), model.getFrameId("right_sole_link")]
    B = 6
    x, a, f, cs, items = _id_cases(model, q0, rng, B)
    out = {}
    for name, lib in (("hip", _capi.load_hip_library()), ("ref", _oracle.load())):
        solver = qp_utils.IDSolver_ulim(model, [1.0, 1e-3], 2, 0.8, 0.1, 0.075, ids, 6, False, library=lib, batch=B)
        solver.qp.settings.eps_abs, solver.qp.settings.max_iter, solver.qp.settings.max_iter_in = (1e-6, 200, 100) if complete else (1e-7, 60, 40)
        out[name] = solver.solve_batch_device(x, a, f, cs, return_matrices=True)
        assert all(i.status == 0 for i in solver.last_info), (name, [(i.status, i.prim_res, i.dual_res, i.iters) for i in solver.last_info])
        if name == "hip":
            host = solver.solve_batch(items)
            A, b, C, l = out[name][3]
            for i in range(B):
                Ah, bh, Ch, lh = solver.computeMatrice(*items[i])
                assert np.max(np.abs(A[i] - Ah)) < 1e-10 * max(1.0, np.max(np.abs(Ah)))
                assert np.max(np.abs(b[i] - bh)) < 1e-10 * max(1.0, np.max(np.abs(bh)))
                assert np.array_equal(C[i], Ch) and np.max(np.abs(l[i] - lh)) < 1e-12 * max(1.0, np.max(np.abs(lh)))
                for k in range(3):
                    assert np.max(np.abs(out[name][k][i] - host[i][k])) < 1e-6 * max(1.0, np.max(np.abs(host[i][k])))
    for k in range(3):
        assert np.max(np.abs(out["hip"][k] - out["ref"][k])) < 1e-6 * max(1.0, np.max(np.abs(out["ref"][k])))
    for mh, mr in zip(out["hip"][3], out["ref"][3]):
        assert np.max(np.abs(mh - mr)) < 1e-10 * max(1.0, np.max(np.abs(mr)))


def test_id_entry_point_rejects_wrong_shapes_on_the_device():
    from mpc_benchmark_amd import qp_utils
    from mpc_benchmark_amd.robot.talos_synth import load_talos
    _, model, _, q0 = load_talos()
    ids = [model.getFrameId("left_sole_link"), model.getFrameId("right_sole_link")]
    s = qp_utils.IDSolver_ulim(model, [1.0, 1e-3], 2, 0.8, 0.1, 0.075, ids, 6, False, library=_capi.load_hip_library(), batch=1)
    x = np.concatenate((q0, np.zeros(model.nv)))
    with pytest.raises(RuntimeError, match="mpc_qp_set_model first"):
        s.qp._nqv, s.qp._nv = model.nq + model.nv, model.nv
        s.qp.solve_id(np.array([0, 1], dtype=np.int32), [1.0, 1e-3], s.Cmin, 1.0, x, np.zeros(model.nv), np.zeros(12), [1, 1])
    s.enable_device_assembly()
    with pytest.raises(RuntimeError, match="dimensions"):
        s.qp.solve_id(s._frame_idx[:1], s._weights, s.Cmin, 1.0, x, np.zeros(model.nv), np.zeros(6), [1])
    with pytest.raises(RuntimeError, match="frame index"):
        s.qp.solve_id(np.array([0, 99], dtype=np.int32), s._weights, s.Cmin, 1.0, x, np.zeros(model.nv), np.zeros(12), [1, 1])
    a_new, f_new, tau = s.solve_batch_device(x, np.zeros(model.nv), np.zeros(12), [1, 1])
    assert s.last_info[0].status in (0, 1) and np.all(np.isfinite(tau))


@pytest.mark.parametrize("complete", [False, True])
def test_ikid_qp_assembled_on_the_device_equals_the_host_mirror(complete):
    """mpc_qp_solve_ikid: H, g, A, b, C, l of the IK + ID QP (QP_utils.py:584-762) built in one kernel per batch from the robot state and
    the task errors — the matrices equal the numpy mirror's (double and single support, non-uniform gains), the solution the checker's."""
    from tests.test_qp_utils import _ikid_case, _ikid_solver, _ikid_compare
    from mpc_benchmark_amd.robot.talos_synth import load_talos
    cm, rm, qc, qr = load_talos()
    model, q0 = (cm, qc) if complete else (rm, qr)
    rows = _ikid_case(model, q0, np.random.default_rng(22), 6)
    dev_h = _ikid_compare(_ikid_solver(model, _capi.load_hip_library(), 6), model, rows, tol_m=1e-10)
    dev_o = _ikid_compare(_ikid_solver(model, _oracle.load(), 6), model, rows)
    for k in range(3):
        assert np.max(np.abs(dev_h[k] - dev_o[k])) < 1e-4 * max(1.0, np.max(np.abs(dev_o[k]))), k
    _ikid_compare(_ikid_solver(model, _capi.load_hip_library(), 1), model, rows[:1], tol_m=1e-10)
