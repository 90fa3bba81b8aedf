"""Developer tool: throughput of the batched QP kernel (N3) on inverse-dynamics QPs of the reference's shape, and the oracle
on the host beside it.  usage: python tools/qp_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import _qp_cases as cases, _oracle
from mpc_benchmark_amd import _capi
from mpc_benchmark_amd._qp_capi import BatchedQP

for nv, B in ((28, 64), (28, 256), (28, 1024), (38, 256)):
    rng = np.random.default_rng(1)
    qs = [cases.id_qp(rng, nv=nv) for _ in range(min(B, 64))]
    st = lambda k: np.stack([qs[i % len(qs)][k] for i in range(B)])
    args = [st(k) for k in ("H", "g", "A", "b", "C", "l", "u")]
    n, neq, nin = args[0].shape[1], args[2].shape[1], args[4].shape[1]
    for name, lib, reps in (("hip", _capi.load_hip_library(), 5), ("oracle (1 core)", _oracle.load(), 1)):
        if name != "hip" and B > 64:
            continue
        qp = BatchedQP(B, n, neq, nin, library=lib)
        qp.settings.eps_abs, qp.settings.max_iter, qp.settings.max_iter_in = 1e-3, 10, 10  # QP_utils.py:502-507
        qp.solve(*args)
        t0 = time.perf_counter()
        for _ in range(reps):
            x, y, z, zb, info = qp.solve(*args)
        dt = (time.perf_counter() - t0) / reps
        print("n=%d neq=%d nin=%d batch %4d %-16s %8.3f ms per call  %9.0f QPs/s  (Newton steps per QP: mean %.1f, solved %d/%d)" % (
            n, neq, nin, B, name, dt * 1e3, B / dt, np.mean([i.iters_in for i in info]), sum(i.status == 0 for i in info), B))

# ---- the IDSolver_ulim mirror end to end: matrices assembled on the host in numpy (as the reference does with Pinocchio) against
#      mpc_qp_solve_id, which builds them on the device from (x, a, forces, contact states) ----
from tests.test_qp_utils import _id_cases
from mpc_benchmark_amd import qp_utils
from mpc_benchmark_amd.robot import dynamics as dyn
from mpc_benchmark_amd.robot.talos_synth import load_talos
complete, reduced, qc, qr = load_talos()
for model, q0, B in ((reduced, qr, 1), (reduced, qr, 64), (reduced, qr, 256), (complete, qc, 64), (complete, qc, 256)):
    ids = [model.getFrameId("left_sole_link"), model.getFrameId("right_sole_link")]
    x, a, f, cs, items = _id_cases(model, q0, np.random.default_rng(3), B)
    solver = qp_utils.IDSolver_ulim(model, [1.0, 1e-3], 2, 0.8, 0.1, 0.075, ids, 6, False, batch=B)
    for _ in range(3):
        solver.solve_batch_device(x, a, f, cs)
    t0 = time.perf_counter()
    for _ in range(10):
        dev = solver.solve_batch_device(x, a, f, cs)
    t_dev = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    items = [(dyn.compute_all_terms(model, model.createData(), xi[:model.nq], xi[model.nq:]), list(c), xi[model.nq:], ai, fi, None) for xi, ai, fi, c in zip(x, a, f, cs)]
    items = [(d, c, v, ai, fi, d.M) for d, c, v, ai, fi, _ in items]
    t_terms = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = solver.solve_batch(items)
    t_host = time.perf_counter() - t0
    err = max(np.max(np.abs(dev[k][i] - host[i][k])) for i in range(B) for k in range(3))
    print("IDSolver_ulim nv=%d batch %4d: device assembly + solve %.3f ms per call (%.0f QPs/s) ; host: rigid-body terms in numpy %.1f ms + assembly, upload, solve %.1f ms ; "
          "max difference of (a, f, tau) %.1e" % (model.nv, B, t_dev * 1e3, B / t_dev, t_terms * 1e3, t_host * 1e3, err))

# ---- the IKIDSolver_f6 mirror: the IK + ID QP assembled on the device (mpc_qp_solve_ikid) against host assembly ----
from tests.test_qp_utils import _ikid_case, _ikid_solver
for model, q0, B in ((reduced, qr, 64), (reduced, qr, 256), (complete, qc, 256)):
    rows = _ikid_case(model, q0, np.random.default_rng(4), B)
    solver = _ikid_solver(model, _capi.load_hip_library(), B)
    solver.qp.settings.eps_abs, solver.qp.settings.max_iter, solver.qp.settings.max_iter_in = 1e-3, 100, 100  # QP_utils.py:652-657
    stack = lambda k: np.array([r[k] for r in rows])
    args = (stack("x"), stack("q_diff"), stack("dq_diff"), stack("LF"), stack("dLF"), stack("RF"), stack("dRF"), stack("base"), stack("dbase"), stack("torso"), stack("dtorso"),
            stack("forces"), stack("dH"), np.array([r["cs"] for r in rows], dtype=np.int32))
    for _ in range(3):
        solver.solve_batch_device(*args)
    t0 = time.perf_counter()
    for _ in range(10):
        solver.solve_batch_device(*args)
    t_dev = (time.perf_counter() - t0) / 10
    print("IKIDSolver_f6 nv=%d batch %4d: device assembly + solve %.3f ms per call (%.0f QPs/s), solved %d/%d, Newton steps per QP mean %.1f" % (
        model.nv, B, t_dev * 1e3, B / t_dev, sum(i.status == 0 for i in solver.last_info), B, np.mean([i.iters_in for i in solver.last_info])))
