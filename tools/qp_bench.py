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
