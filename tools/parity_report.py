"""Developer tool (GPU box): block-wise errors of the per-phase dumps, HIP against the oracle, for choosing / checking the tolerances
of the parity tests.  usage: python tools/parity_report.py [complete] [kino]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpc_benchmark_amd import _capi
from tests import _oracle, _metrics
from tests import test_gpu_fulldynamic as T

complete = "complete" in sys.argv
hip, ora = _capi.load_hip_library(), _oracle.load()
fp, sh = T._run_one_iteration(hip, complete)
_, sr = T._run_one_iteration(ora, complete)
N = len(T.PATTERN)
shapes = {"H": lambda a, nz, n: a.reshape(nz, nz), "AB": lambda a, nz, n: a.reshape(n, nz), "CD": lambda a, nz, n: a.reshape(-1, nz),
          "P": lambda a, nz, n: a.reshape(n, n), "K": lambda a, nz, n: a.reshape(-1, n), "Knu": lambda a, nz, n: a.reshape(-1, n)}
print("%-6s %10s %10s %10s %10s   (max|b|, min row max|b|)" % ("q", "global", "rows f=1e-9", "tiles 1e-9", "tiles f=1"))
n = fp.space.ndx
for q in T.PHASES + T.GAINS + T.STEPS:
    g = r = t = t1 = 0.0
    mx, mn = 0.0, 1e300
    for k in range(N + 1):
        if k == N and q in ("AB", "f", "E6", "xdot", "wrench", "xnext", "K", "kff", "Mx", "mx", "du"):
            continue
        a, b = sh._native.debug_get(q, k), sr._native.debug_get(q, k)
        nz = n + (fp.nu if k < N else 0)
        if q in shapes and a.size:
            a, b = shapes[q](a, nz, n), shapes[q](b, nz, n)
        g = max(g, _metrics.rel_global(a, b)); r = max(r, _metrics.rel_rows(a, b, 1e-9)); t = max(t, _metrics.rel_tiles(a, b, 1e-9)); t1 = max(t1, _metrics.rel_tiles(a, b, 1.0))
        if a.size:
            mx = max(mx, float(np.max(np.abs(b)))); rm = np.max(np.abs(np.atleast_2d(b)), axis=1); rm = rm[rm > 0]; mn = min(mn, float(rm.min()) if rm.size else mn)
    print("%-6s %10.2e %10.2e %10.2e %10.2e   (%.2e, %.2e)" % (q, g, r, t, t1, mx, mn))
