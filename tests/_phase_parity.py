"""Comparison of the per-phase dumps of two libraries (HIP against the oracle) with block-wise norms (tests/_metrics.py).

Matrices (H, AB, CD, E6, P, K, ...) are held tile by tile (16 x 16), vectors against their own largest entry (no clamp to 1).
Dual quantities (rows = constraint rows: Knu, knu, dvs, Znu, Knup): the active rows that take part in a linear dependency of the
active set — e.g. more than six active rows of one 17-row wrench cone on a 6-D wrench — have multipliers that the stage KKT system
fixes through its mu = 1e-8 regularisation only; rounding there is amplified by 1 / mu in BOTH libraries (the oracle itself is 4e-4
from a pivoted dense KKT solve on such rows, tests/test_oracle_lq.py).  Those rows are identified from the oracle's [C D]
(``dependent_active_rows``) and reported; every other row is held to the strict tolerance.  What the regularisation DOES determine
of the dependent rows is their combined action on the controls, ``D_dep^T nu_dep`` (the term of the dependent rows in the
stage's stationarity condition in u; the dependency is one of the control columns, so the x columns are not pinned): that is compared as '<q>/dependent_combined' and held to a tight tolerance by the callers, so the
multiplier path of those rows is not left unchecked."""
import numpy as np

from tests._metrics import dependent_active_rows, rel_rows, rel_tiles

MATRIX_COLS = {"H": "nz", "AB": "nz", "CD": "nz", "E6": 6, "P": "n", "K": "n", "Mx": "n", "Phi": "n", "Gam": "n", "Ku": "n", "Lm": "n", "Sg": "n",
               "Zx": "n", "calP": "n"}
DUAL = ("Knu", "knu", "dvs", "Znu", "Knup")
FLOOR = 1e-9


def _shape(q, a, n, nz):
    cols = MATRIX_COLS.get(q)
    if cols is None or a.size == 0:
        return a
    cols = n if cols == "n" else nz if cols == "nz" else cols
    return a.reshape(-1, cols) if a.size % cols == 0 else a


def dual_rows(nr, k, n, nz):
    """(active, dependent) masks over the constraint rows of knot k, from the reference library's dumps."""
    CD = nr.debug_get("CD", k).reshape(-1, nz)
    Kr = nr.debug_get("Knu", k).reshape(CD.shape[0], -1) if CD.shape[0] else np.zeros((0, n))
    kr = nr.debug_get("knu", k).ravel()[:CD.shape[0]]
    act = np.any(Kr != 0, axis=1) | (kr != 0)
    return act, dependent_active_rows(CD, act, n), CD


def compare(nh, nr, quantities, knots, n, nu, N, skip_terminal=()):
    """worst block-wise error per quantity over the knots; dual quantities return two entries: '<q>' (rows outside every
    dependency of the active set) and '<q>/dependent' (the others, informational)."""
    worst = {}
    for k in knots:
        nz = n + (nu if k < N else 0)
        masks = None
        for q in quantities:
            if k == N and q in skip_terminal:
                continue
            a, b = nh.debug_get(q, k), nr.debug_get(q, k)
            assert a.shape == b.shape, (q, k, a.shape, b.shape)
            if q in DUAL:
                if masks is None:
                    masks = dual_rows(nr, k, n, nz)
                act, dep, CD = masks
                c = act.size
                if c == 0:
                    continue
                cols = 1 if q in ("knu", "dvs") else a.size // c  # (dvs is dumped with the handle's maximal row count: the first c rows count)
                a2 = (a.ravel()[:c * cols]).reshape(c, cols)
                b2 = (b.ravel()[:c * cols]).reshape(c, cols)
                good = ~dep
                if good.any():
                    worst[q] = max(worst.get(q, 0.0), rel_rows(a2[good], b2[good], FLOOR))
                if dep.any():
                    worst[q + "/dependent"] = max(worst.get(q + "/dependent", 0.0), rel_rows(a2[dep], b2[dep], FLOOR))
                    ca, cb = CD[dep][:, n:].T @ a2[dep], CD[dep][:, n:].T @ b2[dep]  # (nu x cols): what the dependent multipliers do to u together
                    e = float(np.max(np.abs(ca - cb)) / (np.max(np.abs(cb)) + FLOOR))
                    worst[q + "/dependent_combined"] = max(worst.get(q + "/dependent_combined", 0.0), e)
            else:
                worst[q] = max(worst.get(q, 0.0), rel_tiles(_shape(q, a, n, nz), _shape(q, b, n, nz), FLOOR))
    return worst
