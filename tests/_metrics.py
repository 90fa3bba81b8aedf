"""Error norms of the parity tests.

``rel_global`` is the norm-wise bound the first rounds used: max|a - b| / max(1, max|b|) over the whole array.  For matrices
whose entries span many decades (H, P: 1e-4 .. 1e8) it lets an element-wise wrong small block pass, so the per-phase tests use
the block-wise forms: the array is cut in rows (``rel_rows``) or 16 x 16 tiles (``rel_tiles``) and every block is held to
``max|a - b|_block <= tol * (max|b|_block + floor)`` — the error is normalised by the block's own magnitude; ``floor`` (an
absolute number, stated per quantity in the test) only keeps blocks of exact zeros / pure rounding noise from dividing by nothing."""
import numpy as np


def rel_global(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0


def _as2d(a):
    a = np.asarray(a, float)
    if a.ndim == 0:
        return a.reshape(1, 1)
    if a.ndim == 1:
        return a.reshape(1, -1)
    return a.reshape(-1, a.shape[-1])


def rel_rows(a, b, floor):
    """max over rows of max|a - b|_row / (max|b|_row + floor)."""
    a, b = _as2d(a), _as2d(b)
    if a.size == 0:
        return 0.0
    num = np.max(np.abs(a - b), axis=1)
    den = np.max(np.abs(b), axis=1) + floor
    return float(np.max(num / den))


def rel_tiles(a, b, floor, tile=16):
    """max over tile x tile blocks (of the last two axes) of max|a - b|_block / (max|b|_block + floor)."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    if a.size == 0:
        return 0.0
    if a.ndim < 2:
        return rel_rows(a, b, floor)
    a = a.reshape((-1,) + a.shape[-2:])
    b = b.reshape((-1,) + b.shape[-2:])
    worst = 0.0
    for i in range(0, a.shape[1], tile):
        for j in range(0, a.shape[2], tile):
            da = np.abs(a[:, i:i + tile, j:j + tile] - b[:, i:i + tile, j:j + tile]).reshape(a.shape[0], -1).max(axis=1)
            db = np.abs(b[:, i:i + tile, j:j + tile]).reshape(a.shape[0], -1).max(axis=1)
            worst = max(worst, float(np.max(da / (db + floor))))
    return worst


def worst_block(a, b, floor, tile=16):
    """(error, (i, j), max|b| of that block) of the worst tile — for failure messages."""
    a, b = _as2d(a), _as2d(b)
    out = (0.0, (0, 0), 0.0)
    for i in range(0, a.shape[0], tile):
        for j in range(0, a.shape[1], tile):
            da = float(np.max(np.abs(a[i:i + tile, j:j + tile] - b[i:i + tile, j:j + tile])))
            db = float(np.max(np.abs(b[i:i + tile, j:j + tile])))
            e = da / (db + floor)
            if e > out[0]:
                out = (e, (i, j), db)
    return out


def dependent_active_rows(CD, active, n, rtol=1e-9, involve_tol=1e-6):
    """Rows of the active set whose multipliers the stage KKT system determines only through its mu-regularisation: with D_a the
    control columns of the active rows (normalised), the rows that carry weight in a left null vector of D_a (y^T D_a = 0).
    Returns a boolean mask over all constraint rows (False for inactive rows and for active rows outside every dependency)."""
    CD = np.asarray(CD, float)
    active = np.asarray(active, bool)
    out = np.zeros(active.shape, bool)
    idx = np.flatnonzero(active)
    if idx.size == 0 or CD.shape[1] <= n:
        return out
    D = CD[idx][:, n:]
    nrm = np.linalg.norm(D, axis=1)
    idx, D, nrm = idx[nrm > 0], D[nrm > 0], nrm[nrm > 0]  # rows without a control part (joint limits): nu = (...) / mu directly
    if idx.size == 0:
        return out
    D = D / nrm[:, None]
    U, s, _ = np.linalg.svd(D, full_matrices=True)
    rank = int(np.sum(s > rtol * max(s[0], 1e-300))) if s.size else 0
    if rank >= idx.size:
        return out
    null = U[:, rank:]
    out[idx[np.max(np.abs(null), axis=1) > involve_tol]] = True
    return out


def rel_cols(a, b, floor):
    """max over the components (last axis) of max|a - b| / (max|b| + floor), taken over all other axes: a trajectory whose
    components live on different scales (positions ~1, forces ~500, velocities ~0.01) is held component by component."""
    a, b = _as2d(a), _as2d(b)
    if a.size == 0:
        return 0.0
    return float(np.max(np.max(np.abs(a - b), axis=0) / (np.max(np.abs(b), axis=0) + floor)))


def traj_err(xs_a, us_a, xs_b, us_b, x_floor=1e-3, u_floor=1.0):
    """Deviation of a trajectory (xs [.., nx], us [.., nu]) from a reference one, component by component: every state component against
    its own range (floor 1e-3: base positions ~1, joint velocities ~1e-2), every control component against its own range (floor 1: torques
    and forces of 1 .. 500).  States and controls are never normalised by one another."""
    return max(rel_cols(xs_a, xs_b, x_floor), rel_cols(us_a, us_b, u_floor))
