"""QPs with the structure of the reference's whole-body inverse-dynamics problems (QP_utils.py:437-575): variables
(da, df, tau), dynamics + contact-acceleration equalities, wrench-cone inequalities, optional torque box."""
import numpy as np

CMIN = lambda mu, L, W: np.array([[-1, 0, mu, 0, 0, 0], [1, 0, mu, 0, 0, 0], [-1, 0, mu, 0, 0, 0], [1, 0, mu, 0, 0, 0], [0, 0, 1, 0, 0, 0],
                                  [0, 0, W, -1, 0, 0], [0, 0, W, 1, 0, 0], [0, 0, L, 0, -1, 0], [0, 0, L, 0, 1, 0]], dtype=float)


def id_qp(rng, nv=28, nk=2, weights=(1.0, 1e-3), mu=0.8, L=0.1, W=0.075, torque_limit=None, contact=(True, True)):
    """-> dict(H, g, A, b, C, l, u[, l_box, u_box]) of one IDSolver_ulim-like problem with random (physically scaled) data."""
    fs = 6
    n, neq, nin = 2 * nv - 6 + fs * nk, nv + fs * nk, 9 * nk
    R = rng.normal(size=(nv, nv))
    M = R @ R.T / nv + np.diag(rng.uniform(0.5, 30.0, nv))   # mass-matrix like: SPD, mixed scales
    Jc = np.zeros((fs * nk, nv))
    for i in range(nk):
        if contact[i]:
            Jc[fs * i:fs * (i + 1)] = rng.normal(size=(fs, nv)) * 0.5
    a = rng.normal(size=nv) * 0.5
    forces = np.zeros(fs * nk)
    for i in range(nk):
        if contact[i]:
            forces[fs * i:fs * (i + 1)] = np.array([rng.normal() * 20, rng.normal() * 20, 450 + rng.normal() * 50, rng.normal() * 5, rng.normal() * 5, rng.normal()])
    nle = rng.normal(size=nv) * 30
    gamma = rng.normal(size=fs * nk) * 0.3
    for i in range(nk):
        if not contact[i]:
            gamma[fs * i:fs * (i + 1)] = 0.0  # a foot in the air contributes empty rows (0 = 0), as in QP_utils.py:520-530
    S = np.zeros((nv, nv - 6)); S[6:] = np.eye(nv - 6)
    A = np.zeros((neq, n)); b = np.zeros(neq)
    A[:nv, :nv] = M; A[:nv, nv:nv + fs * nk] = -Jc.T; A[:nv, nv + fs * nk:] = -S; A[nv:, :nv] = Jc
    b[:nv] = -nle - M @ a + Jc.T @ forces
    b[nv:] = -gamma - Jc @ a
    C = np.zeros((nin, n)); l = np.zeros(nin)
    cm = CMIN(mu, L, W)
    for i in range(nk):
        if contact[i]:
            f = forces[fs * i:fs * (i + 1)]
            l[9 * i:9 * (i + 1)] = -cm @ f          # C (f + df) >= 0  <=>  C df >= -C f
            C[9 * i:9 * (i + 1), nv + fs * i:nv + fs * (i + 1)] = cm
    H = np.zeros((n, n)); H[:nv, :nv] = np.eye(nv) * weights[0]; H[nv:nv + fs * nk, nv:nv + fs * nk] = np.eye(fs * nk) * weights[1]
    out = dict(H=H, g=np.zeros(n), A=A, b=b, C=C, l=l, u=np.full(nin, 1e5))
    if torque_limit is not None:
        lb = np.full(n, -1e5); ub = np.full(n, 1e5)
        lb[nv + fs * nk:] = -torque_limit; ub[nv + fs * nk:] = torque_limit
        out["l_box"], out["u_box"] = lb, ub
    return out


def kkt_residuals(q, x, y, z, zb=None):
    """Optimality of a convex QP, independent of how it was solved: stationarity, primal feasibility, sign and
    complementarity of the inequality multipliers (z > 0 on an upper bound, < 0 on a lower bound)."""
    H, g, A, b, C, l, u = (q[k] for k in ("H", "g", "A", "b", "C", "l", "u"))
    stat = H @ x + g + A.T @ y + C.T @ z + (zb if zb is not None else 0.0)
    s = C @ x
    prim = max(np.max(np.abs(A @ x - b)), np.max(np.maximum(s - u, 0)), np.max(np.maximum(l - s, 0)))
    comp = max(np.max(np.abs(np.maximum(z, 0) * (u - s))), np.max(np.abs(np.minimum(z, 0) * (s - l))))
    if zb is not None:
        prim = max(prim, np.max(np.maximum(x - q["u_box"], 0)), np.max(np.maximum(q["l_box"] - x, 0)))
        comp = max(comp, np.max(np.abs(np.maximum(zb, 0) * (q["u_box"] - x))), np.max(np.abs(np.minimum(zb, 0) * (x - q["l_box"]))))
    return float(np.max(np.abs(stat))), float(prim), float(comp)
