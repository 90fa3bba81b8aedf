"""Dense reference solve of the proximal LQ subproblem of one ProxDDP iteration,
assembled from the phase dumps of a backend: used to cross-check the Riccati sweep (numpy, pivoted LU)."""
import numpy as np


def proj_normal(role, z, lo, hi):
    if role == 1:
        return z, True
    if role == 2:
        return (z, True) if z > 0 else (0.0, False)
    if role == 3:
        if z < lo:
            return z - lo, True
        if z > hi:
            return z - hi, True
    return 0.0, False


def solve_dense_lq(nat, N, n, m, mu, mud, roles_lo_hi, vs_e=None, lams_e=None, multibody=True):
    """-> dict(dx[N+1][n], du[N][m], lam_new[N+1][n], nu_new[list])"""
    knots = []
    for k in range(N + 1):
        mk = m if k < N else 0
        nz = n + mk
        kn = {"H": nat.debug_get("H", k).reshape(nz, nz), "g": nat.debug_get("grad", k),
              "cval": nat.debug_get("cval", k)}
        kn["CD"] = nat.debug_get("CD", k).reshape(-1, nz) if kn["cval"].size else np.zeros((0, nz))
        if k < N:
            kn["AB"] = nat.debug_get("AB", k).reshape(n, nz)
            kn["f"] = nat.debug_get("f", k)
            E = -np.eye(n)
            if multibody:
                E[:6, :6] = nat.debug_get("E6", k).reshape(6, 6)
            kn["E"] = E
        knots.append(kn)
    # unknown layout
    off = {}
    cur = 0
    for k in range(1, N + 1):
        off[("x", k)] = cur; cur += n
    for k in range(N):
        off[("u", k)] = cur; cur += m
    for k in range(N):
        off[("l", k + 1)] = cur; cur += n
    act = []
    for k in range(N + 1):
        rl = roles_lo_hi[k]
        rows, dt = [], []
        for i, (role, lo, hi) in enumerate(rl):
            ve = 0.0 if vs_e is None else vs_e[k][i]
            pn, a = proj_normal(role, knots[k]["cval"][i] + mu * ve, lo, hi)
            if a:
                rows.append(i); dt.append(pn)
        act.append((rows, np.array(dt)))
        off[("v", k)] = cur; cur += len(rows)
    Kmat = np.zeros((cur, cur)); rhs = np.zeros(cur)
    for k in range(N + 1):
        kn = knots[k]
        mk = m if k < N else 0
        rows, dt = act[k]
        Ca = kn["CD"][rows]
        xs_ = off.get(("x", k))
        if xs_ is not None:
            r = slice(xs_, xs_ + n)
            Kmat[r, r] += kn["H"][:n, :n]
            rhs[r] -= kn["g"][:n]
            if k < N:
                Kmat[r, off[("u", k)]:off[("u", k)] + m] += kn["H"][:n, n:]
                Kmat[r, off[("l", k + 1)]:off[("l", k + 1)] + n] += kn["AB"][:, :n].T
            Kmat[r, off[("v", k)]:off[("v", k)] + len(rows)] += Ca[:, :n].T
            Kmat[r, off[("l", k)]:off[("l", k)] + n] += knots[k - 1]["E"].T
        if k < N:
            ru = slice(off[("u", k)], off[("u", k)] + m)
            Kmat[ru, ru] += kn["H"][n:, n:]
            rhs[ru] -= kn["g"][n:]
            if xs_ is not None:
                Kmat[ru, xs_:xs_ + n] += kn["H"][n:, :n]
            Kmat[ru, off[("l", k + 1)]:off[("l", k + 1)] + n] += kn["AB"][:, n:].T
            Kmat[ru, off[("v", k)]:off[("v", k)] + len(rows)] += Ca[:, n:].T
            rl_ = slice(off[("l", k + 1)], off[("l", k + 1)] + n)
            if xs_ is not None:
                Kmat[rl_, xs_:xs_ + n] += kn["AB"][:, :n]
            Kmat[rl_, ru] += kn["AB"][:, n:]
            Kmat[rl_, off[("x", k + 1)]:off[("x", k + 1)] + n] += kn["E"]
            Kmat[rl_, rl_] -= mud * np.eye(n)
            le = 0.0 if lams_e is None else lams_e[k + 1]
            rhs[rl_] -= kn["f"] + mud * le
        rv = slice(off[("v", k)], off[("v", k)] + len(rows))
        if xs_ is not None:
            Kmat[rv, xs_:xs_ + n] += Ca[:, :n]
        if k < N:
            Kmat[rv, off[("u", k)]:off[("u", k)] + m] += Ca[:, n:]
        Kmat[rv, rv] -= mu * np.eye(len(rows))
        rhs[rv] -= dt
    sol = np.linalg.solve(Kmat, rhs)
    dx = np.zeros((N + 1, n)); du = np.zeros((N, m)); lam = np.zeros((N + 1, n)); nus = []
    for k in range(1, N + 1):
        dx[k] = sol[off[("x", k)]:off[("x", k)] + n]
        lam[k] = sol[off[("l", k)]:off[("l", k)] + n]
    for k in range(N):
        du[k] = sol[off[("u", k)]:off[("u", k)] + m]
    for k in range(N + 1):
        rows, _ = act[k]
        full = np.zeros(len(roles_lo_hi[k]))
        full[rows] = sol[off[("v", k)]:off[("v", k)] + len(rows)]
        nus.append(full)
    return {"dx": dx, "du": du, "lam": lam, "nu": nus, "cond": np.linalg.cond(Kmat)}


def stage_rows(desc, params):
    """(role, lo, hi) per constraint row from a lowered stage table."""
    rows = []
    nt = desc[5]
    for t in range(nt):
        w = desc[8 + 8 * t: 16 + 8 * t]
        role, dim, woff = w[1], w[2], w[6]
        if role == 0:
            continue
        for i in range(dim):
            if role == 3:
                rows.append((3, params[woff + i], params[woff + dim + i]))
            else:
                rows.append((int(role), 0.0, 0.0))
    return rows
