"""ctypes binding of include/mpc_qp_abi.h (batched dense QP, "next" row N3).  The product library is the HIP one
(``_capi.load_hip_library()``); there is no CPU fallback — tests pass the oracle library explicitly as the checker."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi as K


class QpDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("batch", "n", "neq", "nin", "box", "device")]


class QpSettings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("eps_abs", "rho", "mu_eq", "mu_in", "mu_min_eq", "mu_min_in", "mu_update_factor", "alpha_bcl", "beta_bcl")] + \
               [(n, C.c_int32) for n in ("max_iter", "max_iter_in", "warm_start", "reserved")]


class QpInfo(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("prim_res", "dual_res", "mu_eq", "mu_in")] + \
               [(n, C.c_int32) for n in ("iters", "iters_in", "status", "n_active")]


_DP = C.POINTER(C.c_double)
_bound = set()


def _bind(lib):
    if id(lib) in _bound:
        return lib
    lib.mpc_qp_create.restype = C.c_int
    lib.mpc_qp_create.argtypes = [C.POINTER(QpDims), C.POINTER(C.c_void_p)]
    lib.mpc_qp_destroy.restype = None
    lib.mpc_qp_destroy.argtypes = [C.c_void_p]
    lib.mpc_qp_last_error.restype = C.c_char_p
    lib.mpc_qp_last_error.argtypes = [C.c_void_p]
    lib.mpc_qp_default_settings.restype = None
    lib.mpc_qp_default_settings.argtypes = [C.POINTER(QpSettings)]
    lib.mpc_qp_solve.restype = C.c_int
    lib.mpc_qp_solve.argtypes = [C.c_void_p, C.POINTER(QpSettings)] + [_DP] * 13 + [C.POINTER(QpInfo)]
    _IP = C.POINTER(C.c_int32)
    lib.mpc_qp_set_model.restype = C.c_int
    lib.mpc_qp_set_model.argtypes = [C.c_void_p, _IP, C.c_int32, _DP, C.c_int32]
    lib.mpc_qp_solve_id.restype = C.c_int
    lib.mpc_qp_solve_id.argtypes = [C.c_void_p, C.POINTER(QpSettings), C.c_int32, _IP, _DP, _DP, C.c_double, _DP, _DP, _DP, _IP,
                                    _DP, _DP, _DP, C.POINTER(QpInfo), _DP, _DP, _DP, _DP]
    lib.mpc_qp_solve_ikid.restype = C.c_int
    lib.mpc_qp_solve_ikid.argtypes = [C.c_void_p, C.POINTER(QpSettings), C.c_int32, _IP, C.c_int32, C.c_int32, _DP, _DP, _DP, _DP, _DP, _DP, _DP, _DP, _IP,
                                      _DP, _DP, _DP, _DP, C.POINTER(QpInfo), _DP, _DP, _DP, _DP, _DP, _DP]
    lib.mpc_qp_low_level_steps.restype = C.c_int
    lib.mpc_qp_low_level_steps.argtypes = [C.c_void_p, C.POINTER(QpSettings), C.c_void_p, C.c_void_p, C.c_int32, _IP, _DP, _DP, C.c_double, _IP, _DP,
                                           _DP, C.c_int32, C.c_double, _DP, _DP, _DP, _DP, C.POINTER(QpInfo)]
    _bound.add(id(lib))
    return lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(_DP)


class BatchedQP:
    """B dense QPs of one shape: min 1/2 x'Hx + g'x, Ax = b, l <= Cx <= u (, l_box <= x <= u_box)."""

    def __init__(self, batch, n, neq, nin, box=False, library=None, device=0):
        self.lib = _bind(library if library is not None else K.load_hip_library())
        self.dims = QpDims(int(batch), int(n), int(neq), int(nin), int(bool(box)), int(device))
        h = C.c_void_p()
        if self.lib.mpc_qp_create(C.byref(self.dims), C.byref(h)) != 0:
            raise RuntimeError("mpc_qp_create failed")
        self._h = h
        self.settings = QpSettings()
        self.lib.mpc_qp_default_settings(C.byref(self.settings))

    def __del__(self):
        if getattr(self, "_h", None):
            self.lib.mpc_qp_destroy(self._h)
            self._h = None

    def solve(self, H, g, A, b, C_, l, u, l_box=None, u_box=None):
        d = self.dims
        B, n, neq, nin = d.batch, d.n, d.neq, d.nin

        def arr(a, shape):
            if a is None:
                return None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), shape))
            return a
        H = arr(H, (B, n, n)); g = arr(g, (B, n)); A = arr(A, (B, neq, n)); b = arr(b, (B, neq))
        C_ = arr(C_, (B, nin, n)); l = arr(l, (B, nin)); u = arr(u, (B, nin))
        lb = arr(l_box, (B, n)) if d.box else None
        ub = arr(u_box, (B, n)) if d.box else None
        x = np.zeros((B, n)); y = np.zeros((B, neq)); z = np.zeros((B, nin)); zb = np.zeros((B, n))
        info = (QpInfo * B)()
        rc = self.lib.mpc_qp_solve(self._h, C.byref(self.settings), _dp(H), _dp(g), _dp(A), _dp(b), _dp(C_), _dp(l), _dp(u), _dp(lb), _dp(ub),
                                   _dp(x), _dp(y), _dp(z), _dp(zb), info)
        if rc != 0:
            raise RuntimeError("mpc_qp_solve: " + self.lib.mpc_qp_last_error(self._h).decode())
        return x, y, z, zb, list(info)

    # ---- on-device assembly of the inverse-dynamics QP (mpc_qp_set_model / mpc_qp_solve_id) ----
    def set_model(self, itab, dtab):
        itab = np.ascontiguousarray(itab, dtype=np.int32); dtab = np.ascontiguousarray(dtab, dtype=np.float64)
        if self.lib.mpc_qp_set_model(self._h, itab.ctypes.data_as(C.POINTER(C.c_int32)), itab.size, _dp(dtab), dtab.size) != 0:
            raise RuntimeError("mpc_qp_set_model: " + self.lib.mpc_qp_last_error(self._h).decode())
        self._nqv, self._nv = int(itab[1]) + int(itab[2]), int(itab[2])

    @staticmethod
    def _cone_pair(cone, cone_l):
        cone = np.ascontiguousarray(cone, dtype=np.float64)
        if cone.shape != (9, 6):
            raise ValueError("cone must be 9 x 6")
        cone_l = cone if cone_l is None else np.ascontiguousarray(cone_l, dtype=np.float64)
        if cone_l.shape != (9, 6):
            raise ValueError("cone_l must be 9 x 6")
        return np.ascontiguousarray(np.stack([cone, cone_l]))

    def solve_id(self, frames, weights, cone, kd, xrob, acc, forces, contact_states, return_matrices=False, cone_l=None):
        """-> x, y, z, info (, (A, b, C, l) as assembled on the device).  ``cone``: the rows of C; ``cone_l``: the rows that form
        l = - cone_l f (default: the same)."""
        d = self.dims
        B, n, neq, nin = d.batch, d.n, d.neq, d.nin
        frames = np.ascontiguousarray(frames, dtype=np.int32); nk = frames.size
        weights = np.ascontiguousarray(weights, dtype=np.float64)[:2].copy()
        cone = self._cone_pair(cone, cone_l)
        xrob = np.ascontiguousarray(np.broadcast_to(np.asarray(xrob, dtype=np.float64), (B, self._nqv)))
        nv = self._nv
        acc = np.ascontiguousarray(np.broadcast_to(np.asarray(acc, dtype=np.float64), (B, nv)))
        forces = np.ascontiguousarray(np.broadcast_to(np.asarray(forces, dtype=np.float64), (B, 6 * nk)))
        cs = np.ascontiguousarray(np.broadcast_to(np.asarray(contact_states, dtype=np.int32), (B, nk)))
        x = np.zeros((B, n)); y = np.zeros((B, neq)); z = np.zeros((B, nin))
        mats = (np.zeros((B, neq, n)), np.zeros((B, neq)), np.zeros((B, nin, n)), np.zeros((B, nin))) if return_matrices else (None,) * 4
        info = (QpInfo * B)()
        IP = C.POINTER(C.c_int32)
        rc = self.lib.mpc_qp_solve_id(self._h, C.byref(self.settings), nk, frames.ctypes.data_as(IP), _dp(weights), _dp(cone), float(kd),
                                      _dp(xrob), _dp(acc), _dp(forces), cs.ctypes.data_as(IP), _dp(x), _dp(y), _dp(z), info,
                                      _dp(mats[0]), _dp(mats[1]), _dp(mats[2]), _dp(mats[3]))
        if rc != 0:
            raise RuntimeError("mpc_qp_solve_id: " + self.lib.mpc_qp_last_error(self._h).decode())
        if return_matrices:
            return x, y, z, list(info), mats
        return x, y, z, list(info)

    def low_level_steps(self, plan, sim, frames, weights, cone, kd, contact_states, tau_max, steps, dt, x=None, cone_l=None):
        """mpc_qp_low_level_steps: ``steps`` periods of the kinodynamic low-level loop (feedback terms of the plan's knot 0 -> inverse-dynamics QP ->
        clamped torque -> simulator step) without the host in between.  ``plan``, ``sim``: NativeSolver handles of the same library.
        -> x_prev, x, tau, forces, info (states before / after the last period, torques and forces of the last period)."""
        d = self.dims
        B = d.batch
        frames = np.ascontiguousarray(frames, dtype=np.int32); nk = frames.size
        weights = np.ascontiguousarray(weights, dtype=np.float64)[:2].copy()
        cone = self._cone_pair(cone, cone_l)
        nv = self._nv
        cs = np.ascontiguousarray(np.broadcast_to(np.asarray(contact_states, dtype=np.int32), (B, nk)))
        tau_max = np.ascontiguousarray(tau_max, dtype=np.float64)
        if tau_max.size != nv - 6:
            raise ValueError("tau_max must have nv - 6 entries")
        if x is not None:
            x = np.ascontiguousarray(np.broadcast_to(np.asarray(x, dtype=np.float64), (B, self._nqv)))
        x_prev = np.zeros((B, self._nqv)); x_out = np.zeros((B, self._nqv)); tau = np.zeros((B, nv - 6)); forces = np.zeros((B, 6 * nk))
        info = (QpInfo * B)()
        IP = C.POINTER(C.c_int32)
        rc = self.lib.mpc_qp_low_level_steps(self._h, C.byref(self.settings), plan._h, sim._h, nk, frames.ctypes.data_as(IP), _dp(weights), _dp(cone), float(kd),
                                             cs.ctypes.data_as(IP), _dp(tau_max), _dp(x), int(steps), float(dt), _dp(x_prev), _dp(x_out), _dp(tau), _dp(forces), info)
        if rc != 0:
            raise RuntimeError("mpc_qp_low_level_steps: " + self.lib.mpc_qp_last_error(self._h).decode())
        return x_prev, x_out, tau, forces, list(info)

    def solve_ikid(self, frames, base_frame, torso_frame, weights, gains, cone, l_box, u_box, xrob, ik, forces, contact_states, return_matrices=False,
                   cone_l=None):
        """mpc_qp_solve_ikid -> x, y, z, z_box, info (, (H, g, A, b, C, l) as assembled by the library)."""
        d = self.dims
        B, n, neq, nin = d.batch, d.n, d.neq, d.nin
        frames = np.ascontiguousarray(frames, dtype=np.int32); nk = frames.size
        nv = self._nv
        f64 = lambda a, shape=None: np.ascontiguousarray(a if shape is None else np.broadcast_to(np.asarray(a, dtype=np.float64), shape), dtype=np.float64)
        weights, gains, cone, l_box, u_box = f64(weights), f64(gains), self._cone_pair(cone, cone_l), f64(l_box), f64(u_box)
        if weights.size != 5 or gains.size != 2 * nv * nv + 90 or l_box.size != n or u_box.size != n:
            raise ValueError("solve_ikid: weights[5], gains[2 nv^2 + 90], cone[9][6], l_box / u_box [n] expected")
        xrob = f64(xrob, (B, self._nqv)); ik = f64(ik, (B, 2 * nv + 42)); forces = f64(forces, (B, 6 * nk))
        cs = np.ascontiguousarray(np.broadcast_to(np.asarray(contact_states, dtype=np.int32), (B, nk)))
        x = np.zeros((B, n)); y = np.zeros((B, neq)); z = np.zeros((B, nin)); zb = np.zeros((B, n))
        mats = (np.zeros((B, n, n)), np.zeros((B, n)), np.zeros((B, neq, n)), np.zeros((B, neq)), np.zeros((B, nin, n)), np.zeros((B, nin))) if return_matrices else (None,) * 6
        info = (QpInfo * B)()
        IP = C.POINTER(C.c_int32)
        rc = self.lib.mpc_qp_solve_ikid(self._h, C.byref(self.settings), nk, frames.ctypes.data_as(IP), int(base_frame), int(torso_frame), _dp(weights), _dp(gains),
                                        _dp(cone), _dp(l_box), _dp(u_box), _dp(xrob), _dp(ik), _dp(forces), cs.ctypes.data_as(IP),
                                        _dp(x), _dp(y), _dp(z), _dp(zb), info, *[_dp(m_) for m_ in mats])
        if rc != 0:
            raise RuntimeError("mpc_qp_solve_ikid: " + self.lib.mpc_qp_last_error(self._h).decode())
        return (x, y, z, zb, list(info), mats) if return_matrices else (x, y, z, zb, list(info))
