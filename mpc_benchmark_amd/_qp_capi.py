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
