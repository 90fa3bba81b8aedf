"""ctypes binding of the C-ABI declared in ``include/mpc_abi.h``.

The product loads exactly one library: ``mpc_benchmark_amd/csrc/libmpc_hip.so`` (hand-written HIP for
gfx950).  There is no CPU fallback: if the library is missing or fails to load, ``load_hip_library``
raises.  ``bind_library`` is the generic binder; tests use it to bind the CPU oracle
(``oracle/libmpc_oracle.so``) as the *checker* — the package itself never does.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPC_HIP_LIBRARY: developer override pointing at another build of the SAME HIP library (kernel tuning variants)
HIP_LIBRARY_PATH = os.environ.get("MPC_HIP_LIBRARY") or os.path.join(_HERE, "csrc", "libmpc_hip.so")

ABI_VERSION = 3

# constants mirrored from include/mpc_abi.h
SPACE_VECTOR, SPACE_MULTIBODY = 0, 1
JOINT_FREEFLYER, JOINT_RX, JOINT_RY, JOINT_RZ = 0, 1, 2, 3
DYN_NONE, DYN_CENTROIDAL_EULER, DYN_MULTIBODY_CONSTRAINT_SEMIEULER, DYN_KINODYNAMICS_SEMIEULER = 0, 1, 2, 3
(TERM_STATE_ERROR, TERM_CONTROL_ERROR, TERM_FRAME_PLACEMENT, TERM_FRAME_TRANSLATION, TERM_FRAME_VELOCITY,
 TERM_COM_TRANSLATION, TERM_CENTROIDAL_MOMENTUM, TERM_CONTACT_FORCE, TERM_MB_WRENCH_CONE,
 TERM_CENTROIDAL_WRENCH_CONE, TERM_CENTROIDAL_LIN_ACC, TERM_CENTROIDAL_ANG_ACC,
 TERM_CENTROIDAL_MOMENTUM_DER) = range(1, 14)
ROLE_COST, ROLE_EQUALITY, ROLE_NEG_ORTHANT, ROLE_BOX = 0, 1, 2, 3
TERM_FLAG_DIAG_WEIGHT = 1
STAGE_HEADER_WORDS, TERM_WORDS = 8, 8


class MpcDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "horizon", "batch", "space", "nx", "ndx", "nu", "nc_max", "max_stage_ints", "max_stage_doubles", "device")]


class MpcOptions(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "tol", "mu_init", "dyn_al_scale", "reg_init", "ls_armijo_c1", "ls_alpha_min",
        "bcl_prim_alpha", "bcl_prim_beta", "bcl_dual_alpha", "bcl_dual_beta",
        "bcl_mu_update_factor", "bcl_mu_lower_bound", "inner_tol0", "prim_tol0", "corrector_prim_tol")] + [(n, C.c_int32) for n in (
        "max_iters", "max_al_iters", "force_initial_condition", "rollout_linear", "ls_max_steps",
        "num_threads", "riccati_legs", "forward_mode", "refine_appended_knot", "corrector_window")]


def default_options(tol=1e-5, mu_init=1e-8):
    """Defaults of the knobs the scripts do not touch (documented in DESIGN.md; upstream values unpinned)."""
    o = MpcOptions()
    o.tol, o.mu_init, o.dyn_al_scale, o.reg_init = tol, mu_init, 1e-3, 1e-9
    o.ls_armijo_c1, o.ls_alpha_min = 1e-4, 1e-7
    o.bcl_prim_alpha, o.bcl_prim_beta, o.bcl_dual_alpha, o.bcl_dual_beta = 0.1, 0.9, 1.0, 1.0
    o.bcl_mu_update_factor, o.bcl_mu_lower_bound = 0.01, 1e-8
    o.inner_tol0, o.prim_tol0 = 1.0, 1.0
    o.max_iters, o.max_al_iters = 1000, 100
    o.force_initial_condition, o.rollout_linear, o.ls_max_steps = 0, 0, 8
    o.num_threads, o.riccati_legs, o.forward_mode = 1, 1, 0
    o.refine_appended_knot = 0
    o.corrector_prim_tol, o.corrector_window = 0.0, 0
    return o


class MpcWalkConfig(C.Structure):
    """mpc_walk_config of include/mpc_abi.h (reference generation in the library)"""
    _fields_ = [(n, C.c_int32) for n in ("T_ss", "T_ds", "frame_lf", "frame_rf", "off_lf", "off_rf", "off_xref_z", "toff_com", "toff_lf", "toff_rf")] + [
        ("swing_apex", C.c_double), ("t_left", C.c_double * 3), ("t_right", C.c_double * 3), ("rot_diff", C.c_double * 9), ("com0", C.c_double * 3),
        ("feet_z0", C.c_double), ("xref_z0", C.c_double), ("z_follow", C.c_double), ("lf0", C.c_double * 12), ("rf0", C.c_double * 12), ("floor_z", C.c_double)]


class MpcStats(C.Structure):
    _fields_ = [("num_iters", C.c_int32), ("converged", C.c_int32), ("al_iters", C.c_int32), ("ls_steps", C.c_int32),
                ("traj_cost", C.c_double), ("merit", C.c_double), ("prim_infeas", C.c_double),
                ("dual_infeas", C.c_double), ("mu", C.c_double), ("alpha", C.c_double)]


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int32)

_SIGNATURES = {
    "mpc_abi_version": (C.c_int, []),
    "mpc_backend_name": (C.c_char_p, []),
    "mpc_create": (C.c_int, [C.POINTER(MpcDims), C.POINTER(C.c_void_p)]),
    "mpc_destroy": (None, [C.c_void_p]),
    "mpc_last_error": (C.c_char_p, [C.c_void_p]),
    "mpc_set_options": (C.c_int, [C.c_void_p, C.POINTER(MpcOptions)]),
    "mpc_set_model": (C.c_int, [C.c_void_p, _IP, C.c_int32, _DP, C.c_int32]),
    "mpc_set_stage": (C.c_int, [C.c_void_p, C.c_int32, _IP, C.c_int32, _DP, C.c_int32]),
    "mpc_update_stage_params": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _DP, C.c_int32]),
    "mpc_update_stage_params_batch": (C.c_int, [C.c_void_p, C.c_int32, _IP, _IP, _IP, _DP]),
    "mpc_cycle": (C.c_int, [C.c_void_p, _IP, C.c_int32, _DP, C.c_int32]),
    "mpc_walk_init": (C.c_int, [C.c_void_p, C.POINTER(MpcWalkConfig)]),
    "mpc_walk_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _DP]),
    "mpc_walk_get_state": (C.c_int, [C.c_void_p, _DP]),
    "mpc_walk_set_state": (C.c_int, [C.c_void_p, _DP]),
    "mpc_set_x0": (C.c_int, [C.c_void_p, _DP]),
    "mpc_simulate": (C.c_int, [C.c_void_p, C.c_int32, C.c_double]),
    "mpc_simulate_push": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, _DP]),
    "mpc_simulate_torque": (C.c_int, [C.c_void_p, _DP, _DP, C.c_int32, C.c_double, _DP]),
    "mpc_set_tick_reuse": (C.c_int, [C.c_void_p, C.c_int32]),
    "mpc_enable_instance_params": (C.c_int, [C.c_void_p]),
    "mpc_update_instance_params_batch": (C.c_int, [C.c_void_p, C.c_int32, _IP, _IP, _IP, _IP, _DP]),
    "mpc_set_failure_policy": (C.c_int, [C.c_void_p, C.c_int32]),
    "mpc_revive_instance": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mpc_poll": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mpc_get_x0": (C.c_int, [C.c_void_p, _DP]),
    "mpc_setup": (C.c_int, [C.c_void_p]),
    "mpc_run": (C.c_int, [C.c_void_p, _DP, _DP, C.POINTER(MpcStats)]),
    "mpc_run_shifted": (C.c_int, [C.c_void_p, C.POINTER(MpcStats)]),
    "mpc_run_shifted_async": (C.c_int, [C.c_void_p]),
    "mpc_wait": (C.c_int, [C.c_void_p, C.POINTER(MpcStats)]),
    "mpc_wait_state": (C.c_int, [C.c_void_p, C.POINTER(MpcStats), _DP]),
    "mpc_get_gain": (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP]),
    "mpc_state_size": (C.c_int64, [C.c_void_p]),
    "mpc_get_state": (C.c_int64, [C.c_void_p, _DP, C.c_int64]),
    "mpc_set_state": (C.c_int, [C.c_void_p, _DP, C.c_int64]),
    "mpc_get_results": (C.c_int, [C.c_void_p, _DP, _DP, _DP, _DP, _DP, _DP]),
    "mpc_get_stage_data": (C.c_int, [C.c_void_p, C.c_int32, _DP, _DP]),
    "mpc_debug_get": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32, _DP, C.c_int32]),
    "mpc_debug_evaluate": (C.c_int, [C.c_void_p, _DP, _DP]),
    "mpc_profile": (C.c_int, [C.c_void_p, C.c_int32]),
    "mpc_profile_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "mpc_kernel_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(C.c_int32)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def bind_library(path):
    """dlopen ``path`` and attach the argument/return types of every entry point of mpc_abi.h."""
    lib = C.CDLL(path, mode=getattr(os, "RTLD_LOCAL", 0) | getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.mpc_abi_version() != ABI_VERSION:
        raise RuntimeError("%s: ABI version %d, expected %d" % (path, lib.mpc_abi_version(), ABI_VERSION))
    return lib


_hip_lib = None


def load_hip_library():
    """Load the HIP product library. Fails loudly — there is deliberately no CPU fallback."""
    global _hip_lib
    if _hip_lib is None:
        if not os.path.exists(HIP_LIBRARY_PATH):
            raise RuntimeError(
                "HIP solver library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mpc_benchmark_amd/csrc`). No CPU fallback exists." % HIP_LIBRARY_PATH)
        _hip_lib = bind_library(HIP_LIBRARY_PATH)
    return _hip_lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _dp(a):
    return None if a is None else a.ctypes.data_as(_DP)


class NativeSolver:
    """Thin object wrapper over one ``mpc_solver*`` handle of a bound library."""

    def __init__(self, lib, dims: MpcDims):
        self.lib = lib
        self.dims = dims
        self._h = C.c_void_p()
        rc = lib.mpc_create(C.byref(dims), C.byref(self._h))
        if rc != 0:
            raise RuntimeError("mpc_create failed (rc=%d)" % rc)
        self.backend = lib.mpc_backend_name().decode()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.mpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError("%s failed: %s" % (what, self.lib.mpc_last_error(self._h).decode(errors="replace")))
        return rc

    # -- problem upload ------------------------------------------------------------------------
    def set_options(self, opt: MpcOptions):
        self._check(self.lib.mpc_set_options(self._h, C.byref(opt)), "mpc_set_options")

    def set_model(self, itab, dtab):
        itab, dtab = _i32(itab), _f64(dtab)
        self._check(self.lib.mpc_set_model(self._h, itab.ctypes.data_as(_IP), itab.size, _dp(dtab), dtab.size), "mpc_set_model")

    def set_stage(self, k, desc, params):
        desc, params = _i32(desc), _f64(params)
        self._check(self.lib.mpc_set_stage(self._h, k, desc.ctypes.data_as(_IP), desc.size, _dp(params), params.size), "mpc_set_stage")

    def update_stage_params(self, k, offset, vals):
        vals = _f64(vals).ravel()
        self._check(self.lib.mpc_update_stage_params(self._h, k, offset, _dp(vals), vals.size), "mpc_update_stage_params")

    def update_stage_params_batch(self, updates):
        """``updates``: iterable of (k, offset, values) — one library call, one stream synchronisation."""
        updates = list(updates)
        if not updates:
            return
        ks = _i32([u[0] for u in updates])
        offs = _i32([u[1] for u in updates])
        chunks = [_f64(u[2]).ravel() for u in updates]
        lens = _i32([c.size for c in chunks])
        vals = _f64(np.concatenate(chunks))
        self._check(self.lib.mpc_update_stage_params_batch(self._h, len(updates), ks.ctypes.data_as(_IP), offs.ctypes.data_as(_IP),
                                                           lens.ctypes.data_as(_IP), _dp(vals)), "mpc_update_stage_params_batch")

    def cycle(self, desc, params):
        desc, params = _i32(desc), _f64(params)
        self._check(self.lib.mpc_cycle(self._h, desc.ctypes.data_as(_IP), desc.size, _dp(params), params.size), "mpc_cycle")

    def set_x0(self, x0):
        if x0 is None:  # perfect-model feedback (see mpc_abi.h)
            self._check(self.lib.mpc_set_x0(self._h, None), "mpc_set_x0")
            return
        x0 = _f64(x0)
        x0 = np.ascontiguousarray(np.broadcast_to(x0.reshape(-1, self.dims.nx), (self.dims.batch, self.dims.nx)))
        self._check(self.lib.mpc_set_x0(self._h, _dp(x0)), "mpc_set_x0")

    def poll(self):
        """-> (ticks in flight, how many of them have finished on the device); does not block."""
        a, b = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.mpc_poll(self._h, C.byref(a), C.byref(b)), "mpc_poll")
        return a.value, b.value

    def set_tick_reuse(self, on):
        """MPC ticks: keep the records of the accepted full step for the next tick (see mpc_abi.h)."""
        self._check(self.lib.mpc_set_tick_reuse(self._h, int(bool(on))), "mpc_set_tick_reuse")

    def simulate(self, substeps, dt):
        """N2: integrate knot 0's dynamics under u = us[0] - K0 difference(x, xs[0]); the result is the next measured state."""
        self._check(self.lib.mpc_simulate(self._h, int(substeps), float(dt)), "mpc_simulate")

    def simulate_push(self, substeps, dt, f_ext):
        """``simulate`` with a world-frame force at the base origin of every instance: f_ext (B, 3) or (3,) (mpc_simulate_push)."""
        f = np.ascontiguousarray(np.broadcast_to(_f64(f_ext).reshape(-1, 3), (self.dims.batch, 3)))
        self._check(self.lib.mpc_simulate_push(self._h, int(substeps), float(dt), _dp(f)), "mpc_simulate_push")

    def simulate_torque(self, x, tau, substeps, dt, wrenches=False):
        """Torque-driven stand-in for ``BulletRobot.execute`` (mpc_simulate_torque): x (B, nx) or None (continue from the measured
        states), tau (B, nu).  -> the new measured states (B, nx) [, contact wrenches (B, 2, 6)]."""
        d = self.dims
        xa = None if x is None else np.ascontiguousarray(np.broadcast_to(_f64(x).reshape(-1, d.nx), (d.batch, d.nx)))
        ta = np.ascontiguousarray(np.broadcast_to(_f64(tau).reshape(-1, d.nu), (d.batch, d.nu)))
        wr = np.zeros((d.batch, 2, 6)) if wrenches else None
        self._check(self.lib.mpc_simulate_torque(self._h, _dp(xa), _dp(ta), int(substeps), float(dt), _dp(wr)), "mpc_simulate_torque")
        x0 = self.get_x0()
        return (x0, wr) if wrenches else x0

    def get_x0(self):
        x0 = np.zeros((self.dims.batch, self.dims.nx))
        self._check(self.lib.mpc_get_x0(self._h, _dp(x0)), "mpc_get_x0")
        return x0

    def setup(self):
        self._check(self.lib.mpc_setup(self._h), "mpc_setup")

    # -- solve ----------------------------------------------------------------------------------
    def _bcast(self, a, shape):
        a = _f64(a)
        if a.shape != shape:
            a = np.ascontiguousarray(np.broadcast_to(a.reshape(shape[1:]), shape))
        return a

    def run(self, xs, us):
        d = self.dims
        xs = self._bcast(xs, (d.batch, d.horizon + 1, d.nx))
        us = self._bcast(us, (d.batch, d.horizon, d.nu))
        stats = (MpcStats * d.batch)()
        self._check(self.lib.mpc_run(self._h, _dp(xs), _dp(us), stats), "mpc_run")
        return list(stats)

    def run_shifted(self):
        stats = (MpcStats * self.dims.batch)()
        self._check(self.lib.mpc_run_shifted(self._h, stats), "mpc_run_shifted")
        return list(stats)

    def run_shifted_async(self):
        self._check(self.lib.mpc_run_shifted_async(self._h), "mpc_run_shifted_async")

    def wait(self):
        stats = (MpcStats * self.dims.batch)()
        self._check(self.lib.mpc_wait(self._h, stats), "mpc_wait")
        return list(stats)

    def wait_state(self):
        """-> (stats, x_next[B][nx]): ``wait`` plus xs[1] of every instance after the completed tick (mpc_wait_state)."""
        stats = (MpcStats * self.dims.batch)()
        xn = np.zeros((self.dims.batch, self.dims.nx))
        self._check(self.lib.mpc_wait_state(self._h, stats, _dp(xn)), "mpc_wait_state")
        return list(stats), xn

    def enable_instance_params(self):
        """Every instance gets its own copy of the stage PARAMETER tables (mpc_enable_instance_params)."""
        self._check(self.lib.mpc_enable_instance_params(self._h), "mpc_enable_instance_params")

    # -- reference generation in the library (mpc_walk_*) ------------------------------------------------------------------------
    def walk_init(self, cfg: "MpcWalkConfig"):
        self._check(self.lib.mpc_walk_init(self._h, C.byref(cfg)), "mpc_walk_init")

    def walk_update(self, takeoff_RF, takeoff_LF, land_RF, land_LF, forward=None):
        """One tick of the generator for every instance, BEFORE ``cycle``.  ``forward``: (t_left[3], t_right[3], swing_apex) = updateForward."""
        fw = None
        if forward is not None:
            fw = _f64(np.concatenate([np.asarray(forward[0], dtype=float), np.asarray(forward[1], dtype=float), [float(forward[2])]]))
        self._check(self.lib.mpc_walk_update(self._h, int(takeoff_RF), int(takeoff_LF), int(land_RF), int(land_LF),
                                             fw.ctypes.data_as(_DP) if fw is not None else None), "mpc_walk_update")

    def walk_get_state(self):
        """-> [B, 4, 12]: start / final pose of the left foot, start / final pose of the right foot (R row-major, p)."""
        out = np.zeros((self.dims.batch, 4, 12))
        self._check(self.lib.mpc_walk_get_state(self._h, out.ctypes.data_as(_DP)), "mpc_walk_get_state")
        return out

    def walk_set_state(self, plan):
        plan = _f64(plan).reshape(-1)
        self._check(self.lib.mpc_walk_set_state(self._h, plan.ctypes.data_as(_DP)), "mpc_walk_set_state")

    def update_instance_params_batch(self, patches):
        """``patches``: iterable of (instance, stage k, offset, values)."""
        patches = list(patches)
        if not patches:
            return
        insts = _i32([p[0] for p in patches]); ks = _i32([p[1] for p in patches]); offs = _i32([p[2] for p in patches])
        vals = [_f64(p[3]).reshape(-1) for p in patches]
        lens = _i32([v.size for v in vals])
        flat = np.ascontiguousarray(np.concatenate(vals))
        self._check(self.lib.mpc_update_instance_params_batch(self._h, len(patches), insts.ctypes.data_as(_IP), ks.ctypes.data_as(_IP),
                                                              offs.ctypes.data_as(_IP), lens.ctypes.data_as(_IP), _dp(flat)), "mpc_update_instance_params_batch")

    def update_instance_params_arrays(self, insts, ks, offsets, lens, vals):
        """The same from prepared arrays (int32 index arrays, float64 values): no per-patch Python work."""
        self._check(self.lib.mpc_update_instance_params_batch(self._h, int(insts.size), insts.ctypes.data_as(_IP), ks.ctypes.data_as(_IP),
                                                              offsets.ctypes.data_as(_IP), lens.ctypes.data_as(_IP), _dp(vals)), "mpc_update_instance_params_batch")

    def set_failure_policy(self, isolate):
        """isolate: a failed instance is reported (``stats.converged = -code``) and skipped until revived instead of failing the run."""
        self._check(self.lib.mpc_set_failure_policy(self._h, int(bool(isolate))), "mpc_set_failure_policy")

    def revive_instance(self, dst, src=0):
        self._check(self.lib.mpc_revive_instance(self._h, int(dst), int(src)), "mpc_revive_instance")

    def get_gain(self, k=0):
        """-> (K_k[B][nu][ndx], kff_k[B][nu]) of one knot (mpc_get_gain)."""
        d = self.dims
        K, kff = np.zeros((d.batch, d.nu, d.ndx)), np.zeros((d.batch, d.nu))
        self._check(self.lib.mpc_get_gain(self._h, int(k), _dp(K), _dp(kff)), "mpc_get_gain")
        return K, kff

    def get_state(self):
        """Checkpoint of the handle (stage tables of the horizon, iterate, multipliers, measured state, penalties): a float64 array,
        portable between libraries of the same dimensions (mpc_get_state)."""
        n = int(self.lib.mpc_state_size(self._h))
        buf = np.zeros(n)
        if self.lib.mpc_get_state(self._h, _dp(buf), n) != n:
            self._check(-1, "mpc_get_state")
        return buf

    def set_state(self, state):
        state = _f64(state)
        self._check(self.lib.mpc_set_state(self._h, _dp(state), state.size), "mpc_set_state")

    def get_results(self, gains=True, multipliers=False):
        d = self.dims
        B, N = d.batch, d.horizon
        out = {"xs": np.zeros((B, N + 1, d.nx)), "us": np.zeros((B, N, d.nu))}
        if gains:
            out["K"] = np.zeros((B, N, d.nu, d.ndx))
            out["kff"] = np.zeros((B, N, d.nu))
        if multipliers:
            out["vs"] = np.zeros((B, N + 1, max(int(d.nc_max), 1)))  # the library keeps (and copies) at least one row per knot
            out["lams"] = np.zeros((B, N + 1, d.ndx))
        self._check(self.lib.mpc_get_results(self._h, _dp(out["xs"]), _dp(out["us"]), _dp(out.get("K")), _dp(out.get("kff")),
                                             _dp(out.get("vs")), _dp(out.get("lams"))), "mpc_get_results")
        return out

    def get_stage_data(self, k):
        d = self.dims
        xdot = np.zeros((d.batch, d.ndx))
        wr = np.zeros((d.batch, 2, 6))
        self._check(self.lib.mpc_get_stage_data(self._h, k, _dp(xdot), _dp(wr)), "mpc_get_stage_data")
        return xdot, wr

    # -- per-kernel timing ----------------------------------------------------------------------
    def profile(self, mode):
        self._check(self.lib.mpc_profile(self._h, int(mode)), "mpc_profile")

    def profile_read(self, slots=False):
        """-> {kernel name: (launches, total_ms)}  (``slots=True``: (launches, total_ms, slot index) — the index is the bit
        of ``profile(16 * mask)``)"""
        out = {}
        name = C.create_string_buffer(64)
        cnt, ms = C.c_int32(0), C.c_double(0.0)
        nslots = self._check(self.lib.mpc_profile_read(self._h, 0, name, 64, C.byref(cnt), C.byref(ms)), "mpc_profile_read")
        for i in range(nslots):
            self._check(self.lib.mpc_profile_read(self._h, i, name, 64, C.byref(cnt), C.byref(ms)), "mpc_profile_read")
            if cnt.value:
                out[name.value.decode()] = (cnt.value, ms.value, i) if slots else (cnt.value, ms.value)
        return out

    def kernel_info(self):
        """-> [(kernel, {threads, vgprs, scratch_bytes, static_lds, dynamic_lds, workgroups_per_cu, workgroups_per_launch,
        waves_per_simd})] for the kernels one pass of this handle launches (mpc_abi.h, mpc_kernel_info)."""
        keys = ("threads", "vgprs", "scratch_bytes", "static_lds", "dynamic_lds", "workgroups_per_cu", "workgroups_per_launch", "waves_per_simd")
        name = C.create_string_buffer(96)
        info = (C.c_int32 * 8)()
        n = self._check(self.lib.mpc_kernel_info(self._h, -1, name, 96, info), "mpc_kernel_info")
        out = []
        for i in range(n):
            self._check(self.lib.mpc_kernel_info(self._h, i, name, 96, info), "mpc_kernel_info")
            out.append((name.value.decode(), dict(zip(keys, [int(v) for v in info]))))
        return out

    # -- parity hooks ---------------------------------------------------------------------------
    def debug_evaluate(self, xs, us):
        d = self.dims
        xs = self._bcast(xs, (d.batch, d.horizon + 1, d.nx))
        us = self._bcast(us, (d.batch, d.horizon, d.nu))
        self._check(self.lib.mpc_debug_evaluate(self._h, _dp(xs), _dp(us)), "mpc_debug_evaluate")

    def debug_get(self, name, k, b=0):
        cap = 4 * (self.dims.ndx + self.dims.nu + self.dims.nc_max + 8) ** 2
        buf = np.zeros(cap)
        n = self._check(self.lib.mpc_debug_get(self._h, name.encode(), b, k, _dp(buf), cap), "mpc_debug_get(%s)" % name)
        return buf[:n].copy()
