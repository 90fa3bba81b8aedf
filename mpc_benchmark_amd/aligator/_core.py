"""Python mirror of the part of the ``aligator`` module that the reference scripts use
(SURVEY.md §8b-1): residuals, costs, dynamics, constraint sets, StageModel, TrajOptProblem.

These classes hold parameters only.  No arithmetic of the hot path happens here: a problem is
*lowered* (``_lower``) to the flat int32/float64 stage tables of ``include/mpc_abi.h`` and solved by the
native library.  Semantics reproduced from the reference's usage:

* composition copies (``TrajOptProblem(stages)``, ``StageModel(cost, dyn)``, ``addCost``, ``addConstraint``,
  ``QuadraticResidualCost(space, residual, W)``, ``replaceStageCircular``, ``addTerminalConstraint``);
* accessors return live references (``problem.stages[j]``, ``.cost``, ``getComponent``, ``.residual``,
  ``.dynamics.differential_dynamics``, ``.contact_map.contact_poses``);
* a mutation after composition marks the owning stage dirty so that the next ``solver.run`` refreshes its
  parameter table on the device.
"""
from __future__ import annotations

import numpy as np

from .. import _capi as K
from . import manifolds as _manifolds


def _vec(a, n=None):
    a = np.array(a, dtype=float).reshape(-1)
    if n is not None and a.size != n:
        raise ValueError("expected a vector of size %d, got %d" % (n, a.size))
    return a


def _se3_flat(M):
    out = np.empty(12)
    out[:9] = np.asarray(M.rotation, dtype=float).reshape(-1)
    out[9:] = np.asarray(M.translation, dtype=float).reshape(-1)
    return out


class _Owned:
    """Base of every object that can sit inside a stage: keeps a back-pointer to its owner so that
    ``setReference`` & co can flag the stage dirty."""
    _owner = None

    def _touch(self):
        o = self
        while o is not None:
            if isinstance(o, (StageModel, _TerminalNode)):
                o._dirty = True
            o = getattr(o, "_owner", None)

    def _adopt(self, child):
        if isinstance(child, _Owned):
            child._owner = self
        return child

    def _ref_flat(self):
        """The reference as it sits in the stage's parameter table (what ``_lower`` returns as its last element)."""
        return np.asarray(self._ref, dtype=float).reshape(-1)

    def _owner_node(self):
        o = self
        while o is not None:
            if isinstance(o, (StageModel, _TerminalNode)):
                return o
            o = getattr(o, "_owner", None)
        return None

    def _reference_changed(self, flat):
        """A reference / target value changed.  Fast path (what the MPC loops do 2 N times per tick, fulldynamic_talos.py:
        461-463): the stage is lowered and structurally unchanged, so the new values are written straight into its
        parameter table at the slot recorded by ``lower_stage`` and only those doubles travel to the device.
        Anything else falls back to re-lowering the whole stage."""
        slot = getattr(self, "_slot", None)
        if slot is not None:
            node, off, size = slot
            flat = np.asarray(flat, dtype=float).reshape(-1)
            if (node._lowered is not None and not node._dirty and flat.size == size and self._owner_node() is node
                    and off + size <= node._lowered[1].size):
                tab = node._lowered[1]
                if (tab[off:off + size] != flat).any():  # an unchanged reference (most of the 2 N per tick) costs a compare
                    tab[off:off + size] = flat
                    node._patches.append((off, size))
                return
        self._touch()


# ------------------------------------------------------------------------------------------------
# constraint sets (aligator.constraints)
# ------------------------------------------------------------------------------------------------
class EqualityConstraintSet:
    _role = K.ROLE_EQUALITY

    def copy(self):
        return EqualityConstraintSet()


class NegativeOrthant:
    _role = K.ROLE_NEG_ORTHANT

    def copy(self):
        return NegativeOrthant()


class BoxConstraint:
    _role = K.ROLE_BOX

    def __init__(self, lower, upper):
        self.lower_limit = _vec(lower)
        self.upper_limit = _vec(upper)

    def copy(self):
        return BoxConstraint(self.lower_limit, self.upper_limit)


# ------------------------------------------------------------------------------------------------
# residual functions
# ------------------------------------------------------------------------------------------------
class StageFunction(_Owned):
    _type = 0
    nr = 0

    def copy(self):
        import copy as _c
        new = _c.copy(self)
        new._owner = None
        for k, v in list(self.__dict__.items()):
            if isinstance(v, np.ndarray):
                new.__dict__[k] = v.copy()
            elif isinstance(v, ContactMap):
                new.__dict__[k] = new._adopt(v.copy())
            elif hasattr(v, "rotation") and hasattr(v, "translation"):
                new.__dict__[k] = v.copy()
        return new

    def __getitem__(self, idx):
        return FunctionSlice(self, idx)

    # (type, dim, iarg0, iarg1, params)
    def _lower(self, ctx):
        raise NotImplementedError(type(self).__name__)


class FunctionSlice(StageFunction):
    """``residual[a:b]`` / ``residual[i]`` (fulldynamic_talos.py:169-172, 208)."""

    def __init__(self, func, idx):
        self.func = self._adopt(func.copy())
        if isinstance(idx, slice):
            self.indices = list(range(*idx.indices(func.nr)))
        elif isinstance(idx, (int, np.integer)):
            self.indices = [int(idx)]
        else:
            self.indices = [int(i) for i in idx]
        if self.indices != list(range(self.indices[0], self.indices[0] + len(self.indices))):
            raise NotImplementedError("only contiguous function slices are supported")
        self.nr = len(self.indices)
        self.ndx, self.nu = func.ndx, func.nu

    def copy(self):
        new = FunctionSlice.__new__(FunctionSlice)
        new.func = new._adopt(self.func.copy())
        new.indices, new.nr, new.ndx, new.nu = list(self.indices), self.nr, self.ndx, self.nu
        return new

    def setReference(self, ref):
        self.func.setReference(ref)

    def getReference(self):
        return self.func.getReference()

    def _lower(self, ctx):
        t, dim, i0, i1, p = self.func._lower(ctx)
        start = self.indices[0]
        if t in (K.TERM_STATE_ERROR, K.TERM_CONTROL_ERROR):
            return t, self.nr, i0 + start, i1, p
        if t in (K.TERM_FRAME_TRANSLATION, K.TERM_COM_TRANSLATION):
            return t, self.nr, i0, i1 + start, p
        raise NotImplementedError("slicing of %s is not supported" % type(self.func).__name__)


class StateErrorResidual(StageFunction):
    _type = K.TERM_STATE_ERROR

    def __init__(self, space, nu, target):
        self.space = space
        self.ndx, self.nu, self.nr = space.ndx, int(nu), space.ndx
        self.target = _vec(target, space.nx)

    def _lower(self, ctx):
        return self._type, self.nr, 0, 0, self.target


class ControlErrorResidual(StageFunction):
    _type = K.TERM_CONTROL_ERROR

    def __init__(self, ndx, target):
        if isinstance(target, (int, np.integer)):
            target = np.zeros(int(target))
        self.target = _vec(target)
        self.ndx, self.nu, self.nr = int(ndx), self.target.size, self.target.size

    def _lower(self, ctx):
        return self._type, self.nr, 0, 0, self.target


class _FrameFunction(StageFunction):
    def _frame(self, ctx):
        return ctx.frame_index(self.pin_model, self.frame_id)


class FramePlacementResidual(_FrameFunction):
    _type = K.TERM_FRAME_PLACEMENT

    def __init__(self, ndx, nu, model, ref, frame_id):
        self.ndx, self.nu, self.nr = int(ndx), int(nu), 6
        self.pin_model, self.frame_id = model, int(frame_id)
        self._ref = ref.copy()

    def _ref_flat(self):
        return _se3_flat(self._ref)

    def setReference(self, ref):
        self._ref = ref.copy()
        # the 2 N calls per tick of the MPC loops (fulldynamic_talos.py:461-463): _reference_changed's fast path without building the
        # flat 12-vector — rotation and translation are compared with, and written into, the parameter table where they lie
        slot = getattr(self, "_slot", None)
        if slot is not None:
            node, off, size = slot
            low = node._lowered
            if low is not None and not node._dirty and size == 12 and off + 12 <= low[1].size and self._owner_node() is node:
                seg = low[1][off:off + 12]
                Rf, tr = np.asarray(self._ref.rotation, dtype=float).reshape(-1), np.asarray(self._ref.translation, dtype=float).reshape(-1)
                if (seg[:9] != Rf).any() or (seg[9:] != tr).any():
                    seg[:9] = Rf
                    seg[9:] = tr
                    node._patches.append((off, 12))
                return
        self._touch()

    def getReference(self):
        return self._ref

    def _lower(self, ctx):
        return self._type, 6, self._frame(ctx), 0, _se3_flat(self._ref)


class FrameTranslationResidual(_FrameFunction):
    _type = K.TERM_FRAME_TRANSLATION

    def __init__(self, ndx, nu, model, ref, frame_id):
        self.ndx, self.nu, self.nr = int(ndx), int(nu), 3
        self.pin_model, self.frame_id = model, int(frame_id)
        self._ref = _vec(ref, 3)

    def setReference(self, ref):
        self._ref = _vec(ref, 3)
        self._reference_changed(self._ref_flat())

    def getReference(self):
        return self._ref

    def _lower(self, ctx):
        return self._type, 3, self._frame(ctx), 0, self._ref


class FrameVelocityResidual(_FrameFunction):
    _type = K.TERM_FRAME_VELOCITY

    def __init__(self, ndx, nu, model, ref, frame_id, ref_frame=0):
        self.ndx, self.nu, self.nr = int(ndx), int(nu), 6
        self.pin_model, self.frame_id = model, int(frame_id)
        self._ref = _vec(getattr(ref, "np", ref), 6)
        if int(ref_frame) != 0:
            raise NotImplementedError("FrameVelocityResidual: only pin.LOCAL is supported")

    def setReference(self, ref):
        self._ref = _vec(getattr(ref, "np", ref), 6)
        self._reference_changed(self._ref_flat())

    def getReference(self):
        return self._ref

    def _lower(self, ctx):
        return self._type, 6, self._frame(ctx), 0, self._ref


class CenterOfMassTranslationResidual(StageFunction):
    _type = K.TERM_COM_TRANSLATION

    def __init__(self, ndx, nu, model, ref):
        self.ndx, self.nu, self.nr = int(ndx), int(nu), 3
        self.pin_model = model
        self._ref = _vec(ref, 3)

    def setReference(self, ref):
        self._ref = _vec(ref, 3)
        self._reference_changed(self._ref_flat())

    def getReference(self):
        return self._ref

    def _lower(self, ctx):
        return self._type, 3, 0, 0, self._ref


class CentroidalMomentumResidual(StageFunction):
    _type = K.TERM_CENTROIDAL_MOMENTUM

    def __init__(self, ndx, nu, model, ref):
        self.ndx, self.nu, self.nr = int(ndx), int(nu), 6
        self.pin_model = model
        self._ref = _vec(ref, 6)

    def setReference(self, ref):
        self._ref = _vec(ref, 6)
        self._reference_changed(self._ref_flat())

    def getReference(self):
        return self._ref

    def _lower(self, ctx):
        return self._type, 6, 0, 0, self._ref


def _check_actuation(B, nv):
    B = np.asarray(B, dtype=float)
    nu = B.shape[1]
    if B.shape[0] != nv or not np.array_equal(B, np.eye(nv, nu, -(nv - nu))):
        raise NotImplementedError("actuation matrix must be np.eye(nv, nu, -(nv - nu)) (fulldynamic_talos.py:76)")
    return nu


class ContactForceResidual(StageFunction):
    _type = K.TERM_CONTACT_FORCE

    def __init__(self, ndx, model, actuation, constraint_models, prox_settings, fref, contact_name):
        self.ndx, self.nr = int(ndx), 6
        self.pin_model = model
        self.nu = _check_actuation(actuation, model.nv)
        self.constraint_models = list(constraint_models)
        self.prox_settings = prox_settings
        self._ref = _vec(fref, 6)
        self.contact_name = contact_name

    def setReference(self, ref):
        self._ref = _vec(ref, 6)
        self._reference_changed(self._ref_flat())

    def getReference(self):
        return self._ref

    def _lower(self, ctx):
        names = [cm.name for cm in self.constraint_models]
        return self._type, 6, names.index(self.contact_name), 0, self._ref


def wrench_cone_matrix(mu, half_length, half_width):
    """17 x 6 matrix A of the surface-contact wrench cone, rows A w <= 0 with w = [f; tau] in the contact
    frame: unilateral fz (1), friction pyramid (4), CoP inside the sole rectangle (4), yaw-torque bounds (8)
    [Caron, Pham, Nakamura, ICRA 2015].  Upstream row order/signs are not recoverable:
    this ordering is the build's definition, shared by the oracle and the HIP path through the stage table."""
    L, W = float(half_length), float(half_width)
    A = np.zeros((17, 6))
    A[0] = [0, 0, -1, 0, 0, 0]
    A[1] = [1, 0, -mu, 0, 0, 0]
    A[2] = [-1, 0, -mu, 0, 0, 0]
    A[3] = [0, 1, -mu, 0, 0, 0]
    A[4] = [0, -1, -mu, 0, 0, 0]
    A[5] = [0, 0, -W, 1, 0, 0]
    A[6] = [0, 0, -W, -1, 0, 0]
    A[7] = [0, 0, -L, 0, 1, 0]
    A[8] = [0, 0, -L, 0, -1, 0]
    r = 9
    for s1 in (1.0, -1.0):
        for s2 in (1.0, -1.0):
            # -tau_z - mu (L+W) fz + s1 (W fx - mu tau_x) + s2 (L fy - mu tau_y) <= 0
            A[r] = [s1 * W, s2 * L, -mu * (L + W), -s1 * mu, -s2 * mu, -1.0]
            # +tau_z - mu (L+W) fz + s1 (W fx + mu tau_x) + s2 (L fy + mu tau_y) <= 0
            A[r + 4] = [s1 * W, s2 * L, -mu * (L + W), s1 * mu, s2 * mu, 1.0]
            r += 1
    return A


class MultibodyWrenchConeResidual(StageFunction):
    _type = K.TERM_MB_WRENCH_CONE

    def __init__(self, ndx, model, actuation, constraint_models, prox_settings, contact_name, mu, half_length, half_width):
        self.ndx, self.nr = int(ndx), 17
        self.pin_model = model
        self.nu = _check_actuation(actuation, model.nv)
        self.constraint_models = list(constraint_models)
        self.prox_settings = prox_settings
        self.contact_name = contact_name
        self.mu, self.half_length, self.half_width = float(mu), float(half_length), float(half_width)

    def _lower(self, ctx):
        names = [cm.name for cm in self.constraint_models]
        A = wrench_cone_matrix(self.mu, self.half_length, self.half_width)
        return self._type, 17, names.index(self.contact_name), 0, A.reshape(-1)


class CentroidalWrenchConeResidual(StageFunction):
    _type = K.TERM_CENTROIDAL_WRENCH_CONE

    def __init__(self, ndx, nu, k, mu, half_length, half_width):
        self.ndx, self.nu, self.nr = int(ndx), int(nu), 17
        self.k = int(k)
        self.mu, self.half_length, self.half_width = float(mu), float(half_length), float(half_width)

    def _lower(self, ctx):
        A = wrench_cone_matrix(self.mu, self.half_length, self.half_width)
        return self._type, 17, self.k, 0, A.reshape(-1)


class _PoseList(list):
    def __init__(self, owner, items):
        super().__init__(np.array(p, dtype=float).reshape(3) for p in items)
        self._owner_map = owner

    def __setitem__(self, i, v):
        super().__setitem__(i, np.array(v, dtype=float).reshape(3))
        self._owner_map._touch()


class ContactMap(_Owned):
    """``aligator.ContactMap(names, states, poses)`` (centroidal_talos.py:210); ``contact_poses[i] = p``
    is item assignment on a live list (centroidal_talos.py:377-384)."""

    def __init__(self, contact_names, contact_states, contact_poses):
        self.contact_names = list(contact_names)
        self.contact_states = [bool(s) for s in contact_states]
        self.contact_poses = _PoseList(self, contact_poses)

    @property
    def size(self):
        return len(self.contact_states)

    def copy(self):
        return ContactMap(self.contact_names, self.contact_states, [p.copy() for p in self.contact_poses])

    def _flat(self):
        out = []
        for s, p in zip(self.contact_states, self.contact_poses):
            out.append([1.0 if s else 0.0, p[0], p[1], p[2]])
        return np.array(out).reshape(-1)


class _CentroidalAccBase(StageFunction):
    def __init__(self, nx, nu, mass, gravity, contact_map, force_size):
        if int(force_size) != 6:
            raise NotImplementedError("only 6D contact wrenches are supported")
        self.ndx, self.nu, self.nr = int(nx), int(nu), 3
        self.mass = float(mass)
        self.gravity = _vec(gravity, 3)
        self.contact_map = self._adopt(contact_map.copy())

    def _lower(self, ctx):
        p = np.concatenate(([self.mass], self.gravity, self.contact_map._flat()))
        return self._type, 3, self.contact_map.size, 0, p


class CentroidalAccelerationResidual(_CentroidalAccBase):
    _type = K.TERM_CENTROIDAL_LIN_ACC


class AngularAccelerationResidual(_CentroidalAccBase):
    _type = K.TERM_CENTROIDAL_ANG_ACC


class _CentroidalSlice(StageFunction):
    """x[a:a+3] - ref on the 9-dim centroidal state — lowered to a sliced state error."""
    _start = 0

    def __init__(self, nx, nu, ref):
        self.ndx, self.nu, self.nr = int(nx), int(nu), 3
        self._ref = _vec(ref, 3)

    def setReference(self, ref):
        self._ref = _vec(ref, 3)
        self._reference_changed(self._ref_flat())

    def getReference(self):
        return self._ref

    def _ref_flat(self):
        full = np.zeros(self.ndx)
        full[self._start:self._start + 3] = self._ref
        return full

    def _lower(self, ctx):
        return K.TERM_STATE_ERROR, 3, self._start, 0, self._ref_flat()


class CentroidalCoMResidual(_CentroidalSlice):
    _start = 0


class LinearMomentumResidual(_CentroidalSlice):
    _start = 3


class AngularMomentumResidual(_CentroidalSlice):
    _start = 6


class CentroidalMomentumDerivativeResidual(StageFunction):
    _type = K.TERM_CENTROIDAL_MOMENTUM_DER

    def __init__(self, ndx, model, gravity, contact_states, contact_ids, force_size):
        if int(force_size) != 6:
            raise NotImplementedError("only 6D contact wrenches are supported")
        self.ndx, self.nr = int(ndx), 6
        self.pin_model = model
        self.nu = model.nv - 6 + 6 * len(contact_ids)
        self.gravity = _vec(gravity, 3)
        self.contact_states = [bool(s) for s in contact_states]
        self.contact_ids = [int(i) for i in contact_ids]

    def _lower(self, ctx):
        frames = [ctx.frame_index(self.pin_model, f) for f in self.contact_ids]
        p = np.concatenate((self.gravity, [1.0 if s else 0.0 for s in self.contact_states], [float(f) for f in frames]))
        return self._type, 6, len(frames), 0, p


# ------------------------------------------------------------------------------------------------
# costs
# ------------------------------------------------------------------------------------------------
class QuadraticResidualCost(_Owned):
    def __init__(self, space, residual, weights):
        self.space = space
        self.residual = self._adopt(residual.copy())
        self.weights = np.array(weights, dtype=float)
        self.nu = residual.nu

    def copy(self):
        c = QuadraticResidualCost(self.space, self.residual, self.weights)
        c.__class__ = type(self)  # (a copied QuadraticStateCost / QuadraticControlCost keeps its setTarget)
        return c


class QuadraticStateCost(QuadraticResidualCost):
    def __init__(self, space, nu, target, weights):
        super().__init__(space, StateErrorResidual(space, nu, target), weights)

    def setTarget(self, target):
        self.residual.target = _vec(target, self.space.nx)
        self._touch()


class QuadraticControlCost(QuadraticResidualCost):
    def __init__(self, space, target, weights):
        if isinstance(target, (int, np.integer)):
            target = np.zeros(int(target))
        super().__init__(space, ControlErrorResidual(space.ndx, target), weights)

    def setTarget(self, target):
        self.residual.target = _vec(target)
        self._touch()


class CostStack(_Owned):
    """``aligator.CostStack(space, nu)``; components are keyed by insertion index or by name
    (fulldynamic_talos.py:462 uses ints, kinodynamic_talos.py:384 uses strings) and stored as
    ``(cost, weight)`` pairs (fulldynamic_talos.py:509)."""

    def __init__(self, space, nu):
        self.space = space
        self.nu = int(nu)
        self.components = {}

    def addCost(self, *args):
        if isinstance(args[0], str):
            key, cost = args[0], args[1]
            weight = float(args[2]) if len(args) > 2 else 1.0
        else:
            key, cost = len(self.components), args[0]
            weight = float(args[1]) if len(args) > 1 else 1.0
        self.components[key] = (self._adopt(cost.copy()), weight)
        self._touch()
        return self.components[key]

    def getComponent(self, key):
        return self.components[key][0]

    def size(self):
        return len(self.components)

    def copy(self):
        new = CostStack(self.space, self.nu)
        for key, (cost, w) in self.components.items():
            new.components[key] = (new._adopt(cost.copy()), w)
        return new


# ------------------------------------------------------------------------------------------------
# dynamics (aligator.dynamics)
# ------------------------------------------------------------------------------------------------
class MultibodyConstraintFwdDynamics(_Owned):
    def __init__(self, space, actuation, constraint_models, prox_settings):
        self.space = space
        self.nu = _check_actuation(actuation, space.model.nv)
        self.actuation_matrix = np.array(actuation, dtype=float)
        self.constraint_models = list(constraint_models)
        self.prox_settings = prox_settings

    def copy(self):
        return MultibodyConstraintFwdDynamics(self.space, self.actuation_matrix, self.constraint_models, self.prox_settings)


class KinodynamicsFwdDynamics(_Owned):
    def __init__(self, space, model, gravity, contact_states, contact_ids, force_size):
        if int(force_size) != 6:
            raise NotImplementedError("only 6D contact wrenches are supported")
        self.space, self.pin_model = space, model
        self.gravity = _vec(gravity, 3)
        self.contact_states = [bool(s) for s in contact_states]
        self.contact_ids = [int(i) for i in contact_ids]
        self.nu = model.nv - 6 + 6 * len(self.contact_ids)

    def copy(self):
        return KinodynamicsFwdDynamics(self.space, self.pin_model, self.gravity, self.contact_states, self.contact_ids, 6)


class CentroidalFwdDynamics(_Owned):
    def __init__(self, space, mass, gravity, contact_map, force_size):
        if int(force_size) != 6:
            raise NotImplementedError("only 6D contact wrenches are supported")
        self.space = space
        self.mass = float(mass)
        self.gravity = _vec(gravity, 3)
        self.contact_map = self._adopt(contact_map.copy())
        self.nu = 6 * self.contact_map.size

    def copy(self):
        return CentroidalFwdDynamics(self.space, self.mass, self.gravity, self.contact_map, 6)


class _Integrator(_Owned):
    def __init__(self, ode, timestep):
        self.differential_dynamics = self._adopt(ode.copy())
        self.timestep = float(timestep)
        self.space = ode.space
        self.nu = ode.nu

    def copy(self):
        return type(self)(self.differential_dynamics, self.timestep)


class IntegratorSemiImplEuler(_Integrator):
    pass


class IntegratorEuler(_Integrator):
    pass


# ------------------------------------------------------------------------------------------------
# stage / problem containers
# ------------------------------------------------------------------------------------------------
class StageConstraint:
    def __init__(self, func, cstr_set):
        self.func = func
        self.set = cstr_set


class _ConstraintStack(_Owned):
    def __init__(self):
        self.funcs = []
        self.sets = []

    def pushBack(self, func, cset):
        self.funcs.append(self._adopt(func.copy()))
        self.sets.append(cset.copy())
        self._touch()

    def clear(self):
        self.funcs, self.sets = [], []
        self._touch()

    def __len__(self):
        return len(self.funcs)

    @property
    def total_dim(self):
        return sum(f.nr for f in self.funcs)


class StageData:
    """Placeholder returned by ``StageModel.createData()``; the native solver owns the real workspace."""


class StageModel(_Owned):
    def __init__(self, cost, dynamics):
        self._dirty = True
        self._lowered = None
        self.cost = self._adopt(cost.copy())
        self.dynamics = self._adopt(dynamics.copy())
        self.constraints = self._adopt(_ConstraintStack())
        self.xspace = dynamics.space
        self.nu = dynamics.nu

    def addConstraint(self, *args):
        if len(args) == 1:
            func, cset = args[0].func, args[0].set
        else:
            func, cset = args
        self.constraints.pushBack(func, cset)

    def createData(self):
        return StageData()

    @property
    def ndx1(self):
        return self.xspace.ndx

    def copy(self):
        new = StageModel(self.cost, self.dynamics)
        for f, s in zip(self.constraints.funcs, self.constraints.sets):
            new.constraints.pushBack(f, s)
        return new


class _TerminalNode(_Owned):
    """Owner of the terminal cost and terminal constraints (so that their mutations mark it dirty)."""

    def __init__(self):
        self._dirty = True
        self._lowered = None


class TrajOptProblem:
    def __init__(self, x0, stages, term_cost):
        self._x0 = _vec(x0)
        self.stages = [s.copy() for s in stages]  # value semantics: `[stage] * N` becomes N independent stages
        self._term = _TerminalNode()
        self.term_cost = self._term._adopt(term_cost.copy())
        self.term_constraints = self._term._adopt(_ConstraintStack())
        self._cycled = []  # stages appended by replaceStageCircular since the last solver sync
        self._version = 0

    @property
    def x0_init(self):
        return self._x0

    @x0_init.setter
    def x0_init(self, x):
        self._x0 = _vec(x, self._x0.size)

    @property
    def num_steps(self):
        return len(self.stages)

    def addTerminalConstraint(self, cstr):
        self.term_constraints.pushBack(cstr.func, cstr.set)

    def removeTerminalConstraint(self):
        self.term_constraints.clear()

    def replaceStageCircular(self, stage):
        new = stage.copy()
        self.stages.pop(0)
        self.stages.append(new)
        self._cycled.append(new)


# ------------------------------------------------------------------------------------------------
# lowering
# ------------------------------------------------------------------------------------------------
class LoweringContext:
    """Collects the model-level tables (frames, contact models) referenced by the stages."""

    def __init__(self):
        self.model = None
        self.frames = []        # list of (parent joint id (pin numbering), SE3 placement) keyed by pin frame id
        self._frame_ids = {}
        self.contacts = []      # RigidConstraintModel-likes, keyed by name
        self._contact_names = {}
        self.prox_mu = 0.0
        self.changed = False

    def set_model(self, model):
        if self.model is None:
            self.model = model
            self.changed = True

    def frame_index(self, model, fid):
        self.set_model(model)
        if fid not in self._frame_ids:
            self._frame_ids[fid] = len(self.frames)
            fr = model.frames[fid]
            parent = getattr(fr, "parentJoint", getattr(fr, "parent", None))
            self.frames.append((int(parent), fr.placement))
            self.changed = True
        return self._frame_ids[fid]

    def contact_index(self, cm):
        if cm.name not in self._contact_names:
            if getattr(cm, "joint2_id", 0) != 0:
                raise NotImplementedError("contact models must be attached to the world on side 2")
            self._contact_names[cm.name] = len(self.contacts)
            self.contacts.append(cm)
            self.changed = True
        return self._contact_names[cm.name]

    def model_tables(self):
        m = self.model
        kinds = {"JointModelFreeFlyer": K.JOINT_FREEFLYER, "JointModelRX": K.JOINT_RX, "JointModelRY": K.JOINT_RY,
                 "JointModelRZ": K.JOINT_RZ}
        nj = m.njoints - 1
        it = [nj, m.nq, m.nv, len(self.frames), len(self.contacts)]
        g = getattr(m, "gravity", None)
        g = np.asarray(getattr(g, "linear", [0.0, 0.0, -9.81]), dtype=float)
        dt = [g[0], g[1], g[2], self.prox_mu]
        for i in range(1, m.njoints):
            jm = m.joints[i]
            name = jm.shortname()
            if name not in kinds:
                raise NotImplementedError("joint type %s is not supported" % name)
            it += [int(m.parents[i]) - 1, kinds[name], int(jm.idx_q), int(jm.idx_v)]
            pl = m.jointPlacements[i]
            Y = m.inertias[i]
            dt += list(_se3_flat(pl)) + [float(Y.mass)] + list(np.asarray(Y.lever, dtype=float).reshape(3)) \
                + list(np.asarray(Y.inertia, dtype=float).reshape(9))
        for parent, pl in self.frames:
            it.append(parent - 1)
            dt += list(_se3_flat(pl))
        for cm in self.contacts:
            it.append(int(cm.joint1_id) - 1)
            dt += list(_se3_flat(cm.joint1_placement)) + list(_se3_flat(cm.joint2_placement)) \
                + list(_vec(cm.corrector.Kp, 6)) + list(_vec(cm.corrector.Kd, 6))
        return np.array(it, dtype=np.int32), np.array(dt, dtype=np.float64)


def _lower_weight(term_type, dim, W, space):
    W = np.asarray(W, dtype=float)
    if W.ndim == 1:
        W = np.diag(W)
    if W.shape != (dim, dim):
        raise ValueError("weight matrix has shape %s, residual has dimension %d" % (W.shape, dim))
    if term_type in (K.TERM_STATE_ERROR, K.TERM_CONTROL_ERROR) and dim > 6:
        if np.count_nonzero(W - np.diag(np.diag(W))):
            raise NotImplementedError("state/control cost weights must be diagonal")
        return np.diag(W).copy(), K.TERM_FLAG_DIAG_WEIGHT
    if not np.count_nonzero(W - np.diag(np.diag(W))):
        return np.diag(W).copy(), K.TERM_FLAG_DIAG_WEIGHT  # diagonal weights travel as their diagonal
    return W.reshape(-1).copy(), 0


def lower_stage(ctx, cost, dynamics, constraints, slots=None):
    """-> (desc int32[], params float64[]) for one stage (``dynamics`` is None on the terminal node).  ``slots`` (a list)
    receives (residual, offset, size) for every term: where its reference sits in the parameter table."""
    params = []
    off = [0]

    def push(a):
        a = np.asarray(a, dtype=float).reshape(-1)
        o = off[0]
        params.append(a)
        off[0] += a.size
        return o

    head = [K.DYN_NONE, 0, 0, 0, 0, 0, 0, 0]
    if dynamics is not None:
        ode = dynamics.differential_dynamics
        if isinstance(ode, CentroidalFwdDynamics):
            if not isinstance(dynamics, IntegratorEuler):
                raise NotImplementedError("CentroidalFwdDynamics is only supported with IntegratorEuler")
            cmap = ode.contact_map
            if cmap.size != 2:
                raise NotImplementedError("exactly two contacts are supported")
            head[0], head[1] = K.DYN_CENTROIDAL_EULER, cmap.size
            head[2], head[3] = int(cmap.contact_states[0]), int(cmap.contact_states[1])
            head[4] = push(np.concatenate(([ode.mass], ode.gravity, [dynamics.timestep], *[p for p in cmap.contact_poses])))
        elif isinstance(ode, MultibodyConstraintFwdDynamics):
            if not isinstance(dynamics, IntegratorSemiImplEuler):
                raise NotImplementedError("MultibodyConstraintFwdDynamics is only supported with IntegratorSemiImplEuler")
            ctx.set_model(ode.space.model)
            ctx.prox_mu = float(getattr(ode.prox_settings, "mu", 0.0))
            ids = [ctx.contact_index(cm) for cm in ode.constraint_models]
            if len(ids) > 2:
                raise NotImplementedError("at most two contacts are supported")
            head[0], head[1] = K.DYN_MULTIBODY_CONSTRAINT_SEMIEULER, len(ids)
            for i, c in enumerate(ids):
                head[2 + i] = c
            head[4] = push([dynamics.timestep])
        elif isinstance(ode, KinodynamicsFwdDynamics):
            if not isinstance(dynamics, IntegratorSemiImplEuler):
                raise NotImplementedError("KinodynamicsFwdDynamics is only supported with IntegratorSemiImplEuler")
            ctx.set_model(ode.pin_model)
            frames = [ctx.frame_index(ode.pin_model, f) for f in ode.contact_ids]
            head[0], head[1] = K.DYN_KINODYNAMICS_SEMIEULER, len(frames)
            head[2], head[3] = int(ode.contact_states[0]), int(ode.contact_states[1])
            head[4] = push(np.concatenate(([dynamics.timestep], ode.gravity, [float(f) for f in frames])))
        else:
            raise NotImplementedError("dynamics %s" % type(ode).__name__)
    records = []
    nc = 0
    for key, (c, w) in cost.components.items():
        if not isinstance(c, QuadraticResidualCost):
            raise NotImplementedError("cost component %s" % type(c).__name__)
        if hasattr(c.residual, "pin_model"):
            ctx.set_model(c.residual.pin_model)
        t, dim, i0, i1, p = c.residual._lower(ctx)
        Wf, flags = _lower_weight(t, dim, c.weights * w, c.space)
        poff = push(p)
        if slots is not None:
            res = c.residual.func if isinstance(c.residual, FunctionSlice) else c.residual
            slots.append((res, poff, int(np.asarray(p).size)))
        records.append([t, K.ROLE_COST, dim, i0, i1, poff, push(Wf), flags])
    for f, s in zip(constraints.funcs, constraints.sets):
        if hasattr(f, "pin_model"):
            ctx.set_model(f.pin_model)
        inner = f.func if isinstance(f, FunctionSlice) else f
        if isinstance(inner, StateErrorResidual) and isinstance(inner.space, _manifolds.MultibodyPhaseSpace):
            ctx.set_model(inner.space.model)
        t, dim, i0, i1, p = f._lower(ctx)
        poff = push(p)
        if slots is not None:
            slots.append((inner, poff, int(np.asarray(p).size)))
        woff = 0
        if s._role == K.ROLE_BOX:
            if s.lower_limit.size != dim or s.upper_limit.size != dim:
                raise ValueError("box constraint bounds do not match the residual dimension")
            woff = push(np.concatenate((s.lower_limit, s.upper_limit)))
        records.append([t, s._role, dim, i0, i1, poff, woff, 0])
        nc += dim
    head[5], head[6] = len(records), nc
    desc = np.array(head + [w for r in records for w in r], dtype=np.int32)
    return desc, (np.concatenate(params) if params else np.zeros(0))
