"""MI355X-native receding-horizon ProxDDP solver behind the ``aligator`` Python surface of the
edantec/MPC_benchmark scripts.  The hot path (per-knot dynamics/cost evaluation, proximal Riccati,
linesearch) runs in hand-written HIP for gfx950 (``csrc/``) reached through the C-ABI of
``include/mpc_abi.h``; this package is the host-side mirror of the reference interface."""
import sys as _sys

__all__ = ["install_as_aligator"]


def install_as_aligator():
    """Make ``import aligator`` (and its submodules) resolve to this package's mirror, so the
    reference scripts run unmodified on a machine that has their other dependencies."""
    from . import aligator as _a
    _sys.modules["aligator"] = _a
    for sub in ("manifolds", "dynamics", "constraints"):
        _sys.modules["aligator." + sub] = getattr(_a, sub)
    return _a
