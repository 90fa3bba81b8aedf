"""Stand-in for the three ``ndcurves`` classes talos_utils.py:290-295 uses to build a swing-foot curve: a Bezier curve for the
translation and a geodesic for the rotation.  The evaluation is the closed form of mpc_benchmark_amd/references.py (same functions), so
the reference's ``footTrajectory`` and this repo's ``FootTrajectory`` can be compared bit for bit.  Build-container tooling."""
import numpy as np

from mpc_benchmark_amd import references as _r


class bezier3:
    def __init__(self, waypoints, t_min=0.0, t_max=1.0):
        self.wps = np.array(waypoints, dtype=float)
        self.t_min, self.t_max = float(t_min), float(t_max)

    def __call__(self, t):
        s = (float(t) - self.t_min) / (self.t_max - self.t_min)
        return _r.bezier_eval(self.wps, s)


class SE3Curve:
    def __init__(self, translation_curve, init_rotation, end_rotation):
        self._tr = translation_curve
        self.R0, self.R1 = np.array(init_rotation, dtype=float), np.array(end_rotation, dtype=float)

    def translation(self, t):
        return self._tr(t)

    def rotation(self, t):
        s = (float(t) - self._tr.t_min) / (self._tr.t_max - self._tr.t_min)
        return _r.slerp_rotation(self.R0, self.R1, s)


class piecewise_SE3:
    def __init__(self, curve):
        self._c = curve

    def translation(self, t):
        return self._c.translation(t)

    def rotation(self, t):
        return self._c.rotation(t)
