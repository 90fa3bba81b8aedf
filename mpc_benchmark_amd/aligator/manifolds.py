"""``aligator.manifolds`` mirror: the two state spaces the reference uses
(fulldynamic_talos.py:62, centroidal_talos.py:46-47)."""
from __future__ import annotations

import numpy as np

from ..robot import minipin as _mp


class VectorSpace:
    def __init__(self, n):
        self.nx = int(n)
        self.ndx = int(n)

    def neutral(self):
        return np.zeros(self.nx)

    def difference(self, x0, x1):
        return np.asarray(x1, dtype=float) - np.asarray(x0, dtype=float)

    def integrate(self, x, dx):
        return np.asarray(x, dtype=float) + np.asarray(dx, dtype=float)

    def copy(self):
        return VectorSpace(self.nx)


class MultibodyPhaseSpace:
    """x = (q, v); ndx = 2 nv.  ``difference(x0, x1) = (pin.difference(q0, q1), v1 - v0)``."""

    def __init__(self, model):
        self.model = model
        self.nx = int(model.nq + model.nv)
        self.ndx = int(2 * model.nv)

    def neutral(self):
        return np.concatenate((_mp.neutral(self.model), np.zeros(self.model.nv)))

    def difference(self, x0, x1):
        nq = self.model.nq
        x0 = np.asarray(x0, dtype=float)
        x1 = np.asarray(x1, dtype=float)
        return np.concatenate((_mp.difference(self.model, x0[:nq], x1[:nq]), x1[nq:] - x0[nq:]))

    def integrate(self, x, dx):
        nq, nv = self.model.nq, self.model.nv
        x = np.asarray(x, dtype=float)
        dx = np.asarray(dx, dtype=float)
        return np.concatenate((_mp.integrate(self.model, x[:nq], dx[:nv]), x[nq:] + dx[nv:]))

    def copy(self):
        return MultibodyPhaseSpace(self.model)
