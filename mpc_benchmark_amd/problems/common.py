"""Shared pieces of the three Talos OCP builders: robot loading, contact schedules, reference tables.
Constants are the reference's (cited per item); the structure is this repo's own."""
from __future__ import annotations

import numpy as np

from ..robot import minipin as pin
from ..robot import talos_synth

FOOT_FRAMES = ("left_sole_link", "right_sole_link")
FRICTION_MU, FOOT_HALF_LENGTH, FOOT_HALF_WIDTH = 0.8, 0.1, 0.075  # fulldynamic_talos.py:69-71
DT, HORIZON = 0.01, 100                                            # fulldynamic_talos.py:250-251


class Robot:
    """Model + the handful of derived quantities the scripts compute at import time."""

    def __init__(self, complete=False):
        model_c, model_r, q_c, q_r = talos_synth.load_talos()
        self.model = model_c if complete else model_r
        self.q0 = (q_c if complete else q_r).copy()
        m = self.model
        self.nq, self.nv = m.nq, m.nv
        self.data = m.createData()
        pin.framesForwardKinematics(m, self.data, self.q0)
        self.foot_frame_ids = [m.getFrameId(n) for n in FOOT_FRAMES]
        self.foot_joint_ids = [m.frames[f].parentJoint for f in self.foot_frame_ids]
        self.foot_placements = [self.data.oMf[f].copy() for f in self.foot_frame_ids]
        self.com0 = pin.centerOfMass(m, self.data, self.q0)
        self.mass = pin.computeTotalMass(m)
        self.x0 = np.concatenate((self.q0, np.zeros(m.nv)))


def contact_schedule(t_ds, t_ss, total_steps, horizon, final_left_step=False):
    """[left, right] contact flags per MPC tick (fulldynamic_talos.py:255-266, kinodynamic_talos.py:190-198,
    centroidal_talos.py:108-116)."""
    ds, left, right = [True, True], [True, False], [False, True]
    phases = [ds] * t_ds
    for _ in range(total_steps):
        phases += [left] * t_ss + [ds] * t_ds + [right] * t_ss + [ds] * t_ds
    if final_left_step:
        phases += [left] * t_ss + [ds] * t_ds
    phases += [ds] * (2 * horizon)
    return [list(p) for p in phases]


def force_reference_ramp(mass, t_ds, t_ss, total_steps, horizon, nu, gravity_z=-9.81):
    """Vertical-force references u[2] (left) / u[8] (right) over the schedule
    (kinodynamic_talos.py:200-237, centroidal_talos.py:132-169)."""
    f_full, f_half = -mass * gravity_z, -mass * gravity_z / 2.0
    refs = []

    def add(fl, fr):
        u = np.zeros(nu)
        u[2], u[8] = fl, fr
        refs.append(u)

    for i in range(total_steps):
        for j in range(t_ds):
            if i == 0:
                add(f_full * j / t_ds + f_half * (t_ds - j) / t_ds, f_half * (t_ds - j) / t_ds)
            else:
                add(f_full * (j + 1) / t_ds, f_full * (t_ds - j) / t_ds)
        for j in range(t_ss):
            add(f_full, 0.0)
        for j in range(t_ds):
            add(f_full * (t_ds - j) / t_ds, f_full * (j + 1) / t_ds)
        for j in range(t_ss):
            add(0.0, f_full)
    return refs, f_full, f_half
