"""Run logs in the reference's archive format ("next" row N4 of SURVEY.md §8f): one compressed ``.npz`` holding a pickled
dict under the key ``data`` with the fields the scripts record every tick and plot.py reads back (talos_utils.py:113-154,
180-185; plot.py:22-96) — ``xs us com LF_force RF_force LF_torque RF_torque LF_pose RF_pose LF_pose_ref RF_pose_ref
L_measured time`` — plus the centre of pressure of the two foot wrenches (talos_utils.py:156-178).

``TrajectoryLog`` fills those fields from the solver's own outputs (measured state, first control, contact wrenches of
knot 0, foot placements) so that a headless run of the GPU loop can be looked at with the reference's plotting script.
"""
from __future__ import annotations

import os
import time as _time

import numpy as np

FIELDS = ("xs", "us", "com", "LF_force", "RF_force", "LF_torque", "RF_torque", "LF_pose", "RF_pose", "LF_pose_ref", "RF_pose_ref",
          "L_measured", "time")


def save_trajectory(xs, us, com, LF_force, RF_force, LF_torque, RF_torque, time, LF_trans, RF_trans, LF_trans_ref, RF_trans_ref,
                    L_measured, save_name=None, save_dir=None):
    """Argument order and archive layout of talos_utils.save_trajectory; returns the path written."""
    record = dict(zip(FIELDS, (xs, us, com, LF_force, RF_force, LF_torque, RF_torque, LF_trans, RF_trans, LF_trans_ref, RF_trans_ref,
                               L_measured, time)))
    if save_name is None:
        save_name = "sim_data_NO_NAME%d" % int(_time.time())
    if save_dir is None:
        save_dir = os.getcwd()
    os.makedirs(save_dir, exist_ok=True)
    path = os.path.join(save_dir, save_name + ".npz")
    np.savez_compressed(path, data=record)
    return path


def load_data(npz_file):
    """-> the dict stored by ``save_trajectory`` (or by the reference's own scripts)."""
    with np.load(npz_file, allow_pickle=True, encoding="latin1") as d:
        return d["data"][()]


def compute_cop(LF_pose, RF_pose, LF_force, LF_torque, RF_force, RF_torque, min_force=1.0):
    """Centre of pressure of the two foot wrenches (forces / torques in the sole frames, poses with ``.rotation`` /
    ``.translation``): force-weighted mean of the per-foot CoPs ``(-tau_y / f_z, tau_x / f_z, 0)`` moved to the world frame;
    a foot pressing with less than ``min_force`` newtons does not count.  NaN when neither foot is loaded (the reference
    divides by zero there)."""
    total = np.zeros(3)
    fz_sum = 0.0
    for pose, f, tau in ((LF_pose, LF_force, LF_torque), (RF_pose, RF_force, RF_torque)):
        fz = float(f[2])
        if fz > min_force:
            local = np.array([-tau[1] / fz, tau[0] / fz, 0.0])
            total += (np.asarray(pose.rotation) @ local + np.asarray(pose.translation)) * fz
            fz_sum += fz
    if fz_sum <= 0.0:
        return np.full(3, np.nan)
    return total / fz_sum


computeCoP = compute_cop  # the reference's spelling


class TrajectoryLog:
    """Per-tick recorder with the reference's field names; ``save`` writes the archive plot.py loads."""

    def __init__(self):
        self.rows = {k: [] for k in FIELDS}

    def append(self, x, u, com, wrenches, LF_pose, RF_pose, LF_ref, RF_ref, L_measured=None, t=None):
        """``wrenches``: (2, 6) contact wrenches of knot 0, [left, right] x [force(3), torque(3)] in the sole frames
        (``mpc_get_stage_data``); poses as SE3-like objects (``.rotation`` / ``.translation``), 4 x 4 homogeneous matrices or bare
        3-vectors (identity rotation).  The scripts store the SE3 objects themselves (``rdata.oMf[LF_id].copy()``,
        fulldynamic_talos.py:487-490) and plot.py rebuilds ``pin.SE3(LF_pose[i])`` to read ``.translation`` and, in computeCoP,
        ``.rotation`` (plot.py:138-144): a 4 x 4 homogeneous matrix per tick is what ``pin.SE3(...)`` accepts without Pinocchio
        objects in the pickle."""
        def trans(p):
            H = np.eye(4)
            if hasattr(p, "translation"):
                H[:3, :3] = np.asarray(p.rotation, dtype=float); H[:3, 3] = np.asarray(p.translation, dtype=float)
            else:
                a = np.asarray(p, dtype=float)
                if a.shape == (4, 4):
                    H = a.copy()
                else:
                    H[:3, 3] = a.reshape(3)
            return H
        w = np.asarray(wrenches, dtype=float).reshape(2, 6)
        r = self.rows
        r["xs"].append(np.array(x, dtype=float)); r["us"].append(np.array(u, dtype=float)); r["com"].append(np.array(com, dtype=float))
        r["LF_force"].append(w[0, :3].copy()); r["LF_torque"].append(w[0, 3:].copy())
        r["RF_force"].append(w[1, :3].copy()); r["RF_torque"].append(w[1, 3:].copy())
        r["LF_pose"].append(trans(LF_pose)); r["RF_pose"].append(trans(RF_pose))
        r["LF_pose_ref"].append(trans(LF_ref)); r["RF_pose_ref"].append(trans(RF_ref))
        r["L_measured"].append(np.zeros(3) if L_measured is None else np.array(L_measured, dtype=float))
        r["time"].append(float(len(r["time"])) if t is None else float(t))

    def save(self, save_name, save_dir=None):
        r = self.rows
        return save_trajectory(r["xs"], r["us"], r["com"], r["LF_force"], r["RF_force"], r["LF_torque"], r["RF_torque"], r["time"],
                               r["LF_pose"], r["RF_pose"], r["LF_pose_ref"], r["RF_pose_ref"], r["L_measured"], save_name, save_dir)
