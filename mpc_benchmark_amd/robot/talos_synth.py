"""`talos_synth_v1` — deterministic synthetic stand-in for the Talos humanoid.

The reference loads the real robot with ``example_robot_data.load("talos")`` (talos_utils.py:32) — that
package and its URDF are not available here, so this table re-creates the *topology* Pinocchio builds for
Talos (free-flyer + 32 revolute joints, nq=39 / nv=38) with plausible, fixed geometry and box inertias.
No RNG: every number below is a committed constant.  ``load_talos()`` mirrors ``loadTalos()``
(talos_utils.py:31-41) including the locked-joint reduction to nq=29 / nv=28 / nu=22.
"""
from __future__ import annotations

import numpy as np

from . import minipin as pin

_RX, _RY, _RZ = "JointModelRX", "JointModelRY", "JointModelRZ"


def _box(m, dx, dy, dz):
    return np.diag([m / 12.0 * (dy * dy + dz * dz), m / 12.0 * (dx * dx + dz * dz), m / 12.0 * (dx * dx + dy * dy)])


# name, parent name, joint kind, translation from parent joint, mass, com, box dims, effort, (lower, upper)
def _leg(side, sy):
    p = "leg_%s_" % side
    return [
        (p + "1_joint", "root_joint", _RZ, (-0.02, sy * 0.085, -0.27105), 2.05, (0.02, sy * 0.01, 0.02), (0.12, 0.12, 0.10), 100.0, (-0.35, 1.57) if sy > 0 else (-1.57, 0.35)),
        (p + "2_joint", p + "1_joint", _RX, (0.0, 0.0, 0.0), 2.55, (-0.015, sy * 0.01, -0.02), (0.12, 0.14, 0.12), 160.0, (-0.52, 0.52)),
        (p + "3_joint", p + "2_joint", _RY, (0.0, 0.0, 0.0), 6.35, (0.01, sy * 0.05, -0.16), (0.14, 0.14, 0.38), 160.0, (-2.095, 0.7)),
        (p + "4_joint", p + "3_joint", _RY, (0.0, 0.0, -0.38), 3.65, (0.015, sy * 0.02, -0.14), (0.11, 0.11, 0.33), 300.0, (0.0, 2.618)),
        (p + "5_joint", p + "4_joint", _RY, (0.0, 0.0, -0.325), 1.30, (-0.01, sy * 0.02, 0.01), (0.09, 0.12, 0.09), 160.0, (-1.27, 0.68)),
        (p + "6_joint", p + "5_joint", _RX, (0.0, 0.0, 0.0), 1.65, (0.0, 0.0, -0.08), (0.21, 0.14, 0.05), 100.0, (-0.52, 0.52)),
    ]


def _arm(side, sy):
    p = "arm_%s_" % side
    lo_hi_1 = (-1.57, 0.785) if sy > 0 else (-0.785, 1.57)
    lo_hi_2 = (0.0, 2.87) if sy > 0 else (-2.87, 0.0)
    return [
        (p + "1_joint", "torso_2_joint", _RZ, (0.0, sy * 0.1575, 0.232), 2.70, (-0.01, sy * 0.12, 0.03), (0.12, 0.14, 0.12), 44.0, lo_hi_1),
        (p + "2_joint", p + "1_joint", _RX, (0.00493, sy * 0.1365, 0.04673), 2.45, (0.02, sy * 0.0, -0.03), (0.11, 0.11, 0.12), 44.0, lo_hi_2),
        (p + "3_joint", p + "2_joint", _RZ, (0.0, 0.0, 0.0), 2.25, (0.007, 0.0, -0.14), (0.10, 0.10, 0.27), 22.0, (-2.42, 2.42)),
        (p + "4_joint", p + "3_joint", _RY, (0.02, 0.0, -0.273), 1.50, (-0.01, sy * 0.01, -0.06), (0.09, 0.09, 0.16), 22.0, (-2.23, 0.0)),
        (p + "5_joint", p + "4_joint", _RZ, (-0.02, 0.0, -0.2643), 0.95, (0.0, 0.0, 0.07), (0.08, 0.08, 0.12), 17.0, (-2.51, 2.51)),
        (p + "6_joint", p + "5_joint", _RX, (0.0, 0.0, 0.0), 0.55, (0.0, 0.0, 0.0), (0.07, 0.07, 0.07), 17.0, (-1.37, 1.37)),
        (p + "7_joint", p + "6_joint", _RY, (0.0, 0.0, 0.0), 0.45, (0.0, 0.0, -0.05), (0.07, 0.07, 0.08), 17.0, (-0.68, 0.68)),
        ("gripper_%s_joint" % side, p + "7_joint", _RY, (0.0, 0.0, -0.107), 1.00, (0.0, sy * 0.01, -0.06), (0.08, 0.10, 0.14), 10.0, (-0.96, 0.0)),
    ]


_TREE = (
    _leg("left", +1.0)
    + _leg("right", -1.0)
    + [
        ("torso_1_joint", "root_joint", _RZ, (0.0, 0.0, 0.0722), 3.00, (0.0, 0.0, 0.03), (0.16, 0.20, 0.08), 200.0, (-1.25, 1.25)),
        ("torso_2_joint", "torso_1_joint", _RY, (0.0, 0.0, 0.0), 17.55, (-0.045, 0.0, 0.19), (0.22, 0.32, 0.40), 200.0, (-0.22, 0.73)),
    ]
    + _arm("left", +1.0)
    + _arm("right", -1.0)
    + [
        ("head_1_joint", "torso_2_joint", _RY, (0.0, 0.0, 0.316), 0.75, (0.0, 0.0, 0.03), (0.08, 0.08, 0.08), 8.0, (-0.21, 0.78)),
        ("head_2_joint", "head_1_joint", _RZ, (0.039, 0.0, 0.0), 1.35, (0.03, 0.0, 0.10), (0.16, 0.16, 0.20), 4.0, (-1.3, 1.3)),
    ]
)

_BASE_MASS, _BASE_COM, _BASE_BOX = 15.40, (-0.05, 0.0, -0.06), (0.24, 0.30, 0.22)
BASE_HEIGHT = 1.01927  # bullet_robot.py:22

# reference posture ("half_sitting"-like): legs, torso, arm_left(+gripper), arm_right(+gripper), head
_HALF_SITTING_JOINTS = (
    [0.0, 0.0, -0.411354, 0.859395, -0.448041, -0.001708]
    + [0.0, 0.0, -0.411354, 0.859395, -0.448041, -0.001708]
    + [0.0, 0.006761]
    + [0.25847, 0.173046, -0.0002, -0.525366, 0.0, 0.0, 0.1, 0.0]
    + [-0.25847, -0.173046, 0.0002, -0.525366, 0.0, 0.0, 0.1, 0.0]
    + [0.0, 0.0]
)

# joint ids locked by the reference (talos_utils.py:35-36): arm_{left,right}_5..7, both grippers, head
LOCKED_JOINT_IDS = [20, 21, 22, 23, 28, 29, 30, 31, 32, 33]


def build_complete_model():
    m = pin.Model()
    root = m.addJoint(0, "JointModelFreeFlyer", pin.SE3(), "root_joint")
    m.appendBodyToJoint(root, pin.Inertia(_BASE_MASS, _BASE_COM, _box(_BASE_MASS, *_BASE_BOX)))
    m.addFrame(pin.Frame("base_link", root, pin.SE3()))
    for name, parent, kind, trans, mass, com, dims, effort, (lo, hi) in _TREE:
        pid = m.getJointId(parent)
        jid = m.addJoint(pid, kind, pin.SE3(np.eye(3), trans), name, effort=effort, lower=lo, upper=hi)
        m.appendBodyToJoint(jid, pin.Inertia(mass, com, _box(mass, *dims)))
        m.addFrame(pin.Frame(name.replace("_joint", "_link"), jid, pin.SE3()))
    for side in ("left", "right"):
        jid = m.getJointId("leg_%s_6_joint" % side)
        m.addFrame(pin.Frame("%s_sole_link" % side, jid, pin.SE3(np.eye(3), (0.0, 0.0, -0.107))))
    q = pin.neutral(m)
    q[2] = BASE_HEIGHT
    q[7:] = _HALF_SITTING_JOINTS
    m.referenceConfigurations["half_sitting"] = q
    return m


def load_talos():
    """Same return tuple as ``loadTalos()`` (talos_utils.py:31-41): complete model, reduced model,
    complete reference configuration, reduced reference configuration."""
    complete = build_complete_model()
    q_complete = complete.referenceConfigurations["half_sitting"]
    reduced = pin.buildReducedModel(complete, LOCKED_JOINT_IDS, q_complete)
    return complete, reduced, q_complete, reduced.referenceConfigurations["half_sitting"]
