#!/usr/bin/env python3
"""BUILD-CONTAINER TOOL: golden vectors for the host helpers of the MPC loops, from the reference's OWN functions.

``talos_utils.shapeState`` and ``talos_utils.compute_ID_references`` are imported from the reference checkout (``/root/reference`` or
``$MPC_REFERENCE_DIR``) with ``pinocchio`` -> ``mpc_benchmark_amd.robot.minipin`` (tools/dropin/README.md) and called on seeded inputs;
inputs and outputs (numbers only) go to ``tests/golden/talos_utils_vectors.npz``.  tests/test_references.py holds
``mpc_benchmark_amd.references`` to them.  Nothing of the reference is copied."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MPC_REFERENCE_DIR", "/root/reference")
if not os.path.isdir(REF):
    sys.exit("no reference checkout: nothing to do")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools", "dropin"))
from mpc_benchmark_amd.robot import minipin  # noqa: E402
sys.modules["pinocchio"] = minipin
import example_robot_data, ndcurves  # noqa: E401,E402,F401  (stand-ins of tools/dropin: talos_utils imports them)
sys.path.insert(0, REF)
import talos_utils as ref  # noqa: E402
from mpc_benchmark_amd.aligator import manifolds  # noqa: E402
from mpc_benchmark_amd.problems.common import Robot  # noqa: E402

rng = np.random.default_rng(20251003)
out = {}
# shapeState: the full Talos of the simulator (nq 39, nv 38) reduced to the controlled joints of the scripts' reduced model
robot_full = Robot(complete=True)
robot_red = Robot(complete=False)
mf, mr = robot_full.model, robot_red.model
names_red = list(mr.names)[1:]
cj_ids = [int(mf.getJointId(n)) for n in names_red]
cases = []
for _ in range(4):
    q = minipin.integrate(mf, robot_full.q0, 0.2 * rng.standard_normal(mf.nv))
    v = rng.standard_normal(mf.nv)
    cases.append((q, v, np.asarray(ref.shapeState(q, v, mr.nq, mr.nq + mr.nv, cj_ids))))
out["shape_q"] = np.array([c[0] for c in cases]); out["shape_v"] = np.array([c[1] for c in cases]); out["shape_x"] = np.array([c[2] for c in cases])
out["shape_cj_ids"] = np.array(cj_ids); out["shape_nq"] = np.array(mr.nq); out["shape_nxq"] = np.array(mr.nq + mr.nv)
# compute_ID_references on the reduced model
space = manifolds.MultibodyPhaseSpace(mr)
data = mr.createData()
LF_id, RF_id = mr.getFrameId("left_sole_link"), mr.getFrameId("right_sole_link")
base_id, torso_id = mr.getFrameId("base_link"), mr.getFrameId("torso_2_link")
x0 = np.concatenate((robot_red.q0, np.zeros(mr.nv)))
res = []
xs, refs = [], []
for _ in range(4):
    x = space.integrate(x0, 0.05 * rng.standard_normal(2 * mr.nv))
    minipin.forwardKinematics(mr, data, x[:mr.nq], x[mr.nq:])
    minipin.updateFramePlacements(mr, data)
    def pose(fid, d):
        M = data.oMf[fid]
        return minipin.SE3(M.rotation @ minipin.exp3(d[3:]), M.translation + d[:3])
    LF_refs = [pose(LF_id, 0.01 * rng.standard_normal(6)), pose(LF_id, 0.01 * rng.standard_normal(6))]
    RF_refs = [pose(RF_id, 0.01 * rng.standard_normal(6)), pose(RF_id, 0.01 * rng.standard_normal(6))]
    r = ref.compute_ID_references(space, mr, data, LF_id, RF_id, base_id, torso_id, x0, x, LF_refs, RF_refs, 0.001)
    res.append(np.concatenate([np.asarray(v, dtype=float).reshape(-1) for v in r]))
    xs.append(x)
    refs.append(np.concatenate([np.concatenate((M.rotation.reshape(-1), M.translation)) for M in LF_refs + RF_refs]))
out["id_x0"] = x0; out["id_x"] = np.array(xs); out["id_refs"] = np.array(refs); out["id_out"] = np.array(res)
out["id_frames"] = np.array([LF_id, RF_id, base_id, torso_id])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "talos_utils_vectors.npz"), **out)
print("wrote tests/golden/talos_utils_vectors.npz:", {k: v.shape for k, v in out.items()})
