"""N4: the reference's .npz run-log format and the centre-of-pressure helper (talos_utils.py:113-185, plot.py:22-96)."""
import numpy as np

from mpc_benchmark_amd import trajectory_log as tl
from mpc_benchmark_amd.robot import minipin as pin


def test_archive_round_trip_has_the_reference_fields(tmp_path):
    log = tl.TrajectoryLog()
    rng = np.random.default_rng(0)
    yawed = np.array([[np.cos(0.2), -np.sin(0.2), 0.0], [np.sin(0.2), np.cos(0.2), 0.0], [0.0, 0.0, 1.0]])
    for t in range(5):
        lf = pin.SE3(yawed, np.array([0.0, 0.09, 0.0]))
        rf = pin.SE3(np.eye(3), np.array([0.0, -0.09, 0.0]))
        log.append(rng.normal(size=77), rng.normal(size=32), rng.normal(size=3), rng.normal(size=(2, 6)), lf, rf, lf, rf, t=0.01 * t)
    path = log.save("run", str(tmp_path))
    d = tl.load_data(path)
    assert set(d) == set(tl.FIELDS)  # exactly what plot.py indexes
    assert np.array(d["xs"]).shape == (5, 77) and np.array(d["us"]).shape == (5, 32)
    assert np.allclose(np.array(d["time"]), 0.01 * np.arange(5))
    # plot.py's access pattern (plot.py:138-144): pin.SE3(pose[i]).translation, and .rotation inside computeCoP
    for i in range(5):
        lf_se3, rf_se3 = pin.SE3(d["LF_pose"][i]), pin.SE3(d["RF_pose"][i])
        assert np.allclose(lf_se3.translation, [0.0, 0.09, 0.0]) and np.allclose(lf_se3.rotation, yawed)
        assert np.allclose(pin.SE3(d["RF_pose_ref"][i]).translation, [0.0, -0.09, 0.0])
        cop = tl.computeCoP(lf_se3, rf_se3, d["LF_force"][i], d["LF_torque"][i], d["RF_force"][i], d["RF_torque"][i])
        assert cop.shape == (3,)
    # the raw container is the one the reference writes: a pickled dict under "data"
    with np.load(path, allow_pickle=True) as z:
        assert list(z.keys()) == ["data"]


def test_cop_known_answers():
    yaw = 0.3
    R = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1.0]])
    lf = pin.SE3(R, np.array([0.1, 0.09, 0.0]))
    rf = pin.SE3(np.eye(3), np.array([-0.05, -0.09, 0.0]))
    # single support: the CoP is the left foot's local CoP moved to the world
    f, tau = np.array([0.0, 0.0, 500.0]), np.array([10.0, -20.0, 0.0])
    cop = tl.compute_cop(lf, rf, f, tau, np.zeros(3), np.zeros(3))
    assert np.allclose(cop, R @ np.array([0.04, 0.02, 0.0]) + lf.translation)
    # double support with pure normal forces: the force-weighted mean of the foot positions
    cop = tl.computeCoP(lf, rf, np.array([0, 0, 300.0]), np.zeros(3), np.array([0, 0, 100.0]), np.zeros(3))
    assert np.allclose(cop, 0.75 * lf.translation + 0.25 * rf.translation)
    # nothing loaded
    assert np.all(np.isnan(tl.compute_cop(lf, rf, np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3))))
