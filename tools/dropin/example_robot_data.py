"""Stand-in for ``example_robot_data`` (talos_utils.py:5, 12, 32): the Talos URDF is not available in the build environment, so
``load("talos")`` hands out the synthetic Talos of mpc_benchmark_amd/robot/talos_synth.py (same joint order and names as Pinocchio builds
from talos_reduced.urdf).  Build-container tooling: see tools/dropin/README.md."""
from mpc_benchmark_amd.robot import minipin as _pin
from mpc_benchmark_amd.robot import talos_synth as _ts


def getModelPath(subpath, verbose=False):
    return "/nonexistent/example-robot-data/robots"


class _RobotWrapper:
    def __init__(self, model):
        self.model = model
        self.data = model.createData()
        self.q0 = model.referenceConfigurations.get("half_sitting")

    def buildReducedRobot(self, list_of_joints_to_lock, reference_configuration=None):
        q = self.q0 if reference_configuration is None else reference_configuration
        return _RobotWrapper(_pin.buildReducedModel(self.model, list_of_joints_to_lock, q))


def load(name, *a, **k):
    if name != "talos":
        raise ValueError("stand-in example_robot_data only knows 'talos'")
    return _RobotWrapper(_ts.build_complete_model())
