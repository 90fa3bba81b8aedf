"""Drop-in mirror of the ``aligator`` Python surface used by the reference scripts
(centroidal_talos.py, fulldynamic_talos.py, kinodynamic_talos.py — symbol list in SURVEY.md §8b-1),
backed by the MI355X-native solver library through the ctypes C-ABI of ``include/mpc_abi.h``.

Use as ``import mpc_benchmark_amd.aligator as aligator`` or call
``mpc_benchmark_amd.install_as_aligator()`` to make plain ``import aligator`` resolve here.
"""
from . import manifolds, dynamics, constraints  # noqa: F401
from ._core import (  # noqa: F401
    StageFunction, FunctionSlice, StateErrorResidual, ControlErrorResidual, FramePlacementResidual,
    FrameTranslationResidual, FrameVelocityResidual, CenterOfMassTranslationResidual, CentroidalMomentumResidual,
    CentroidalMomentumDerivativeResidual, ContactForceResidual, MultibodyWrenchConeResidual,
    CentroidalWrenchConeResidual, CentroidalAccelerationResidual, AngularAccelerationResidual,
    LinearMomentumResidual, AngularMomentumResidual, CentroidalCoMResidual, ContactMap,
    QuadraticResidualCost, QuadraticStateCost, QuadraticControlCost, CostStack,
    StageModel, StageConstraint, StageData, TrajOptProblem, wrench_cone_matrix,
)
from ._solver import (  # noqa: F401
    SolverProxDDP, Results, Workspace, VerboseLevel,
    ROLLOUT_LINEAR, ROLLOUT_NONLINEAR, LQ_SOLVER_SERIAL, LQ_SOLVER_PARALLEL, LQ_SOLVER_STAGEDENSE,
)

__version__ = "0.10.0+mi355x"
