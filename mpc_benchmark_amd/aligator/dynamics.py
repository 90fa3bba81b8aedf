"""``aligator.dynamics`` mirror (fulldynamic_talos.py:103-110, kinodynamic_talos.py:108-111, centroidal_talos.py:203-204)."""
from ._core import (  # noqa: F401
    MultibodyConstraintFwdDynamics, KinodynamicsFwdDynamics, CentroidalFwdDynamics,
    IntegratorSemiImplEuler, IntegratorEuler,
)
