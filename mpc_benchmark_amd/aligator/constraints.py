"""``aligator.constraints`` mirror (fulldynamic_talos.py:207-225, 504)."""
from ._core import EqualityConstraintSet, NegativeOrthant, BoxConstraint  # noqa: F401
