"""Control layer (stays Python and calls the stepper).  Reference: src/continuum_robot/control/__init__.py:1-4."""
from .linear_quadratic_regulator import LinearQuadraticRegulator
from .full_state_linear import FullStateLinear

__all__ = ["LinearQuadraticRegulator", "FullStateLinear"]
