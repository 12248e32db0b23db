"""Control layer of the drop-in package: stays in Python and drives the GPU stepper
(``BeamEnsemble.step_feedback`` takes the gain computed here)."""
from .full_state_linear import FullStateLinear
from .linear_quadratic_regulator import LinearQuadraticRegulator

__all__ = ["FullStateLinear", "LinearQuadraticRegulator"]
