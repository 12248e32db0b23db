"""FullStateLinear: state feedback u = K (r - x) as an input handler.

Reference: src/continuum_robot/control/full_state_linear.py:5-64 (same checks, same messages, in the same
order).  The batched counterpart is BeamEnsemble.step_feedback / crb_step_rk4_feedback.
"""
import numpy as np

from ..models.abstractions import AbstractInputHandler

# (predicate over (K, x, r), message) -- evaluated in this order, first hit raises
_INPUT_CHECKS = (
    (lambda K, x, r: r.ndim != 1, "Input vector r must be a 1D array."),
    (lambda K, x, r: x.ndim != 1, "State vector x must be a 1D array."),
    (lambda K, x, r: x.shape[0] != r.shape[0], "State vector and refrence vector must have the same length."),
    (lambda K, x, r: K.shape[1] != x.shape[0], "Gain matrix column dimension must match state vector length."),
)


class FullStateLinear(AbstractInputHandler):
    def __init__(self, gain_matrix: np.ndarray, enabled: bool = True):
        if gain_matrix.ndim != 2:
            raise ValueError("Gain matrix must be a 2D array.")
        self.gain_matrix, self.enabled = gain_matrix, enabled

    def is_enabled(self) -> bool:
        return self.enabled

    def compute_input(self, x: np.ndarray, r: np.ndarray, t: float) -> np.ndarray:
        for failed, message in _INPUT_CHECKS:
            if failed(self.gain_matrix, x, r):
                raise ValueError(message)
        return self.gain_matrix @ (r - x)
