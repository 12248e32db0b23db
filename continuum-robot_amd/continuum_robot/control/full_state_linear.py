"""FullStateLinear: state feedback u = K (r - x) as an input handler.

Reference: src/continuum_robot/control/full_state_linear.py:5-64 (same checks and messages).
"""
import numpy as np

from ..models.abstractions import AbstractInputHandler


class FullStateLinear(AbstractInputHandler):
    def __init__(self, gain_matrix: np.ndarray, enabled: bool = True):
        if gain_matrix.ndim != 2:
            raise ValueError("Gain matrix must be a 2D array.")
        self.gain_matrix = gain_matrix
        self.enabled = enabled

    def compute_input(self, x: np.ndarray, r: np.ndarray, t: float) -> np.ndarray:
        if r.ndim != 1:
            raise ValueError("Input vector r must be a 1D array.")
        if x.ndim != 1:
            raise ValueError("State vector x must be a 1D array.")
        if x.shape[0] != r.shape[0]:
            raise ValueError("State vector and refrence vector must have the same length.")
        if self.gain_matrix.shape[1] != x.shape[0]:
            raise ValueError("Gain matrix column dimension must match state vector length.")
        return self.gain_matrix @ (r - x)

    def is_enabled(self) -> bool:
        return self.enabled
