"""ForceParams -- configuration object of the force terms (drop-in for the reference's
src/continuum_robot/models/force_params.py:6-69: same fields, defaults, validation messages and the
auto-disable of an all-zero gravity vector)."""
from dataclasses import dataclass, field
from typing import List

import numpy as np


def _as_gravity(vec) -> np.ndarray:
    g = np.array(vec, dtype=float)
    if g.shape != (3,):
        raise ValueError("gravity_vector must have exactly 3 components [gx, gy, gz]")
    return g


@dataclass
class ForceParams:
    """Which force terms a beam model carries and with what constants."""

    fluid_density: float = 0.0
    enable_fluid_effects: bool = False
    gravity_vector: List[float] = field(default_factory=lambda: [0.0, -9.81, 0.0])
    enable_gravity_effects: bool = False

    def __post_init__(self):
        self.gravity_vector = _as_gravity(self.gravity_vector)
        if not self.gravity_vector.any():
            # a zero vector cannot produce a force: the reference switches the term off (:29-31)
            self.enable_gravity_effects = False
        if self.enable_fluid_effects and self.fluid_density <= 0:
            raise ValueError("fluid_density must be positive when fluid effects are enabled")

    def __bool__(self) -> bool:
        return bool(self.enable_fluid_effects or self.enable_gravity_effects)

    def get_gravity_vector(self) -> np.ndarray:
        return self.gravity_vector.copy()

    def set_gravity_vector(self, gravity_vector: List[float]) -> None:
        self.gravity_vector = _as_gravity(gravity_vector)
        if not self.gravity_vector.any():
            self.enable_gravity_effects = False
