"""FluidDragForce: quadratic drag on the transverse DOFs (API object of the drop-in).

Reference: src/continuum_robot/models/fluid_forces.py:24-142.  The batched stepper applies the same
term inside the RHS kernel (crb_math.h drag_force, per-node factor built by crb_plan_create); this
class is the host-side API object of the functional-composition interface: it receives numpy and
returns numpy, like every user-supplied force callable.
"""
import numpy as np

from .abstractions import AbstractForce


class FluidDynamicsParams:
    """Legacy container kept for import compatibility (reference :5-21)."""

    def __init__(self, fluid_density: float = 0.0, enable_fluid_effects: bool = False):
        self.fluid_density = fluid_density
        self.enable_fluid_effects = enable_fluid_effects

    def __bool__(self) -> bool:
        return self.enable_fluid_effects


class FluidDragForce(AbstractForce):
    def __init__(self, fluid_data, state_mapping, fluid_density, enabled=True):
        self.fluid_data = fluid_data
        self.state_mapping = state_mapping
        self.fluid_density = fluid_density
        self.enabled = enabled
        self.fluid_coefficients = None
        if self.is_enabled():
            self._precompute_fluid_coefficients()

    def is_enabled(self) -> bool:
        return self.enabled

    def _precompute_fluid_coefficients(self) -> None:
        if not self.is_enabled():
            return
        # per NODE: the node's own segment row, the tip node repeating the last row (reference :59-61)
        wet = np.asarray(self.fluid_data["wetted_area"].values, dtype=float)
        cd = np.asarray(self.fluid_data["drag_coef"].values, dtype=float)
        wet = np.append(wet, wet[-1])
        cd = np.append(cd, cd[-1])
        n_nodes = wet.size
        vel_of, pos_of = {}, {}
        for idx, (param, node) in self.state_mapping.items():
            if node < n_nodes:
                if param == "dw_dt":
                    vel_of[node] = idx
                elif param == "w":
                    pos_of[node] = idx
        nodes = sorted(set(vel_of) & set(pos_of))
        self.fluid_coefficients = {
            "w_vel_indices": [vel_of[k] for k in nodes],
            "w_pos_indices": [pos_of[k] for k in nodes],
            "drag_factors": [0.5 * self.fluid_density * cd[k] * wet[k] for k in nodes],
            "n_pos_states": len(self.state_mapping) // 2,
        }

    def compute_forces(self, x: np.ndarray, t: float) -> np.ndarray:
        n = len(x) // 2
        out = np.zeros(n)
        if not self.is_enabled() or self.fluid_coefficients is None:
            return out
        fc = self.fluid_coefficients
        vel = np.asarray(x)[fc["w_vel_indices"]]
        out[fc["w_pos_indices"]] = -np.asarray(fc["drag_factors"]) * vel * np.abs(vel)
        return out
