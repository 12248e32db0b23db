"""GravityForce: weight of each segment, rotated into the local frame (API object of the drop-in).

Reference: src/continuum_robot/models/gravity_forces.py:6-173.  Bug-compatible by design
(SURVEY App. B-2): the segment loop addresses the state vector it is handed with the index
arithmetic of the UNCONSTRAINED layout (3*i .. 3*i+5) and drops whatever falls outside, so on a
reduced state the load lands one node outboard.  The batched stepper reproduces exactly this
through its index table (crb_plan_create); this class is the host-side API object.
"""
from typing import List, Optional

import numpy as np

from .abstractions import AbstractForce


class GravityForce(AbstractForce):
    def __init__(self, beam_params, gravity_vector: Optional[List[float]] = None, enabled: bool = True):
        self.beam_params = beam_params
        self.gravity_vector = np.array(gravity_vector if gravity_vector is not None else [0.0, -9.81, 0.0])
        self.enabled = enabled
        if len(self.gravity_vector) != 3:
            raise ValueError("Gravity vector must have exactly 3 components [gx, gy, gz]")
        self._precompute_segment_masses()

    def _precompute_segment_masses(self):
        if not self.enabled:
            self._segment_masses = []
            return
        bp = self.beam_params
        self._segment_masses = [float(r["density"] * r["cross_area"] * r["length"]) for _, r in bp.iterrows()]

    def compute_forces(self, x: np.ndarray, t: float) -> np.ndarray:
        n = len(x) // 2
        forces = np.zeros(n)
        if not self._segment_masses:
            raise RuntimeError(
                "Cannot compute gravity forces: beam instance does not have segments available "
                "or segment masses were not pre-computed.")
        q = np.asarray(x)[:n]
        mass = np.asarray(self._segment_masses)
        seg = np.arange(mass.size)
        a_idx, b_idx = 3 * seg + 2, 3 * seg + 5           # the two rotations a segment averages
        has_a, has_b = a_idx < n, b_idx < n
        qa = np.where(has_a, q[np.minimum(a_idx, n - 1)], 0.0)
        qb = np.where(has_b, q[np.minimum(b_idx, n - 1)], 0.0)
        phi = np.where(has_a & has_b, 0.5 * (qa + qb), np.where(has_a, qa, qb))
        gx, gy = self.gravity_vector[0], self.gravity_vector[1]
        c, s = np.cos(phi), np.sin(phi)
        axial = (c * gx + s * gy) * mass * 0.5
        transverse = (-s * gx + c * gy) * mass * 0.5
        for offset, part in ((0, axial), (1, transverse), (3, axial), (4, transverse)):
            idx = 3 * seg + offset
            ok = idx < n
            np.add.at(forces, idx[ok], part[ok])
        return forces

    def is_enabled(self) -> bool:
        return self.enabled

    def set_enabled(self, enabled: bool) -> None:
        self.enabled = enabled

    def set_gravity_vector(self, gravity_vector: List[float]) -> None:
        if len(gravity_vector) != 3:
            raise ValueError("Gravity vector must have exactly 3 components [gx, gy, gz]")
        self.gravity_vector = np.array(gravity_vector)

    def get_gravity_vector(self) -> np.ndarray:
        return self.gravity_vector.copy()
