"""LinearSegment / NonlinearSegment / SegmentFactory of the drop-in API.

Reference: src/continuum_robot/models/segments.py (LinearSegment :8-78, NonlinearSegment :81-472,
SegmentFactory :475-491).  No element formula lives in this file: the 6x6 mass and stiffness
matrices come from the native library's assembly (crb_plan_get_mass / crb_plan_get_stiffness of a
one-element, unconstrained plan -- the same code that feeds the kernels), and the nonlinear
internal-force callable runs the element kernel on the GPU (crb_internal_force).
"""
from typing import Callable, Union

import numpy as np

from .. import _native as nat
from .abstractions import ElementType, ISegment, ISegmentFactory, Properties


def _one_element_columns(props: Properties):
    return {
        "length": [props.length],
        "elastic_modulus": [props.elastic_modulus],
        "moment_inertia": [props.moment_inertia],
        "density": [props.density],
        "cross_area": [props.cross_area],
        "type": [props.element_type],
        "boundary_condition": ["NONE"],
    }


class _SegmentBase(ISegment):
    _required_type = None

    def __init__(self, properties: Properties):
        super().__init__(properties)
        if properties.get_element_type() != self._required_type:
            raise ValueError(
                f"{type(self).__name__} requires {self._required_type.name} element type, "
                f"got {properties.element_type}")
        self._host_plan = None

    def _plan(self):
        if self._host_plan is None:
            self._host_plan = nat.Plan(_one_element_columns(self.properties), device=-1)
        return self._host_plan

    def get_mass_matrix(self) -> np.ndarray:
        """Consistent 6x6 mass matrix in the order [u1, w1, phi1, u2, w2, phi2]."""
        return self._plan().mass()

    def get_element_type(self) -> ElementType:
        return self._required_type


class LinearSegment(_SegmentBase):
    _required_type = ElementType.LINEAR

    def get_stiffness_func(self) -> Union[np.ndarray, Callable[[np.ndarray], np.ndarray]]:
        """Constant 6x6 stiffness matrix."""
        return self._plan().stiffness()


class NonlinearSegment(_SegmentBase):
    _required_type = ElementType.NONLINEAR

    def get_stiffness_func(self) -> Union[np.ndarray, Callable[[np.ndarray], np.ndarray]]:
        """Callable x_e[6] -> internal force [6] of the von-Karman element, evaluated by the
        element kernel on the GPU (raises without a HIP device)."""
        from ..batched import BeamEnsemble  # deferred: needs torch + GPU only when called

        cols = _one_element_columns(self.properties)
        holder = {}

        def stiffness_func(x: np.ndarray) -> np.ndarray:
            if "ens" not in holder:
                holder["ens"] = BeamEnsemble(cols, 1)
            q = np.asarray(x, dtype=np.float64).reshape(1, 6)
            return holder["ens"].internal_force(q).cpu().numpy()[0]

        return stiffness_func


class SegmentFactory(ISegmentFactory):
    def create_segment(self, properties: Properties) -> ISegment:
        kind = self.detect_element_type(properties)
        if kind == ElementType.LINEAR:
            return LinearSegment(properties)
        if kind == ElementType.NONLINEAR:
            return NonlinearSegment(properties)
        raise ValueError(f"Unknown element type: {kind}")

    def detect_element_type(self, properties: Properties) -> ElementType:
        return properties.get_element_type()
