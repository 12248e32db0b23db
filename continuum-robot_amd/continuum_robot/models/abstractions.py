"""Types of the drop-in API: enums, the per-segment property record and the abstract interfaces.

Mirrors the public surface of the reference's src/continuum_robot/models/abstractions.py
(enums :9-20, Properties :23-67, interfaces :79-197, create_properties_from_dataframe :200-233):
same names, fields, defaults and error messages, so user code and tests written against the
reference import and behave the same.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from dataclasses import dataclass
from enum import Enum
from typing import Callable, Dict, List, Optional, Union

import numpy as np
import pandas as pd


class ElementType(Enum):
    LINEAR = "linear"
    NONLINEAR = "nonlinear"


class BoundaryConditionType(Enum):
    FIXED = "fixed"    # u, w and phi of the node are constrained
    PINNED = "pinned"  # u and w constrained, phi free


# ---------------------------------------------------------------- composition interfaces
class AbstractForce(ABC):
    """A force term f(x, t) -> ndarray over the position DOFs."""

    @abstractmethod
    def compute_forces(self, x: np.ndarray, t: float) -> np.ndarray:
        ...

    @abstractmethod
    def is_enabled(self) -> bool:
        ...


class AbstractInputHandler(ABC):
    """An additive modification of the input vector."""

    @abstractmethod
    def compute_input(self, x: np.ndarray, r: np.ndarray, t: float) -> np.ndarray:
        ...

    @abstractmethod
    def is_enabled(self) -> bool:
        ...


# ---------------------------------------------------------------- structural interfaces
class IBeam(ABC):
    def __init__(self, segments: List[ISegment]):
        self.segments = segments

    @abstractmethod
    def assemble_mass_matrix(self) -> np.ndarray:
        ...

    @abstractmethod
    def create_stiffness_function(self) -> Callable:
        ...

    @abstractmethod
    def apply_boundary_conditions(self, boundary_conditions: Dict) -> None:
        ...

    @abstractmethod
    def get_constrained_dofs(self) -> List[int]:
        ...


class ISegmentFactory(ABC):
    @abstractmethod
    def create_segment(self, properties: Properties) -> ISegment:
        ...

    @abstractmethod
    def detect_element_type(self, properties: Properties) -> ElementType:
        ...


class ISegment(ABC):
    """A two-node element with DOFs [u1, w1, phi1, u2, w2, phi2]."""

    def __init__(self, properties: Properties):
        self.properties = properties
        self.segment_id = properties.segment_id

    @abstractmethod
    def get_mass_matrix(self) -> np.ndarray:
        ...

    @abstractmethod
    def get_stiffness_func(self) -> Union[np.ndarray, Callable[[np.ndarray], np.ndarray]]:
        ...

    @abstractmethod
    def get_element_type(self) -> ElementType:
        ...

    def validate_properties(self) -> None:
        return None  # Properties validates itself on construction

    def get_properties(self) -> Properties:
        return self.properties


@dataclass
class AssemblyContext:
    global_dof_offset: int
    node_start: int
    node_end: int


# ---------------------------------------------------------------- per-segment record
_POSITIVE_FIELDS = (
    ("length", "Length"),
    ("elastic_modulus", "Elastic modulus"),
    ("moment_inertia", "Moment of inertia"),
    ("density", "Density"),
    ("cross_area", "Cross area"),
)


@dataclass
class Properties:
    """One row of the beam CSV, validated."""

    length: float
    elastic_modulus: float
    moment_inertia: float
    density: float
    cross_area: float
    segment_id: int
    element_type: str
    wetted_area: Optional[float] = None
    drag_coef: Optional[float] = None

    def __post_init__(self):
        for attr, label in _POSITIVE_FIELDS:
            value = getattr(self, attr)
            if value <= 0:
                raise ValueError(f"{label} must be positive, got {value}")
        if self.element_type.lower() not in {t.value for t in ElementType}:
            raise ValueError(f"Invalid element type: {self.element_type}")

    def get_element_type(self) -> ElementType:
        return ElementType(self.element_type.lower())

    def has_fluid_properties(self) -> bool:
        return self.wetted_area is not None and self.drag_coef is not None


def create_properties_from_dataframe(df: pd.DataFrame, segment_id: int) -> Properties:
    if segment_id >= len(df):
        raise IndexError(f"Segment ID {segment_id} exceeds DataFrame length {len(df)}")
    row = df.iloc[segment_id]
    optional = {name: row[name] for name in ("wetted_area", "drag_coef") if name in df.columns}
    return Properties(
        length=row["length"],
        elastic_modulus=row["elastic_modulus"],
        moment_inertia=row["moment_inertia"],
        density=row["density"],
        cross_area=row["cross_area"],
        segment_id=segment_id,
        element_type=row["type"],
        **optional,
    )
