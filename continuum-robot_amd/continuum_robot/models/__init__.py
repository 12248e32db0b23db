"""Model layer of the drop-in package.

Public names of the reference's ``continuum_robot.models`` (DynamicEulerBernoulliBeam, EulerBernoulliBeam,
GravityForce and the abstraction types) plus the remaining pieces of the force-composition API, so that
``from continuum_robot.models import X`` works for every class the package defines.
"""
from .abstractions import AbstractForce, AbstractInputHandler, BoundaryConditionType, ElementType, IBeam, ISegment
from .abstractions import Properties, create_properties_from_dataframe
from .force_params import ForceParams
from .force_registry import ForceRegistry, InputRegistry
from .fluid_forces import FluidDragForce
from .gravity_forces import GravityForce
from .segments import LinearSegment, NonlinearSegment, SegmentFactory
from .euler_bernoulli_beam import EulerBernoulliBeam
from .dynamic_beam_model import DynamicEulerBernoulliBeam

__all__ = [
    "AbstractForce", "AbstractInputHandler", "BoundaryConditionType", "DynamicEulerBernoulliBeam", "ElementType",
    "EulerBernoulliBeam", "FluidDragForce", "ForceParams", "ForceRegistry", "GravityForce", "IBeam", "ISegment",
    "InputRegistry", "LinearSegment", "NonlinearSegment", "Properties", "SegmentFactory",
    "create_properties_from_dataframe",
]
