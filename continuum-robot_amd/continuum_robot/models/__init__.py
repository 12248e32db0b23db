"""Reference: src/continuum_robot/models/__init__.py:1-10 (same re-exports)."""
from .dynamic_beam_model import DynamicEulerBernoulliBeam
from .euler_bernoulli_beam import EulerBernoulliBeam
from .gravity_forces import GravityForce
from .abstractions import (
    IBeam,
    ISegment,
    Properties,
    ElementType,
    BoundaryConditionType,
)
