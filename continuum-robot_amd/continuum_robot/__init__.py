"""continuum_robot -- MI355X-native drop-in for the beam-dynamics path of cram9030/continuum-robot.

Same import paths as the reference package (reference: src/continuum_robot/__init__.py:1-9); the
element assembly, force evaluators, mass solve and time step run as HIP kernels behind
libcrbeam.so (see include/crbeam.h).  ``continuum_robot.batched.BeamEnsemble`` is the batched
entry point the planning/control layers call.
"""
from continuum_robot.models.dynamic_beam_model import DynamicEulerBernoulliBeam
from continuum_robot.models.euler_bernoulli_beam import EulerBernoulliBeam
from continuum_robot.models.abstractions import (
    IBeam,
    ISegment,
    Properties,
    ElementType,
    BoundaryConditionType,
)
