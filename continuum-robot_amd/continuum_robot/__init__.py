"""continuum_robot -- MI355X-native drop-in for the beam-dynamics path of cram9030/continuum-robot.

The package keeps the reference's import paths (``continuum_robot.models...``, ``continuum_robot.control``)
and top-level names; the element assembly, force evaluators, mass solve and time step behind them run as
HIP kernels in libcrbeam.so (include/crbeam.h).  ``continuum_robot.batched.BeamEnsemble`` is the batched
entry point for rollouts of many beams; ``continuum_robot.distributed`` shards ensembles over GPUs.
"""
from .models import (BoundaryConditionType, DynamicEulerBernoulliBeam, ElementType, EulerBernoulliBeam, IBeam, ISegment,
                     Properties)

__all__ = ["BoundaryConditionType", "DynamicEulerBernoulliBeam", "ElementType", "EulerBernoulliBeam", "IBeam",
           "ISegment", "Properties"]
