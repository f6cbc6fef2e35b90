"""Common utilities (reference: pddp/utils/__init__.py:17-22 - the same
sub-modules under the same names)."""
from . import (angular, autodiff, classproperty, constraint, encoding,
               evaluation, gaussian_variable, particles, trajectory)

__all__ = ["angular", "autodiff", "classproperty", "constraint", "encoding",
           "evaluation", "gaussian_variable", "particles", "trajectory"]
