"""Common utilities (reference: pddp/utils/__init__.py)."""
from . import (angular, autodiff, classproperty, constraint, encoding,
               evaluation, gaussian_variable, particles, trajectory)

__all__ = ["angular", "autodiff", "classproperty", "constraint", "encoding",
           "evaluation", "gaussian_variable", "particles", "trajectory"]
