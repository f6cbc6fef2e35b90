"""Common utilities (reference: pddp/utils/__init__.py).  The reference's
`trajectory` helpers and the `constrain_env` / `constrain_model` decorators are
outside the hot path (out of scope) and are
not provided."""
from . import (angular, autodiff, classproperty, constraint, encoding,
               evaluation, gaussian_variable, particles)

__all__ = ["angular", "autodiff", "classproperty", "constraint", "encoding",
           "evaluation", "gaussian_variable", "particles"]
