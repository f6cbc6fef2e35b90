"""Common utilities (reference: pddp/utils/__init__.py)."""
from . import angular, classproperty, constraint, encoding, gaussian_variable

__all__ = ["angular", "classproperty", "constraint", "encoding",
           "gaussian_variable"]
