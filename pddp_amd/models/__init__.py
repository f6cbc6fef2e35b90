"""Dynamics models (reference: pddp/models/__init__.py)."""
from .base import DynamicsModel

__all__ = ["DynamicsModel"]
