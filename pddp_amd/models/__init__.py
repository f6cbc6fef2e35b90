"""Dynamics models (reference: pddp/models/__init__.py)."""
from . import bnn
from .base import DynamicsModel

__all__ = ["DynamicsModel", "bnn"]
