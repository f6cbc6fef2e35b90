"""Dynamics models (reference: pddp/models/__init__.py, which exports the BNN;
`gp` is this build's own plugin for BASELINE.json's "GP dynamics" - the
reference has none)."""
from . import bnn, gp
from .base import DynamicsModel

__all__ = ["DynamicsModel", "bnn", "gp"]
