"""Controllers (reference: pddp/controllers/__init__.py)."""
from .base import Controller
from .ilqr import iLQRController, iLQRState
from .pddp import PDDPController

__all__ = ["Controller", "iLQRController", "iLQRState", "PDDPController"]
