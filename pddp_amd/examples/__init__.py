"""Sample dynamics models (reference: pddp/examples/__init__.py)."""
from . import cartpole, double_cartpole, pendulum, rendezvous
from .problems import SampleProblems

__all__ = ["SampleProblems", "cartpole", "double_cartpole", "pendulum",
           "rendezvous"]
