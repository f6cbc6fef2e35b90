"""Cost functions (reference: pddp/costs/__init__.py)."""
from .base import AggregateCost, Cost
from .quadratic import QRCost

__all__ = ["AggregateCost", "Cost", "QRCost"]
