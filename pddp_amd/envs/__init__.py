"""Environments (reference: pddp/envs/__init__.py)."""
from .base import Env, ModelEnv

__all__ = ["Env", "ModelEnv"]
