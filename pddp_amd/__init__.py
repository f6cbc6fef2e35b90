"""pddp_amd - MI355X-native PDDP / iLQR hot path behind the plugin API of
anassinator/pddp (controllers / costs / models / envs / utils).

The data-parallel path - derivative records, backward Riccati sweep, batched
line search, accept / regularisation state machine - runs in hand-written HIP
kernels for gfx950 (pddp_amd/csrc) behind the C ABI of include/pddp_hip.h.
"""
__version__ = "0.1.0"

from . import controllers, costs, envs, examples, models, utils
from .utils.encoding import StateEncoding
from .utils.gaussian_variable import GaussianVariable

__all__ = ["controllers", "costs", "envs", "examples", "models", "utils",
           "GaussianVariable", "StateEncoding"]
