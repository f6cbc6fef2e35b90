"""Control constraints (reference: pddp/utils/constraint.py:146-147 `clamp`).

The box-constrained QP of constraint.py:150-266 lives inside the HIP backward
sweep (pddp_amd/csrc/gains.hpp `boxqp`); it has no host-side twin.
"""
import torch

BOXQP_RESULTS = {
    -1: "Hessian is not positive definite",
    0: "No descent direction found",
    1: "Maximum main iterations exceeded",
    2: "Maximum line-search iterations exceeded",
    3: "No bounds, returning Newton point",
    4: "Improvement smaller than tolerance",
    5: "Gradient norm smaller than tolerance",
    6: "All dimensions are clamped",
}


def clamp(u, min_bounds, max_bounds):
    return torch.min(torch.max(u, min_bounds), max_bounds)
