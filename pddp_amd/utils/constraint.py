"""Control constraints (reference: pddp/utils/constraint.py:146-147 `clamp`).

The box-constrained QP of constraint.py:150-266 lives inside the HIP backward
sweep (pddp_amd/csrc/gains.hpp `boxqp`, riccati_n4.hpp `boxqp1`); `boxqp` below
exposes the one-dimensional device routine for batches of scalar problems.
"""
import torch

from .. import _native

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


@torch.no_grad()
def boxqp(x0, Q, c, lower, upper, **kwargs):
    """Batched scalar BoxQP on the GPU (constraint.py:150-266 with D = 1).

    x0, c, lower, upper: tensors of shape (..., 1) or (...); Q of shape
    (..., 1, 1) or (...).  Returns (x, result, Ufree, free) like the reference,
    with a leading batch shape; Ufree = sqrt(Q)."""
    _native.require_gpu(Q)
    shape = x0.shape
    flat = lambda t: t.reshape(-1).contiguous()
    x0f, Qf, cf = flat(x0), flat(Q), flat(c)
    lf, uf = flat(lower.expand_as(x0)), flat(upper.expand_as(x0))
    n = x0f.numel()
    if Qf.numel() != n:
        raise NotImplementedError(
            "the device routine solves one-dimensional problems (m = 1)")
    x = torch.empty_like(x0f)
    result = torch.empty(n, dtype=torch.int32, device=Q.device)
    free = torch.empty(n, dtype=torch.uint8, device=Q.device)
    p = _native.ptr
    _native.call("pddp_boxqp_m1", Q.dtype, n, p(x0f), p(Qf), p(cf), p(lf),
                 p(uf), p(x), p(result), p(free),
                 _native.stream_handle(Q.device))
    return x.reshape(shape), result, Qf.sqrt().reshape(Q.shape), free
