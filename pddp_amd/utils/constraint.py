"""Control constraints (reference: pddp/utils/constraint.py:146-147 `clamp`,
:150-266 `boxqp`).

The box-constrained QP lives inside the HIP backward sweep (csrc/gains.hpp
`boxqp`, riccati_n4.hpp `QpClosed` / `BoxQp1`); `boxqp` below exposes the same
device routines as a callable: batches of problems with up to four dimensions.
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
    """constraint.py:150-266 on the GPU: min 0.5 x^T Q x + c^T x subject to
    lower <= x <= upper, warm-started at x0, for D <= 4 dimensions.

    Un-batched like the reference - x0, c, lower, upper of shape (D,), Q (D, D)
    - it returns (x, result, Ufree, free) with `result` an int (BOXQP_RESULTS),
    `free` a uint8 mask (D,) and `Ufree` the upper Cholesky factor of the free
    block (n_free, n_free), exactly the reference's tuple.  With leading batch
    dimensions (..., D) / (..., D, D) every problem is solved by one lane of
    the same kernel; `result` and `free` then carry the batch shape and
    `Ufree` is (..., D, D) with identity rows / columns in place of the
    clamped dimensions (a batch cannot hold blocks of different sizes).
    Scalar problems may also be given without the trailing axis."""
    _native.require_gpu(Q)
    if Q.dim() == x0.dim():  # scalar problems without the trailing axes
        x0, c = x0.unsqueeze(-1), c.unsqueeze(-1)
        Q = Q.unsqueeze(-1).unsqueeze(-1)
        lower, upper = lower.unsqueeze(-1), upper.unsqueeze(-1)
        x, result, U, free = boxqp(x0, Q, c, lower, upper)
        return x.squeeze(-1), result, U.squeeze(-1).squeeze(-1), \
            free.squeeze(-1)
    D = x0.shape[-1]
    if D > 4:
        raise NotImplementedError(
            "the device routine solves problems of up to four dimensions")
    batch = x0.shape[:-1]
    opts = dict(dtype=Q.dtype, device=Q.device)
    flat = lambda t, *s: t.to(**opts).expand(*batch, *s).reshape(
        -1, *s).contiguous()
    x0f, cf = flat(x0, D), flat(c, D)
    lf, uf = flat(lower, D), flat(upper, D)
    Qf = flat(Q, D, D)
    n = x0f.shape[0]
    x = torch.empty_like(x0f)
    U = torch.empty_like(Qf)
    result = torch.empty(n, dtype=torch.int32, device=Q.device)
    free = torch.empty(n, D, dtype=torch.uint8, device=Q.device)
    p = _native.ptr
    if D == 1:  # the sweep kernels' own scalar routine (riccati_n4.hpp)
        _native.call("pddp_boxqp_m1", Q.dtype, n, p(x0f), p(Qf), p(cf), p(lf),
                     p(uf), p(x), p(result), p(free),
                     _native.stream_handle(Q.device))
        U = Qf.sqrt()
    else:
        _native.call("pddp_boxqp", Q.dtype, n, D, p(x0f), p(Qf), p(cf),
                     p(lf), p(uf), p(x), p(result), p(U), p(free),
                     _native.stream_handle(Q.device))
    if len(batch) == 0:
        keep = free[0].bool()
        return x[0], int(result[0]), U[0][keep][:, keep], free[0]
    return (x.reshape(*batch, D), result.reshape(batch),
            U.reshape(*batch, D, D), free.reshape(*batch, D))
