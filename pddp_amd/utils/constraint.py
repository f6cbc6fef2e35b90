"""Control constraints (reference: pddp/utils/constraint.py:35-48 `constrain`
- the BNN model factory's optional action squash, modules.py:117-123 -,
:51-143 the `constrain_env` / `constrain_model` class decorators, :146-147
`clamp`, :150-266 `boxqp`).

The box-constrained QP lives inside the HIP backward sweep (csrc/gains.hpp
`boxqp`, riccati_n4.hpp `QpClosed` / `BoxQp1`); `boxqp` below exposes the same
device routines as a callable: batches of problems with up to four dimensions.
"""
import torch

from .. import _native
from .linalg import cholesky_solve

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


def constrain(u, min_bounds, max_bounds):
    """Smooth squash of an unbounded action into the box: the tanh image of u
    scaled to [min_bounds, max_bounds] (differentiable everywhere)."""
    half = (max_bounds - min_bounds) / 2.0
    centre = (max_bounds + min_bounds) / 2.0
    return half * torch.tanh(u) + centre


def constrain_env(min_bounds, max_bounds):
    """Class decorator: the environment's `apply` squashes its action first."""
    def decorate(cls):
        inner = cls.apply

        def apply(self, u):
            return inner(self, constrain(u, min_bounds, max_bounds))
        cls.apply = apply
        return cls
    return decorate


def constrain_model(min_bounds, max_bounds):
    """Class decorator: the model's `forward` squashes its action first; the
    instances get `min_bounds` / `max_bounds` parameters and a `constrain`
    method (constraint.py:83-143; its `min_bounds` parameter is set from the
    upper bound there - here it holds the lower one)."""
    def decorate(cls):
        init, fwd = cls.__init__, cls.forward

        def __init__(self, *args, **kwargs):
            init(self, *args, **kwargs)
            # (tensor bounds keep their dtype, like the reference's
            # torch.tensor(bounds), constraint.py:100-102; Python scalars take
            # the default dtype)
            P = lambda b: torch.nn.Parameter(
                (b.detach() if torch.is_tensor(b) else torch.as_tensor(
                    b, dtype=torch.get_default_dtype())).expand(
                        cls.action_size).clone(), requires_grad=False)
            self.max_bounds, self.min_bounds = P(max_bounds), P(min_bounds)

        def forward(self, z, u, i, encoding=None, **kwargs):
            u = constrain(u, min_bounds, max_bounds)
            if encoding is None:
                return fwd(self, z, u, i, **kwargs)
            return fwd(self, z, u, i, encoding=encoding, **kwargs)

        cls.__init__, cls.forward = __init__, forward
        cls.constrain = lambda self, u: constrain(u, min_bounds, max_bounds)
        return cls
    return decorate


def clamp(u, min_bounds, max_bounds):
    return torch.min(torch.max(u, min_bounds), max_bounds)


def _boxqp_torch(x0, Q, c, lower, upper, max_iter=100, min_grad=1e-8,
                 tol=1e-8, step_dec=0.6, min_step=1e-22, armijo=0.1):
    """constraint.py:150-266 for ONE problem of any dimension in torch ops on
    the tensors' device (the device routines stop at four dimensions - the
    sweep kernels' m <= 4): projected Newton steps with an Armijo
    back-tracking line search along the clamped path."""
    obj = lambda x: 0.5 * (x @ Q @ x) + x @ c
    x = clamp(x0, lower, upper)
    x = torch.where(torch.isinf(x), torch.zeros_like(x), x)
    D = x.shape[0]
    clamped = torch.zeros(D, dtype=torch.bool, device=x.device)
    free = ~clamped
    Ufree = torch.zeros(D, D, dtype=x.dtype, device=x.device)
    f, f_old, result = obj(x), None, 0
    for it in range(max_iter):
        if it > 0 and bool((f_old - f) < tol * f_old.abs()):
            result = 4
            break
        f_old = f
        g = Q @ x + c
        was = clamped
        clamped = ((x == lower) & (g > 0)) | ((x == upper) & (g < 0))
        free = ~clamped
        if bool(clamped.all()):
            result = 6
            break
        if it == 0 or bool((was != clamped).any()):
            Ufree, info = torch.linalg.cholesky_ex(Q[free][:, free], upper=True)
            if int(info) != 0:
                result = -1
                break
        if float(g[free].norm()) < min_grad:
            result = 5
            break
        g_c = Q @ (x * clamped.to(x.dtype)) + c
        step_dir = torch.zeros_like(x)
        step_dir[free] = -cholesky_solve(
            g_c[free].unsqueeze(1), Ufree, upper=True).squeeze(1) - x[free]
        slope = (step_dir * g).sum()
        step = 1.0
        xc = clamp(x + step * step_dir, lower, upper)
        fc = obj(xc)
        while bool((fc - f_old) / (step * slope) < armijo):
            step *= step_dec
            xc = clamp(x + step * step_dir, lower, upper)
            fc = obj(xc)
            if step < min_step:
                result = 2
                break
        x, f = xc, fc
        if result != 0:
            break
    else:
        result = 1
    return x, result, Ufree, free.to(torch.uint8)


@torch.no_grad()
def boxqp(x0, Q, c, lower, upper, **kwargs):
    """constraint.py:150-266 on the GPU: min 0.5 x^T Q x + c^T x subject to
    lower <= x <= upper, warm-started at x0 (D <= 4: the sweep kernels' device
    routine, batched; larger un-batched problems: the same algorithm in torch
    ops on the GPU).

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
        if x0.dim() != 1:
            raise NotImplementedError(
                "batches: the device routine solves problems of up to four "
                "dimensions")
        opts = dict(dtype=Q.dtype, device=Q.device)
        b = lambda t: torch.as_tensor(t).to(**opts).expand(D)
        return _boxqp_torch(x0.to(**opts), Q, c.to(**opts), b(lower), b(upper))
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
