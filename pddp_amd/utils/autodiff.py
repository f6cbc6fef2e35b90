"""Small autograd helpers with the reference's names and semantics
(reference: pddp/utils/autodiff.py:20-90): a gradient that is zero - not None -
for an input the output does not depend on, a row-by-row Jacobian, and the
one-pass Jacobian of a row-wise function (the replicate-the-input trick the
derivative rollout is built on, utils/evaluation.py:203-235,268-280)."""
import torch


def grad(y, x, allow_unused=True, **kwargs):
    """d y / d x for a scalar y; zeros where the graph does not reach x.  The
    graph is kept (so that rows of a Jacobian can be taken one after the
    other) and the result can itself be differentiated when `create_graph`
    is passed."""
    kwargs.setdefault("retain_graph", True)
    g = torch.autograd.grad(y, [x], allow_unused=allow_unused, **kwargs)[0]
    if g is None:
        g = torch.zeros_like(x)
    return g if g.requires_grad else g.requires_grad_()


def jacobian(y, x, **kwargs):
    """[len(y), len(x)] matrix of d y_r / d x, one backward pass per row."""
    rows = [grad(y_r, x, **kwargs) for y_r in y.unbind(0)]
    J = torch.stack(rows) if rows else x.new_zeros(0, x.shape[-1])
    return J if J.requires_grad else J.requires_grad_()


def batch_jacobian(f, x, m=None, **kwargs):
    """Jacobian [m, len(x)] of a function that maps rows to rows: x is copied
    m times, f runs once on the stack and ONE backward pass with the identity
    as cotangent pulls row r's gradient out of copy r."""
    if m is None:
        with torch.no_grad():
            m = int(f(x).shape[-1])
    stack = x.detach().unsqueeze(0).repeat(m, 1).requires_grad_()
    out = f(stack)
    seed = torch.eye(m, dtype=x.dtype, device=x.device)
    kwargs.setdefault("retain_graph", True)
    J = torch.autograd.grad(out, [stack], seed, allow_unused=True, **kwargs)[0]
    if J is None:
        J = torch.zeros(m, x.shape[-1], dtype=x.dtype, device=x.device)
    return J if J.requires_grad else J.requires_grad_()
