"""Value and derivatives of a cost / dynamics plugin at ONE (z, u) - the
reference's public helpers around autograd (pddp/utils/evaluation.py:23-288),
same names, argument order and return tuples:

    eval_cost, batch_eval_cost   -> l, l_z, l_u, l_zz, l_uz, l_uu
    eval_dynamics, batch_eval_dynamics -> z_next, F_z, F_u

The `batch_` forms evaluate the plugin once on the replicated input and take
one (cost: two) backward passes - they are what the derivative rollout of the
plugin path runs for every trajectory and time step
(controllers/plugin.py:_cost_derivs / _dyn_derivs, which these call with a
batch of one); the plain forms differentiate row by row and exist as their
cross-check (the reference's tests/utils/test_evaluation.py holds the two to
each other at 1e-3).  The sample problems never come here: their derivatives
are closed-form HIP kernels (csrc/problem_kernels.hip, default_kernels.hip,
bnn_jvp.hip, qr_cost_derivs.hip)."""
import torch

from .autodiff import jacobian
from .encoding import StateEncoding


def _split_cost(l, g, H, n, terminal):
    l_z, l_zz = g[:n], H[:n, :n]
    if terminal:
        return l, l_z, None, l_zz, None, None
    return l, l_z, g[n:], l_zz, H[n:, :n], H[n:, n:]


def eval_cost(cost, z, u, i, terminal=False, encoding=StateEncoding.DEFAULT,
              approximate=False, **kwargs):
    """Row-by-row second derivatives.  `approximate`: outer products of the
    gradient in place of the Hessian blocks (evaluation.py:76-81)."""
    n = z.shape[-1]
    zu = (z if terminal else torch.cat([z, u], -1)).detach().requires_grad_()
    l = cost(zu[:n], None if terminal else zu[n:], i, terminal=terminal,
             encoding=encoding, **kwargs)
    g, = torch.autograd.grad(l, zu, create_graph=not approximate)
    if approximate:
        gd = g.detach()
        H = torch.outer(gd, gd)
    else:
        H = jacobian(g, zu).detach()
    cost.zero_grad()
    return _split_cost(l.detach(), g.detach(), H, n, terminal)


def eval_dynamics(model, z, u, i, encoding=StateEncoding.DEFAULT, **kwargs):
    n = z.shape[-1]
    zu = torch.cat([z, u], -1).detach().requires_grad_()
    z_next = model(zu[:n], zu[n:], i, encoding, **kwargs)
    J = jacobian(z_next, zu).detach()
    model.zero_grad()
    return z_next.detach(), J[:, :n], J[:, n:]


def _problem(model, cost, encoding, model_opts=None, cost_opts=None):
    from ..controllers.plugin import TorchProblem
    return TorchProblem(model, cost, encoding, model_opts or {}, cost_opts or {})


def batch_eval_cost(cost, z, u, i, terminal=False,
                    encoding=StateEncoding.DEFAULT, approximate=False,
                    **kwargs):
    if approximate:
        zu = (z if terminal else torch.cat([z, u], -1)).detach()
        zu.requires_grad_()
        n = z.shape[-1]
        l = cost(zu[:n], None if terminal else zu[n:], i, terminal=terminal,
                 encoding=encoding, **kwargs)
        g, = torch.autograd.grad(l, zu)
        cost.zero_grad()
        return _split_cost(l.detach(), g, torch.outer(g, g), n, terminal)
    out = _problem(None, cost, encoding, cost_opts=kwargs)._cost_derivs(
        z.unsqueeze(0), None if terminal else u.unsqueeze(0), i, terminal)
    cost.zero_grad()
    return tuple(None if t is None else t[0] for t in out)


def batch_eval_dynamics(model, z, u, i, encoding=StateEncoding.DEFAULT,
                        **kwargs):
    with torch.no_grad():
        z_next = model(z, u, i, encoding, **kwargs)
    F_z, F_u = _problem(model, None, encoding, model_opts=kwargs)._dyn_derivs(
        z.unsqueeze(0), u.unsqueeze(0), i)
    model.zero_grad()
    return z_next, F_z[0], F_u[0]
