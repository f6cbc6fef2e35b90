"""Quadratic cost (reference: pddp/costs/quadratic.py:23-99)."""
import torch

from .base import Cost
from ..utils.encoding import StateEncoding, decode_covar, decode_mean


class QRCost(Cost):
    r"""E[L] = tr(Q \Sigma) + (\mu - x_goal)^T Q (\mu - x_goal)
             + (u - u_goal)^T R (u - u_goal);  the terminal cost uses Q_term
    and drops the action term."""

    def __init__(self, Q, R, Q_term=None, x_goal=0.0, u_goal=0.0):
        super(QRCost, self).__init__()
        Q_term = Q if Q_term is None else Q_term
        P = lambda t: torch.nn.Parameter(torch.as_tensor(t).clone(),
                                         requires_grad=False)
        self.Q, self.R, self.Q_term = P(Q), P(R), P(Q_term)
        self.x_goal, self.u_goal = P(x_goal), P(u_goal)

    def forward(self, z, u, i, terminal=False, encoding=StateEncoding.DEFAULT,
                **kwargs):
        Q = self.Q_term if terminal else self.Q
        dx = decode_mean(z, encoding) - self.x_goal
        cost = ((dx @ Q) * dx).sum(-1)
        if not terminal:
            du = u - self.u_goal
            cost = cost + ((du @ self.R) * du).sum(-1)
        if encoding != StateEncoding.IGNORE_UNCERTAINTY:
            C = decode_covar(z, encoding)
            cost = cost + (C * Q.t()).sum((-2, -1))  # tr(Q Sigma)
        return cost
