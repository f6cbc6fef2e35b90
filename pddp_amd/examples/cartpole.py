"""Cartpole swing-up (reference: pddp/examples/cartpole/{model,cost,env}.py).

state [x, x', theta, theta'], action [F]; theta = 0 is up."""
import numpy as np
import torch
from torch.nn import Parameter

from ._common import AugmentedQRCost, build_problem
from ..envs.base import ModelEnv
from ..models.base import DynamicsModel
from ..utils.angular import augment_state, infer_augmented_state_size
from ..utils.classproperty import classproperty
from ..utils.encoding import StateEncoding, decode_mean, decode_var, encode


class CartpoleDynamicsModel(DynamicsModel):
    """cartpole/model.py:27-141"""

    def __init__(self, dt, mc=0.5, mp=0.5, l=0.5, mu=0.1, g=9.82):
        super(CartpoleDynamicsModel, self).__init__()
        self.dt = Parameter(torch.tensor(dt), requires_grad=False)
        self.mc = Parameter(torch.tensor(mc))
        self.mp = Parameter(torch.tensor(mp))
        self.l = Parameter(torch.tensor(l))
        self.mu = Parameter(torch.tensor(mu))
        self.g = Parameter(torch.tensor(g))

    @classproperty
    def action_size(cls):
        return 1

    @classproperty
    def state_size(cls):
        return 4

    @classproperty
    def angular_indices(cls):
        return torch.tensor([2]).long()

    @classproperty
    def non_angular_indices(cls):
        return torch.tensor([0, 1, 3]).long()

    def fit(self, X, U, dX, quiet=False, **kwargs):
        pass  # exact model

    def forward(self, z, u, i, encoding=StateEncoding.DEFAULT, **kwargs):
        dt, mc, mp, l, mu, g = (self.dt, self.mc, self.mp, self.l, self.mu,
                                self.g)
        mean = decode_mean(z, encoding)
        var = decode_var(z, encoding)
        x, xd, th, thd = mean.unbind(-1)
        F = u[..., 0]
        s, c = th.sin(), th.cos()
        a0 = mp * l * thd ** 2 * s
        a1 = g * s
        a2 = F - mu * xd
        a3 = 4 * (mc + mp) - 3 * mp * c ** 2
        thdd = -3 * (a0 * c + 2 * ((mc + mp) * a1 + a2 * c)) / (l * a3)
        xdd = (2 * a0 + 3 * mp * a1 * c + 4 * a2) / a3
        nxd = xd + xdd * dt
        nthd = thd + thdd * dt
        mean = torch.stack([x + nxd * dt, nxd, th + nthd * dt, nthd], dim=-1)
        return encode(mean, V=var, encoding=encoding)

    def native_problem(self, encoding, cost=None):
        return build_problem("cartpole", self, cost, encoding,
                             ["dt", "mc", "mp", "l", "mu", "g"])


class CartpoleCost(AugmentedQRCost):
    """cartpole/cost.py:28-58: distance of the pole tip to the goal on the
    augmented state [x, x', theta', sin theta, cos theta]."""

    model_class = CartpoleDynamicsModel

    def __init__(self, pole_length=0.5):
        model = CartpoleDynamicsModel
        na = infer_augmented_state_size(model.angular_indices,
                                        model.non_angular_indices)
        Q = torch.zeros(na, na)
        Q_term = torch.eye(na)
        Q[0, 0] = 1.0
        Q[0, 3] = Q[3, 0] = pole_length
        Q[3, 3] = Q[4, 4] = pole_length ** 2
        R = 0.1 * torch.eye(model.action_size)
        x_goal = augment_state(torch.tensor([0.0, 0.0, np.pi, 0.0]),
                               model.angular_indices,
                               model.non_angular_indices)
        super(CartpoleCost, self).__init__(Q, R, Q_term=Q_term, x_goal=x_goal)


class CartpoleEnv(ModelEnv):
    """cartpole/env.py:31-118 without gym / rendering."""

    def __init__(self, model=None, dt=0.1, render=False):
        self.dt = dt
        if model is None:
            model = CartpoleDynamicsModel(dt)
        super(CartpoleEnv, self).__init__(model, np.zeros(4))
