"""Pendulum swing-up (reference: pddp/examples/pendulum/{model,cost,env}.py).

state [theta, theta'], action [torque]; theta = 0 is up."""
import numpy as np
import torch
from torch.nn import Parameter

from ._common import AugmentedQRCost, build_problem
from ..envs.base import ModelEnv
from ..models.base import DynamicsModel
from ..utils.angular import augment_state, infer_augmented_state_size
from ..utils.classproperty import classproperty
from ..utils.encoding import StateEncoding, decode_mean, decode_var, encode


class PendulumDynamicsModel(DynamicsModel):
    """pendulum/model.py:27-119"""

    def __init__(self, dt, m=1.0, l=1.0, mu=0.1, g=9.80665):
        super(PendulumDynamicsModel, self).__init__()
        self.dt = Parameter(torch.tensor(dt), requires_grad=False)
        self.m = Parameter(torch.tensor(m))
        self.l = Parameter(torch.tensor(l))
        self.mu = Parameter(torch.tensor(mu))
        self.g = Parameter(torch.tensor(g))

    @classproperty
    def action_size(cls):
        return 1

    @classproperty
    def state_size(cls):
        return 2

    @classproperty
    def angular_indices(cls):
        return torch.tensor([0]).long()

    @classproperty
    def non_angular_indices(cls):
        return torch.tensor([1]).long()

    def fit(self, X, U, dX, quiet=False, **kwargs):
        pass

    def forward(self, z, u, i, encoding=StateEncoding.DEFAULT, **kwargs):
        dt, m, l, mu, g = self.dt, self.m, self.l, self.mu, self.g
        mean = decode_mean(z, encoding)
        var = decode_var(z, encoding)
        th, thd = mean.unbind(-1)
        temp = m * l
        acc = u[..., 0] - mu * thd - 0.5 * temp * g * th.sin()
        acc = 3 * acc / (temp * l)
        mean = torch.stack([th + thd * dt, thd + acc * dt], dim=-1)
        return encode(mean, V=var, encoding=encoding)

    def native_problem(self, encoding, cost=None):
        return build_problem("pendulum", self, cost, encoding,
                             ["dt", "m", "l", "mu", "g"])


class PendulumCost(AugmentedQRCost):
    """pendulum/cost.py:29-60 on [theta', sin theta, cos theta]."""

    model_class = PendulumDynamicsModel

    def __init__(self, pendulum_length=0.5):
        model = PendulumDynamicsModel
        na = infer_augmented_state_size(model.angular_indices,
                                        model.non_angular_indices)
        Q = torch.zeros(na, na)
        Q[0, 0] = 1.0
        Q[0, 1] = Q[1, 0] = pendulum_length
        Q[1, 1] = Q[2, 2] = pendulum_length ** 2
        Q_term = 100 * torch.eye(na)
        R = 0.1 * torch.eye(model.action_size)
        x_goal = augment_state(torch.tensor([np.pi, 0.0]),
                               model.angular_indices,
                               model.non_angular_indices)
        super(PendulumCost, self).__init__(Q, R, Q_term=Q_term, x_goal=x_goal)


class PendulumEnv(ModelEnv):

    def __init__(self, model=None, dt=0.1, render=False):
        self.dt = dt
        if model is None:
            model = PendulumDynamicsModel(dt)
        super(PendulumEnv, self).__init__(model, np.zeros(2))
