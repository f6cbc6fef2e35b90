"""Two-vehicle rendezvous (reference:
pddp/examples/rendezvous/{model,cost,env}.py).

state [x0, y0, x1, y1, x0', y0', x1', y1'], action [Fx0, Fy0, Fx1, Fy1]."""
import numpy as np
import torch
from torch.nn import Parameter

from ._common import AugmentedQRCost, build_problem
from ..envs.base import ModelEnv
from ..models.base import DynamicsModel
from ..utils.classproperty import classproperty
from ..utils.encoding import StateEncoding, decode_covar, decode_mean, encode


class RendezvousDynamicsModel(DynamicsModel):
    """rendezvous/model.py:27-115"""

    def __init__(self, dt, m=1.0, alpha=0.1):
        super(RendezvousDynamicsModel, self).__init__()
        self.dt = Parameter(torch.tensor(dt), requires_grad=False)
        self.m = Parameter(torch.tensor(m))
        self.alpha = Parameter(torch.tensor(alpha))

    @classproperty
    def action_size(cls):
        return 4

    @classproperty
    def state_size(cls):
        return 8

    @classproperty
    def angular_indices(cls):
        return torch.tensor([]).long()

    @classproperty
    def non_angular_indices(cls):
        return torch.arange(8).long()

    def fit(self, X, U, dX, quiet=False, **kwargs):
        pass

    def forward(self, z, u, i, encoding=StateEncoding.DEFAULT, **kwargs):
        dt = self.dt
        x = decode_mean(z, encoding)
        pos, vel = x[..., :4], x[..., 4:]
        acc = vel * (1 - self.alpha * dt / self.m)
        acc = acc + u * dt / self.m
        mean = torch.cat([pos + vel * dt, vel + acc * dt], dim=-1)
        if encoding == StateEncoding.IGNORE_UNCERTAINTY:
            return mean
        return encode(mean, C=decode_covar(z, encoding), encoding=encoding)

    def native_problem(self, encoding, cost=None):
        return build_problem("rendezvous", self, cost, encoding,
                             ["dt", "m", "alpha"])


class RendezvousCost(AugmentedQRCost):
    """rendezvous/cost.py:26-43"""

    model_class = RendezvousDynamicsModel

    def __init__(self):
        model = RendezvousDynamicsModel
        Q = torch.eye(model.state_size)
        Q[0, 2] = Q[2, 0] = -1
        Q[1, 3] = Q[3, 1] = -1
        R = 0.1 * torch.eye(model.action_size)
        super(RendezvousCost, self).__init__(Q, R)


class RendezvousEnv(ModelEnv):

    def __init__(self, model=None, dt=0.1, render=False):
        self.dt = dt
        if model is None:
            model = RendezvousDynamicsModel(dt)
        super(RendezvousEnv, self).__init__(
            model,
            np.array([-10.0, -10.0, 10.0, 10.0, 0.0, -5.0, 5.0, 0.0]))
