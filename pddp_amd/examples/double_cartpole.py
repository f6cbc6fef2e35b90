"""Double cartpole (reference:
pddp/examples/double_cartpole/{model,cost,env}.py).

state [x, x', theta1, theta1', theta2, theta2'], action [F]."""
import numpy as np
import torch
from torch.nn import Parameter

from ._common import AugmentedQRCost, build_problem
from ..envs.base import ModelEnv
from ..models.base import DynamicsModel
from ..utils.angular import augment_state, infer_augmented_state_size
from ..utils.classproperty import classproperty
from ..utils.encoding import StateEncoding, decode_mean, decode_var, encode


class DoubleCartpoleDynamicsModel(DynamicsModel):
    """double_cartpole/model.py:27-195"""

    def __init__(self, dt, mc=0.5, mp1=0.5, mp2=0.5, l1=0.6, l2=0.6, mu=0.1,
                 g=9.80665):
        super(DoubleCartpoleDynamicsModel, self).__init__()
        self.dt = Parameter(torch.tensor(dt), requires_grad=False)
        for name, v in (("mc", mc), ("mp1", mp1), ("mp2", mp2), ("l1", l1),
                        ("l2", l2), ("mu", mu), ("g", g)):
            setattr(self, name, Parameter(torch.tensor(v)))

    @classproperty
    def action_size(cls):
        return 1

    @classproperty
    def state_size(cls):
        return 6

    @classproperty
    def angular_indices(cls):
        return torch.tensor([2, 4]).long()

    @classproperty
    def non_angular_indices(cls):
        return torch.tensor([0, 1, 3, 5]).long()

    def fit(self, X, U, dX, quiet=False, **kwargs):
        pass

    def forward(self, z, u, i, encoding=StateEncoding.DEFAULT, **kwargs):
        dt, mc, mp1, mp2, l1, l2, mu, g = (self.dt, self.mc, self.mp1,
                                           self.mp2, self.l1, self.l2,
                                           self.mu, self.g)
        mean = decode_mean(z, encoding)
        var = decode_var(z, encoding)
        x, xd, t1, t1d, t2, t2d = mean.unbind(-1)
        F = u[..., 0]
        s1, c1, s2, c2 = t1.sin(), t1.cos(), t2.sin(), t2.cos()
        sd, cd = (t1 - t2).sin(), (t1 - t2).cos()
        a0 = mp2 + 2 * mc
        a1 = mc * l2
        a2 = l1 * t1d ** 2
        a3 = a1 * t2d ** 2
        one = torch.ones_like(x)
        A = torch.stack([
            torch.stack([2 * (mp1 + mp2 + mc) * one, -a0 * l1 * c1,
                         -a1 * c2], dim=-1),
            torch.stack([-3 * a0 * c1, (2 * a0 + 2 * mc) * l1 * one,
                         3 * a1 * cd], dim=-1),
            torch.stack([-3 * c2, 3 * l1 * cd, 2 * l2 * one], dim=-1),
        ], dim=-2)
        b = torch.stack([
            2 * F - 2 * mu * xd - a0 * a2 * s1 - a3 * s2,
            3 * a0 * g * s1 - 3 * a3 * sd,
            3 * a2 * sd + 3 * g * s2,
        ], dim=-1).unsqueeze(-1)
        sol = torch.linalg.solve(A, b).squeeze(-1)
        nxd = xd + sol[..., 0] * dt
        nt1d = t1d + sol[..., 1] * dt
        nt2d = t2d + sol[..., 2] * dt
        mean = torch.stack([x + nxd * dt, nxd, t1 + nt1d * dt, nt1d,
                            t2 + nt2d * dt, nt2d], dim=-1)
        return encode(mean, V=var, encoding=encoding)

    def native_problem(self, encoding, cost=None):
        return build_problem("double_cartpole", self, cost, encoding,
                             ["dt", "mc", "mp1", "mp2", "l1", "l2", "mu", "g"])


class DoubleCartpoleCost(AugmentedQRCost):
    """double_cartpole/cost.py:29-66 on [x, x', th1', th2', sin th1, cos th1,
    sin th2, cos th2]."""

    model_class = DoubleCartpoleDynamicsModel

    def __init__(self, pole1_length=0.6, pole2_length=0.6):
        model = DoubleCartpoleDynamicsModel
        na = infer_augmented_state_size(model.angular_indices,
                                        model.non_angular_indices)
        Q_term = 100 * torch.eye(na)
        Q = torch.zeros(na, na)
        dims = torch.tensor([0, 4, 5, 6, 7])
        C = torch.tensor([[1, -pole1_length, 0, -pole2_length, 0],
                          [0, 0, pole1_length, 0, pole2_length]])
        Q[dims.unsqueeze(1), dims.unsqueeze(0)] = C.t().mm(C)
        R = 0.1 * torch.eye(model.action_size)
        x_goal = augment_state(torch.zeros(model.state_size),
                               model.angular_indices,
                               model.non_angular_indices)
        super(DoubleCartpoleCost, self).__init__(Q, R, Q_term=Q_term,
                                                 x_goal=x_goal)


class DoubleCartpoleEnv(ModelEnv):
    """double_cartpole/env.py: starts hanging down."""

    def __init__(self, model=None, dt=0.05, render=False):
        self.dt = dt
        if model is None:
            model = DoubleCartpoleDynamicsModel(dt)
        super(DoubleCartpoleEnv, self).__init__(
            model, np.array([0.0, 0.0, np.pi, 0.0, np.pi, 0.0]))
