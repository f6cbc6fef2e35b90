"""Dynamics model plugin interface (reference: pddp/models/base.py:24-83)."""
import torch

from ..utils.classproperty import classproperty
from ..utils.encoding import StateEncoding


class DynamicsModel(torch.nn.Module):
    """Base dynamics model.  Same contract as the reference: `forward(z, u, i,
    encoding, identical_inputs=False, **kw)` maps an encoded state distribution
    and an action to the next encoded state distribution; `fit(X, U, dX)`
    trains it.

    MI355X extension: a model may return a `pddp_problem` description from
    `native_problem(encoding)`; the controllers then run the whole iteration in
    HIP kernels (pddp_amd/csrc).  Models that return None are plugin models.
    """

    def reset_parameters(self, initializer=torch.nn.init.normal_):
        for p in self.parameters():
            if p.requires_grad:
                initializer(p)
        return self

    @classproperty
    def action_size(cls):
        raise NotImplementedError

    @classproperty
    def state_size(cls):
        raise NotImplementedError

    def fit(self, X, U, dX, quiet=False, **kwargs):
        raise NotImplementedError

    def forward(self, z, u, i, encoding=StateEncoding.DEFAULT,
                identical_inputs=False, **kwargs):
        raise NotImplementedError

    def native_problem(self, encoding, cost=None):
        """ctypes `PddpProblem` for the HIP path, or None."""
        return None
