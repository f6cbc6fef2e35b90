"""Cost plugin interface (reference: pddp/costs/base.py:22-181)."""
import torch

from ..utils.encoding import StateEncoding


class Cost(torch.nn.Module):
    """Base cost: `forward(z, u, i, terminal, encoding, identical_inputs)`
    returns the expected cost; arithmetic operators build an AggregateCost
    (base.py:25-97)."""

    def __add__(self, other):
        return AggregateCost(self, other, torch.add)

    def __sub__(self, other):
        return AggregateCost(self, other, torch.sub)

    def __mul__(self, other):
        return AggregateCost(self, other, torch.mul)

    def __truediv__(self, other):
        return AggregateCost(self, other, torch.div)

    __div__ = __truediv__

    def __pow__(self, other):
        return AggregateCost(self, other, torch.pow)

    def __neg__(self):
        return AggregateCost(self, -1, torch.mul)

    def forward(self, z, u, i, terminal=False, encoding=StateEncoding.DEFAULT,
                identical_inputs=False, **kwargs):
        raise NotImplementedError


class AggregateCost(Cost):
    """op(first, second) of two costs or a cost and a constant
    (base.py:125-181)."""

    def __init__(self, first, second, op):
        super(AggregateCost, self).__init__()
        self.first, self.second, self.op = first, second, op

    def forward(self, z, u, i, terminal=False, encoding=StateEncoding.DEFAULT,
                **kwargs):
        ev = lambda c: (c(z, u, i, terminal, encoding, **kwargs)
                        if isinstance(c, Cost) else c)
        return self.op(ev(self.first), ev(self.second))
