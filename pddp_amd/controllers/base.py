"""Base controller (reference: pddp/controllers/base.py:22-71)."""
import torch


class Controller(torch.nn.Module):

    def fit(self, U, encoding=None, **kwargs):
        """Determines the optimal path to minimise the cost."""
        raise NotImplementedError

    def forward(self, z, i, encoding=None, **kwargs):
        """Determines the optimal single-step control."""
        raise NotImplementedError
