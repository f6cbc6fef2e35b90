"""Trajectory conveniences of the example scripts (reference:
pddp/utils/trajectory.py:20-73; not used by the controllers)."""
import torch


def _stack(X, pick):
    if len(X) == 0:
        raise ValueError("Trajectory cannot be empty")
    return torch.stack([pick(x) for x in X])


def mean_trajectory(X):
    """List of N GaussianVariable -> means [N, state_size]."""
    return _stack(X, lambda x: x.mean())


def sample_trajectory(X):
    """List of N GaussianVariable -> one sample of each [N, state_size]."""
    return _stack(X, lambda x: x.sample())


def trajectory_to_training_data(X, U):
    """States [N + 1, D] and actions [N, m] -> (inputs [N, D + m], targets
    [N, D]); the target keeps the reference's sign, x_t - x_{t+1}
    (trajectory.py:71)."""
    return torch.cat([X[:-1], U], dim=-1), X[:-1] - X[1:]
