"""Particle helpers (reference: pddp/utils/particles.py:136-149;
`particulate_model` there is untested and broken - SURVEY.md row 11 - and is
not provided)."""
import torch


def particles_covar(x):
    """Unbiased covariance over the FIRST axis (the particles): x
    [P, D] -> [D, D], or [P, B, D] -> [B, D, D] (the layout
    BNNDynamicsModel.forward holds its particle clouds in,
    modules.py:366-377)."""
    P = x.shape[0]
    dev = x - x.mean(dim=0)
    if dev.dim() == 3:
        return torch.einsum("pbi,pbj->bij", dev, dev) / (P - 1)
    return dev.t() @ dev / (P - 1)
