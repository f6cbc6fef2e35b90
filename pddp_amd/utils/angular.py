"""Angle -> (sin, cos) augmentation of state means (reference:
pddp/utils/angular.py:251-354).  Output order
[non_angular..., sin a1, cos a1, sin a2, cos a2, ...]."""
import torch


def augment_state(x, angular_indices, non_angular_indices):
    """angular.py:251-286"""
    if len(angular_indices) == 0:
        return x
    ang = x[..., angular_indices]
    sc = torch.stack([ang.sin(), ang.cos()], dim=-1).reshape(
        *x.shape[:-1], 2 * len(angular_indices))
    return torch.cat([x[..., non_angular_indices], sc], dim=-1)


def reduce_state(x_, angular_indices, non_angular_indices):
    """angular.py:289-326"""
    n_ang, n_non = len(angular_indices), len(non_angular_indices)
    if n_ang == 0:
        return x_
    sc = x_[..., n_non:]
    angles = torch.atan2(sc[..., ::2], sc[..., 1::2])
    if n_non == 0:
        return angles
    x = x_.new_empty(*x_.shape[:-1], n_ang + n_non)
    x[..., angular_indices] = angles
    x[..., non_angular_indices] = x_[..., :n_non]
    return x


def infer_augmented_state_size(angular_indices, non_angular_indices):
    """angular.py:329-340"""
    return len(non_angular_indices) + 2 * len(angular_indices)


def infer_reduced_state_size(angular_indices, non_angular_indices):
    """angular.py:343-354"""
    return len(non_angular_indices) + len(angular_indices)
