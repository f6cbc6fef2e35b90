"""Angle -> (sin, cos) augmentation of state means (reference:
pddp/utils/angular.py:251-354).  Output order
[non_angular..., sin a1, cos a1, sin a2, cos a2, ...]."""
import functools

import torch

from .encoding import (StateEncoding, decode_covar, decode_mean, decode_var,
                       encode)


@functools.lru_cache(maxsize=None)
def _index_tensor(indices, device):
    return torch.tensor(indices, dtype=torch.long, device=device)


def _idx(indices, like):
    """The index list as a LongTensor on `like`'s device, built once: indexing
    a CUDA tensor with a python list or a CPU tensor costs a host-to-device
    copy per call, and these run once per rollout step."""
    if torch.is_tensor(indices):
        indices = indices.tolist()
    return _index_tensor(tuple(int(i) for i in indices), like.device)


def _take(x, indices, dim=-1):
    return x.index_select(dim, _idx(indices, x))


def complementary_indices(indices, size):
    """The indices of range(size) that are not in `indices`, ascending, as a
    long tensor (reference: angular.py:26-43)."""
    idx = torch.as_tensor(indices, dtype=torch.long).reshape(-1)
    keep = torch.ones(size, dtype=torch.bool, device=idx.device)
    keep[idx] = False
    return keep.nonzero()[:, 0]


def augment_state(x, angular_indices, non_angular_indices):
    """angular.py:251-286"""
    if len(angular_indices) == 0:
        return x
    ang = _take(x, angular_indices)
    sc = torch.stack([ang.sin(), ang.cos()], dim=-1).reshape(
        *x.shape[:-1], 2 * len(angular_indices))
    return torch.cat([_take(x, non_angular_indices), sc], dim=-1)


def _angle_moments(mi, vi):
    """E[sin], E[cos] of Gaussian angles: exp(-v/2) sin(mu), exp(-v/2) cos(mu),
    interleaved as [sin a1, cos a1, sin a2, ...]."""
    damp = torch.exp(-0.5 * vi)
    return torch.stack([damp * mi.sin(), damp * mi.cos()], dim=-1).reshape(
        *mi.shape[:-1], 2 * mi.shape[-1])


def augment_moments(mean, covar, angular_indices, non_angular_indices):
    """Moment-matched augmentation of a Gaussian state: mean and FULL
    covariance of [x_na, sin a1, cos a1, ...] (reference: angular.py:161-248
    `_augment_covar`, after kusanagi).  For jointly Gaussian angles with
    covariance c_ij, with q = exp(-(v_i + v_j)/2):

        Cov(sin_i, sin_j) = 0.5 [q (e^c - 1) cos(m_i - m_j) - q (e^-c - 1) cos(m_i + m_j)]
        Cov(cos_i, cos_j) = 0.5 [q (e^c - 1) cos(m_i - m_j) + q (e^-c - 1) cos(m_i + m_j)]
        Cov(sin_i, cos_j) = 0.5 [q (e^c - 1) sin(m_i - m_j) + q (e^-c - 1) sin(m_i + m_j)]
        Cov(x, sin_i) = C[i, :] E[cos_i],  Cov(x, cos_i) = -C[i, :] E[sin_i]
    """
    ai, ni = list(angular_indices), list(non_angular_indices)
    na_, nn = len(ai), len(ni)
    if na_ == 0:
        return mean, covar
    mi = _take(mean, ai)
    ci = _take(_take(covar, ai, -2), ai, -1)
    vi = torch.diagonal(ci, dim1=-2, dim2=-1)
    Ma = _angle_moments(mi, vi)
    lq = -0.5 * (vi.unsqueeze(-1) + vi.unsqueeze(-2))
    q = lq.exp()
    ep = (lq + ci).exp() - q       # q (e^c - 1)
    em = (lq - ci).exp() - q       # q (e^-c - 1)
    dm = mi.unsqueeze(-1) - mi.unsqueeze(-2)
    sm = mi.unsqueeze(-1) + mi.unsqueeze(-2)
    ss = 0.5 * (ep * dm.cos() - em * sm.cos())
    cc = 0.5 * (ep * dm.cos() + em * sm.cos())
    sc = 0.5 * (ep * dm.sin() + em * sm.sin())
    Va = mean.new_zeros(*mean.shape[:-1], 2 * na_, 2 * na_)
    Va[..., 0::2, 0::2] = ss
    Va[..., 1::2, 1::2] = cc
    Va[..., 0::2, 1::2] = sc
    Va[..., 1::2, 0::2] = sc.transpose(-1, -2)
    M = torch.cat([_take(mean, ni), Ma], dim=-1)
    C = mean.new_zeros(*mean.shape[:-1], nn + 2 * na_, nn + 2 * na_)
    C[..., nn:, nn:] = Va
    if nn > 0:
        rows_ni = _take(covar, ni, -2)
        C[..., :nn, :nn] = _take(rows_ni, ni, -1)
        # C[angle_i, x] as the reference reads it (angular.py:243-245 sums
        # over the ROW index: with the full-covariance encoding the two
        # triangles are separate inputs and the partials must land where the
        # reference's do)
        cols = _take(_take(covar, ai, -2), ni, -1).transpose(-1, -2)
        cross = mean.new_zeros(*mean.shape[:-1], nn, 2 * na_)
        cross[..., 0::2] = cols * Ma[..., 1::2].unsqueeze(-2)    # x, sin
        cross[..., 1::2] = -cols * Ma[..., 0::2].unsqueeze(-2)   # x, cos
        C[..., :nn, nn:] = cross
        C[..., nn:, :nn] = cross.transpose(-1, -2)
    return M, C


def augment_moments_var(mean, var, angular_indices, non_angular_indices):
    """Diagonal version (reference: angular.py:87-158 `_augment_var`)."""
    ai, ni = list(angular_indices), list(non_angular_indices)
    if len(ai) == 0:
        return mean, var
    mi, vi = _take(mean, ai), _take(var, ai)
    Ma = _angle_moments(mi, vi)
    q = (-vi).exp()
    u3 = (1.0 - q)                       # q (e^v - 1) cos(0)
    u4 = ((-2.0 * vi).exp() - q) * (2.0 * mi).cos()
    Va = torch.stack([0.5 * (u3 - u4), 0.5 * (u3 + u4)], dim=-1).reshape(
        *mi.shape[:-1], 2 * len(ai))
    return (torch.cat([_take(mean, ni), Ma], dim=-1),
            torch.cat([_take(var, ni), Va], dim=-1))


def augment_encoded_state(z, angular_indices, non_angular_indices,
                          encoding=StateEncoding.DEFAULT, state_size=None):
    """Encoded state -> encoded augmented state (reference:
    angular.py:47-84)."""
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        return augment_state(z, angular_indices, non_angular_indices)
    mean = decode_mean(z, encoding, state_size)
    if encoding in (StateEncoding.FULL_COVARIANCE_MATRIX,
                    StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
        M, C = augment_moments(mean, decode_covar(z, encoding, state_size),
                               angular_indices, non_angular_indices)
        return encode(M, C=C, encoding=encoding)
    M, V = augment_moments_var(mean, decode_var(z, encoding, state_size),
                               angular_indices, non_angular_indices)
    return encode(M, V=V, encoding=encoding)


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
    x.index_copy_(-1, _idx(angular_indices, x), angles)
    x.index_copy_(-1, _idx(non_angular_indices, x), x_[..., :n_non])
    return x


def infer_augmented_state_size(angular_indices, non_angular_indices):
    """angular.py:329-340"""
    return len(non_angular_indices) + 2 * len(angular_indices)


def infer_reduced_state_size(angular_indices, non_angular_indices):
    """angular.py:343-354"""
    return len(non_angular_indices) + len(angular_indices)
