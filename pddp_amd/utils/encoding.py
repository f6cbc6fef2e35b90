"""State encodings: flat vector z <-> (mean, covariance).

Same public surface as the reference's pddp/utils/encoding.py:25-362
(StateEncoding, infer_*_size, encode, decode_mean / var / std / covar /
covar_sqrt) written independently on batched torch ops; works on any device.
The HIP kernels read and write these layouts directly (csrc/problem_kernels.hip,
default_kernels.hip, bnn_*.hip); the functions here are the host-side helpers of
the reference's API.
"""
from enum import IntEnum

import functools

import numpy as np  # noqa: F401  (the reference's module exposes it to `import *`)
import torch


class StateEncoding(IntEnum):
    """encoding.py:25-43"""
    FULL_COVARIANCE_MATRIX = 0
    UPPER_TRIANGULAR_CHOLESKY = DEFAULT = 1
    VARIANCE_ONLY = 2
    STANDARD_DEVIATION_ONLY = 3
    IGNORE_UNCERTAINTY = 4


def _constant_of(Z, t):
    """The moments an encoding does not carry are constants; like the
    reference (encoding.py:213-214, 255-256, 296-297, 357-358) they are marked
    as requiring a gradient when Z does, so that `autograd.grad(..., Z,
    allow_unused=True)` through them returns None instead of raising."""
    return t.requires_grad_() if Z.requires_grad else t


def infer_encoded_state_size(state_size, encoding=StateEncoding.DEFAULT):
    """encoding.py:46-67"""
    D = state_size
    if encoding == StateEncoding.FULL_COVARIANCE_MATRIX:
        return D + D * D
    if encoding == StateEncoding.UPPER_TRIANGULAR_CHOLESKY:
        return D + D * (D + 1) // 2
    if encoding in (StateEncoding.VARIANCE_ONLY,
                    StateEncoding.STANDARD_DEVIATION_ONLY):
        return 2 * D
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        return D
    raise NotImplementedError("Unknown StateEncoding: {}".format(encoding))


def infer_state_size(encoded_state_size, encoding=StateEncoding.DEFAULT):
    """encoding.py:70-96"""
    n = encoded_state_size
    if encoding == StateEncoding.FULL_COVARIANCE_MATRIX:
        return int(0.5 * (-1 + (1 + 4 * n) ** 0.5))
    if encoding == StateEncoding.UPPER_TRIANGULAR_CHOLESKY:
        return int(0.5 * (-3 + (9 + 8 * n) ** 0.5))
    if encoding in (StateEncoding.VARIANCE_ONLY,
                    StateEncoding.STANDARD_DEVIATION_ONLY):
        return n // 2
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        return n
    raise NotImplementedError("Unknown StateEncoding: {}".format(encoding))


@functools.lru_cache(maxsize=None)
def _triu(D, device):
    """torch.triu_indices(D, D) on `device`, built once (these helpers run
    once or twice per rollout step)."""
    return torch.triu_indices(D, D, device=device)


@functools.lru_cache(maxsize=None)
def _eye(D, dtype, device):
    return torch.eye(D, dtype=dtype, device=device)


def _cholesky_upper(C, jitter=1e-12, max_jitter=10.0):
    """Jittered upper Cholesky, escalating x10 (encoding.py:536-564)."""
    eye = _eye(C.shape[-1], C.dtype, C.device)
    while True:
        L, info = torch.linalg.cholesky_ex(C + jitter * eye, upper=True)
        if not bool((info != 0).any()):
            return L
        jitter *= 10
        if jitter > max_jitter:
            raise RuntimeError("covariance is not positive-definite")


def _covar_from(C, V, S):
    if C is not None:
        return C
    if V is None:
        if S is None:
            raise ValueError("At least one of C, V, S must be specified")
        V = S ** 2
    return torch.diag_embed(V)


def _var_from(C, V, S):
    if V is not None:
        return V
    if S is not None:
        return S ** 2
    if C is not None:
        return torch.diagonal(C, dim1=-2, dim2=-1)
    raise ValueError("At least one of C, V, S must be specified")


def encode(M, C=None, V=None, S=None, encoding=StateEncoding.DEFAULT):
    """encoding.py:99-141"""
    D = M.shape[-1]
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        return M
    if encoding == StateEncoding.FULL_COVARIANCE_MATRIX:
        other = _covar_from(C, V, S).reshape(*M.shape[:-1], D * D)
    elif encoding == StateEncoding.UPPER_TRIANGULAR_CHOLESKY:
        L = _cholesky_upper(_covar_from(C, V, S))
        iu = _triu(D, M.device)
        other = L[..., iu[0], iu[1]]
    elif encoding == StateEncoding.VARIANCE_ONLY:
        other = _var_from(C, V, S)
    elif encoding == StateEncoding.STANDARD_DEVIATION_ONLY:
        other = S if S is not None else _var_from(C, V, S).sqrt()
    else:
        raise NotImplementedError("Unknown StateEncoding: {}".format(encoding))
    return torch.cat([M, other], dim=-1)


def _split(Z, encoding, state_size):
    if state_size is None:
        state_size = infer_state_size(Z.shape[-1], encoding)
    return Z[..., :state_size], Z[..., state_size:], state_size


def _upper_from_flat(X, D):
    L = X.new_zeros(*X.shape[:-1], D, D)
    iu = _triu(D, X.device)
    L[..., iu[0], iu[1]] = X
    return L


def decode_mean(Z, encoding=StateEncoding.DEFAULT, state_size=None):
    """encoding.py:144-156"""
    return _split(Z, encoding, state_size)[0]


def decode_covar(Z, encoding=StateEncoding.DEFAULT, state_size=None):
    """encoding.py:159-218"""
    _, other, D = _split(Z, encoding, state_size)
    if encoding == StateEncoding.FULL_COVARIANCE_MATRIX:
        return other.reshape(*Z.shape[:-1], D, D)
    if encoding == StateEncoding.UPPER_TRIANGULAR_CHOLESKY:
        L = _upper_from_flat(other, D)
        return L.transpose(-1, -2) @ L
    if encoding == StateEncoding.VARIANCE_ONLY:
        return torch.diag_embed(other)
    if encoding == StateEncoding.STANDARD_DEVIATION_ONLY:
        return torch.diag_embed(other ** 2)
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        eye = 1e-6 * torch.eye(D, dtype=Z.dtype, device=Z.device)
        return _constant_of(Z, eye.expand(*Z.shape[:-1], D, D))
    raise NotImplementedError("Unknown StateEncoding: {}".format(encoding))


def decode_var(Z, encoding=StateEncoding.DEFAULT, state_size=None):
    """encoding.py:221-260"""
    _, other, D = _split(Z, encoding, state_size)
    if encoding == StateEncoding.FULL_COVARIANCE_MATRIX:
        return other[..., ::D + 1]
    if encoding == StateEncoding.UPPER_TRIANGULAR_CHOLESKY:
        return (_upper_from_flat(other, D) ** 2).sum(dim=-2)
    if encoding == StateEncoding.VARIANCE_ONLY:
        return other
    if encoding == StateEncoding.STANDARD_DEVIATION_ONLY:
        return other ** 2
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        return _constant_of(Z, (1e-6 * torch.ones(
            D, dtype=Z.dtype, device=Z.device)).expand(*Z.shape[:-1], D))
    raise NotImplementedError("Unknown StateEncoding: {}".format(encoding))


def decode_std(Z, encoding=StateEncoding.DEFAULT, state_size=None):
    """encoding.py:263-301"""
    if encoding == StateEncoding.STANDARD_DEVIATION_ONLY:
        return _split(Z, encoding, state_size)[1]
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        D = _split(Z, encoding, state_size)[2]
        return _constant_of(Z, (1e-3 * torch.ones(
            D, dtype=Z.dtype, device=Z.device)).expand(*Z.shape[:-1], D))
    return decode_var(Z, encoding, state_size).sqrt()


def decode_covar_sqrt(Z, encoding=StateEncoding.DEFAULT, state_size=None):
    """encoding.py:304-362 (upper factor L with L^T L = covariance)."""
    _, other, D = _split(Z, encoding, state_size)
    if encoding == StateEncoding.FULL_COVARIANCE_MATRIX:
        return _cholesky_upper(other.reshape(*Z.shape[:-1], D, D))
    if encoding == StateEncoding.UPPER_TRIANGULAR_CHOLESKY:
        return _upper_from_flat(other, D)
    if encoding == StateEncoding.VARIANCE_ONLY:
        return torch.diag_embed(other.sqrt())
    if encoding == StateEncoding.STANDARD_DEVIATION_ONLY:
        return torch.diag_embed(other)
    if encoding == StateEncoding.IGNORE_UNCERTAINTY:
        eye = 1e-3 * torch.eye(D, dtype=Z.dtype, device=Z.device)
        return _constant_of(Z, eye.expand(*Z.shape[:-1], D, D))
    raise NotImplementedError("Unknown StateEncoding: {}".format(encoding))
