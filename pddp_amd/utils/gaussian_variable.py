"""Gaussian state distribution holder (reference:
pddp/utils/gaussian_variable.py:22-275; same constructor and accessors)."""
import torch

from .encoding import (StateEncoding, decode_covar, decode_mean, decode_std,
                       decode_var, encode)


class GaussianVariable(object):

    def __init__(self, mean, covar=None, var=None, std=None):
        if covar is None and var is None and std is None:
            raise ValueError("one of covar, var, std is required")
        self._mean, self._covar, self._var, self._std = mean, covar, var, std

    def __repr__(self):
        return "GaussianVariable({})".format(tuple(self.shape))

    @property
    def device(self):
        return self._mean.device

    @property
    def dtype(self):
        return self._mean.dtype

    @property
    def shape(self):
        return self._mean.shape

    def mean(self):
        return self._mean

    def var(self):
        if self._var is None:
            self._var = (torch.diagonal(self._covar, dim1=-2, dim2=-1)
                         if self._covar is not None else self._std ** 2)
        return self._var

    def std(self):
        if self._std is None:
            self._std = self.var().sqrt()
        return self._std

    def covar(self):
        if self._covar is None:
            self._covar = torch.diag_embed(self.var())
        return self._covar

    def sample(self, sample_shape=torch.Size([])):
        if self._covar is not None:
            dist = torch.distributions.MultivariateNormal(self._mean,
                                                          self._covar)
        else:
            dist = torch.distributions.Normal(self._mean, self.std())
        return dist.sample(sample_shape)

    def encode(self, encoding=StateEncoding.DEFAULT):
        """gaussian_variable.py: encode() -> flat z."""
        if self._covar is not None:
            return encode(self._mean, C=self._covar, encoding=encoding)
        return encode(self._mean, V=self.var(), encoding=encoding)

    @classmethod
    def decode(cls, z, encoding=StateEncoding.DEFAULT, state_size=None):
        mean = decode_mean(z, encoding, state_size)
        if encoding in (StateEncoding.FULL_COVARIANCE_MATRIX,
                        StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
            return cls(mean, covar=decode_covar(z, encoding, state_size))
        if encoding == StateEncoding.STANDARD_DEVIATION_ONLY:
            return cls(mean, std=decode_std(z, encoding, state_size))
        return cls(mean, var=decode_var(z, encoding, state_size))

    def to(self, *args, **kwargs):
        f = lambda t: None if t is None else t.to(*args, **kwargs)
        return GaussianVariable(f(self._mean), f(self._covar), f(self._var),
                                f(self._std))

    @classmethod
    def random(cls, n, reg=1e-1, requires_grad=True, **tensor_opts):
        """A random valid variable of size n: covariance A^T A + reg I from leaf
        tensors that require a gradient by default (gaussian_variable.py:258-275;
        the reference's tests differentiate through it)."""
        mean = torch.randn(n, requires_grad=requires_grad, **tensor_opts)
        A = torch.randn(n, n, requires_grad=requires_grad, **tensor_opts)
        eye = torch.eye(n, dtype=mean.dtype, device=mean.device)
        return cls(mean, covar=A.t().mm(A) + reg * eye)
