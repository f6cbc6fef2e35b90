"""In-memory shims that let the read-only reference at /root/reference import
under Python 3.10 / torch 2.x in THIS container (never on the GPU box).

Used only by tools/make_golden.py to capture golden vectors.  Nothing here is
shipped or imported by the product (`pddp_amd`), the tests, or bench.py.

The shims follow SURVEY.md section 8(c):
  1. `gym` is not installed -> stub modules gym / gym.spaces / gym.utils.seeding;
  2. `collections.Iterable` moved to `collections.abc`;
  3. torch 0.4.1 LAPACK names (potrf / potrs / gesv / trtrs / eig) mapped to
     their torch.linalg equivalents; uint8 mask indexing converted to bool.
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np
import torch

REFERENCE_PATH = os.environ.get("PDDP_REFERENCE_PATH", "/root/reference")


def _install_gym_stub():
    if "gym" in sys.modules:
        return
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")

    class Env(object):
        metadata = {}

        def close(self):
            pass

    class Box(object):
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape if shape is None else shape
            self.dtype = np.dtype(dtype)

        def sample(self):
            return np.random.uniform(-1, 1, self.shape).astype(self.dtype)

    class Discrete(object):
        def __init__(self, n):
            self.n = n
            self.shape = ()
            self.dtype = np.dtype(np.int64)

    def np_random(seed=None):
        return np.random.RandomState(seed), seed

    gym.Env = Env
    spaces.Box = Box
    spaces.Discrete = Discrete
    seeding.np_random = np_random
    gym.spaces = spaces
    gym.utils = utils
    utils.seeding = seeding
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces
    sys.modules["gym.utils"] = utils
    sys.modules["gym.utils.seeding"] = seeding


def _install_torch_legacy():
    T = torch.Tensor
    if getattr(T, "_pddp_legacy", False):
        return

    def potrf(self, upper=True):
        return torch.linalg.cholesky(self, upper=upper)

    def potrs(b, u, upper=True):
        squeeze = b.dim() == 1
        if squeeze:
            b = b.unsqueeze(-1)
        x = torch.cholesky_solve(b, u, upper=upper)
        return x

    def gesv(b, a):
        return torch.linalg.solve(a, b), None

    def trtrs(b, a, upper=True, transpose=False, unitriangular=False):
        return torch.triangular_solve(
            b, a, upper=upper, transpose=transpose, unitriangular=unitriangular)

    def eig(self, eigenvectors=False):
        w, v = torch.linalg.eig(self)
        e = torch.stack([w.real, w.imag], dim=-1).to(self.dtype)
        return e, v.real.to(self.dtype)

    T.potrf = potrf
    T.potrs = potrs
    T.gesv = gesv
    T.eig = eig
    torch.potrs = potrs
    torch.gesv = gesv
    torch.trtrs = trtrs

    _getitem = T.__getitem__
    _setitem = T.__setitem__

    def _fix(idx):
        if isinstance(idx, torch.Tensor) and idx.dtype == torch.uint8:
            return idx.bool()
        if isinstance(idx, tuple):
            return tuple(_fix(i) for i in idx)
        return idx

    def getitem(self, idx):
        return _getitem(self, _fix(idx))

    def setitem(self, idx, value):
        return _setitem(self, _fix(idx), value)

    T.__getitem__ = getitem
    T.__setitem__ = setitem
    T._pddp_legacy = True


def import_reference():
    """Returns the reference `pddp` package, imported with the shims."""
    sys.dont_write_bytecode = True
    _install_gym_stub()
    if not hasattr(collections, "Iterable"):
        collections.Iterable = collections.abc.Iterable
    _install_torch_legacy()
    if REFERENCE_PATH not in sys.path:
        sys.path.insert(0, REFERENCE_PATH)
    import pddp  # noqa: E402  (the reference; read-only)
    import pddp.examples  # noqa: F401
    return pddp
