"""Environment interface (reference: pddp/envs/base.py:21-75) and a gym-free
environment that steps a ground-truth model (what the reference's example
envs do through gym, e.g. pddp/examples/cartpole/env.py:65-118)."""
import abc

import numpy as np
import torch

from ..utils.encoding import StateEncoding
from ..utils.gaussian_variable import GaussianVariable


class Env(abc.ABC):

    def __enter__(self):
        return self

    def __exit__(self, type, value, traceback):
        self.close()

    @property
    @abc.abstractmethod
    def action_size(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def state_size(self):
        raise NotImplementedError

    @abc.abstractmethod
    def apply(self, u):
        raise NotImplementedError

    @abc.abstractmethod
    def get_state(self, var=1e-2):
        raise NotImplementedError

    @abc.abstractmethod
    def reset(self):
        raise NotImplementedError

    @abc.abstractmethod
    def close(self):
        raise NotImplementedError


class ModelEnv(Env):
    """Headless simulator: state <- model(state, u) under IGNORE_UNCERTAINTY,
    with the state rounded to float32 before each step like the reference's
    gym shims (cartpole/env.py:100-113)."""

    def __init__(self, model, initial_state, reset_noise=1e-2):
        self.model = model.eval()
        self._initial_state = np.asarray(initial_state, dtype=np.float64)
        self._reset_noise = reset_noise
        self._state = None
        self.reset()

    @property
    def action_size(self):
        return self.model.action_size

    @property
    def state_size(self):
        return self.model.state_size

    def apply(self, u):
        dtype = torch.get_default_dtype()
        x = torch.tensor(self._state.astype(np.float32), dtype=dtype)
        u = torch.as_tensor(u).detach().cpu().to(dtype).reshape(-1)
        with torch.no_grad():
            x_next = self.model(x, u, 0,
                                encoding=StateEncoding.IGNORE_UNCERTAINTY)
        self._state = x_next.detach().cpu().numpy()

    def get_state(self, var=1e-2):
        s = torch.tensor(self._state, dtype=torch.get_default_dtype())
        return GaussianVariable(s, var=var * torch.ones_like(s))

    def reset(self):
        self._state = self._initial_state + self._reset_noise * \
            np.random.randn(*self._initial_state.shape)

    def close(self):
        pass
