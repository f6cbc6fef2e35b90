"""Sample problem registry (reference: pddp/examples/problems.py:21-95)."""
from enum import IntEnum

from . import cartpole, double_cartpole, pendulum, rendezvous


class SampleProblems(IntEnum):
    CARTPOLE = 1
    DOUBLE_CARTPOLE = 2
    PENDULUM = 3
    RENDEZVOUS = 4

    def _module(self):
        return {1: cartpole, 2: double_cartpole, 3: pendulum,
                4: rendezvous}[int(self)]

    def get_env_class(self):
        mod = self._module()
        return getattr(mod, [n for n in dir(mod) if n.endswith("Env")
                             and n != "ModelEnv"][0])

    def get_cost_class(self):
        mod = self._module()
        return getattr(mod, [n for n in dir(mod) if n.endswith("Cost")
                             and n != "AugmentedQRCost"][0])

    def get_model_class(self):
        mod = self._module()
        return getattr(mod, [n for n in dir(mod)
                             if n.endswith("DynamicsModel")
                             and n != "DynamicsModel"][0])

    def setup(self, dt, render=False, **kwargs):
        """(env, cost, model) like problems.py:30-51."""
        model_class = self.get_model_class()
        model = model_class(dt, **kwargs)
        cost = self.get_cost_class()()
        env = self.get_env_class()(dt=dt, model=model_class(dt, **kwargs),
                                   render=render)
        return env, cost, model
