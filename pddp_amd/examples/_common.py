"""Shared plumbing of the sample problems: constants -> `pddp_problem`."""
import torch

from .. import _native
from ..costs.quadratic import QRCost
from ..utils.angular import augment_encoded_state
from ..utils.encoding import (StateEncoding, decode_mean,
                              infer_encoded_state_size)

MODEL_IDS = {"cartpole": 1, "double_cartpole": 2, "pendulum": 3,
             "rendezvous": 4}  # pddp/examples/problems.py:25-28


class AugmentedQRCost(QRCost):
    """QRCost evaluated on the angle-augmented state of `model_class`
    (reference pattern: pddp/examples/cartpole/cost.py:60-87)."""

    model_class = None

    def forward(self, z, u, i, terminal=False,
                encoding=StateEncoding.DEFAULT, **kwargs):
        mc = self.model_class
        za = augment_encoded_state(z, mc.angular_indices,
                                   mc.non_angular_indices, encoding,
                                   mc.state_size)
        return super(AugmentedQRCost, self).forward(za, u, i, terminal,
                                                    encoding, **kwargs)


def build_problem(name, model, cost, encoding, param_names):
    """Flattens a sample (model, cost) pair into include/pddp_problem.h: the
    problems the HIP kernels evaluate in closed form - every sample problem
    under IGNORE_UNCERTAINTY (csrc/problem_kernels.hip) and cartpole, pendulum,
    double cartpole, rendezvous under DEFAULT = UPPER_TRIANGULAR_CHOLESKY,
    VARIANCE_ONLY, STANDARD_DEVIATION_ONLY and FULL_COVARIANCE_MATRIX
    (csrc/default_kernels.hip; rendezvous carries the full covariance through
    its dynamics, `kCarriesCovar` there).  None otherwise."""
    if encoding not in (StateEncoding.UPPER_TRIANGULAR_CHOLESKY,
                        StateEncoding.VARIANCE_ONLY,
                        StateEncoding.STANDARD_DEVIATION_ONLY,
                        StateEncoding.FULL_COVARIANCE_MATRIX,
                        StateEncoding.IGNORE_UNCERTAINTY):
        return None
    if not isinstance(cost, AugmentedQRCost) or \
            cost.model_class is not type(model):
        return None
    p = _native.PddpProblem()
    p.model = MODEL_IDS[name]
    p.encoding = int(encoding)
    p.state_size = model.state_size
    p.action_size = model.action_size
    p.encoded_size = infer_encoded_state_size(model.state_size, encoding)
    na = cost.Q.shape[0]
    m = model.action_size
    p.aug_size = na
    for i, nm in enumerate(param_names):
        p.params[i] = float(getattr(model, nm).detach().cpu().double())
    Q = cost.Q.detach().cpu().double()
    Qt = cost.Q_term.detach().cpu().double()
    R = cost.R.detach().cpu().double()
    goal = cost.x_goal.detach().cpu().double().expand(na)
    ugoal = cost.u_goal.detach().cpu().double().expand(m)
    for i in range(na):
        p.x_goal[i] = float(goal[i])
        for j in range(na):
            p.Q[i * _native.MAX_AUG + j] = float(Q[i, j])
            p.Q_term[i * _native.MAX_AUG + j] = float(Qt[i, j])
    for i in range(m):
        p.u_goal[i] = float(ugoal[i])
        for j in range(m):
            p.R[i * _native.MAX_ACTION + j] = float(R[i, j])
    return p
