"""PDDP controller: learn a dynamics model from environment trials, optimise
with iLQR on the learned model, refine with MPC trials (reference:
pddp/controllers/pddp.py:32-267; same constructor, `fit` kwargs and helpers).

Host-side orchestration only: environment stepping and dataset plumbing run
here, model training is the plugin's own `fit`, and every optimisation step
goes through `iLQRController` (HIP sweep / accept kernels; all-HIP for the
sample problems, plugin path for learned models).
"""
import torch

from .ilqr import _trajectory_cost, iLQRController, iLQRState
from ..utils.encoding import StateEncoding, decode_mean


class PDDPController(iLQRController):

    def __init__(self, env, model, cost, model_opts={}, cost_opts={},
                 training_opts={}, **kwargs):
        super(PDDPController, self).__init__(env, model, cost, model_opts,
                                             cost_opts, **kwargs)
        self._training_opts = training_opts

    def fit(self, U, encoding=StateEncoding.DEFAULT, quiet=False,
            on_trial=None, max_trials=None, n_initial_sample_trajectories=2,
            sampling_noise=1.0, train_on_start=True, max_dataset_size=1000,
            resample_model=True, u_min=None, u_max=None, **kwargs):
        """pddp.py:61-206.  Returns (Z, U, state) of the last iLQR fit."""
        U = U.detach()
        opts = dict(dtype=U.dtype, device=U.device)
        bounds = {}
        if u_min is not None and u_max is not None:
            bounds = dict(u_min=torch.as_tensor(u_min).to(**opts),
                          u_max=torch.as_tensor(u_max).to(**opts))
        trials = 0
        dataset = None
        if train_on_start:  # initial exploration                    (:119-154)
            for i in range(n_initial_sample_trajectories):
                self.env.reset()
                Ui = U
                if i > 0:
                    Ui = sampling_noise * torch.rand_like(U)
                    if bounds:
                        Ui = (bounds["u_max"] - bounds["u_min"]) * Ui + \
                            bounds["u_min"]
                data, _ = _apply_controller(self.env, self.cost, Ui,
                                            U.shape[0], encoding, False, quiet,
                                            self._cost_opts, **bounds)
                dataset = _concat_datasets(dataset, data, max_dataset_size)
                if callable(on_trial):
                    on_trial(trials, data[0], data[1])
                trials += 1
            self.model.train()
            self.model.fit(*dataset, quiet=quiet, **self._training_opts)

        while True:
            self.env.reset()
            self.model.eval()
            if resample_model and hasattr(self.model, "resample"):
                self.model.resample()  # fresh randomness each episode (:163-165)
            Z, U, state = super(PDDPController, self).fit(
                U, encoding=encoding, quiet=quiet, **bounds, **kwargs)
            if not self.training:
                break
            # closed-loop MPC trial of twice the horizon               (:180-192)
            data, _ = _apply_controller(self.env, self.cost, self,
                                        2 * U.shape[0], encoding, True, quiet,
                                        self._cost_opts, **bounds, **kwargs)
            if callable(on_trial):
                on_trial(trials, data[0], data[1])
            dataset = _concat_datasets(dataset, data, max_dataset_size)
            self.model.train()
            self.model.fit(*dataset, quiet=quiet, **self._training_opts)
            trials += 1
            if max_trials is not None and trials >= max_trials:
                break
        return Z, U, state


def _apply_controller(env, cost, controller, H, encoding, mpc=False,
                      quiet=False, cost_opts={}, **kwargs):
    """Runs `controller` (a feedback controller or an open-loop action
    tensor) on the environment for H steps (pddp.py:209-245).  Returns
    ((X, U, dX), J)."""
    device = None
    if isinstance(controller, torch.Tensor):
        actions = controller
        device = actions.device
        controller = lambda z, i, *a, **k: actions[i]
    else:
        Un = getattr(controller, "_U_nominal", None)
        device = Un.device if Un is not None else None
    Z, U = [], []
    for i in range(H):
        z = env.get_state().encode(encoding)
        if device is not None:
            z = z.to(device)
        Z.append(z)
        u = controller(z, i, encoding, mpc, **kwargs)
        U.append(u)
        env.apply(u)
    z = env.get_state().encode(encoding)
    Z.append(z.to(device) if device is not None else z)
    Z, U = torch.stack(Z).to(U[0].dtype), torch.stack(U)
    J = _trajectory_cost(cost, Z, U, encoding, cost_opts)
    X = decode_mean(Z, encoding=encoding)
    return (X[:-1].detach(), U.detach(), (X[1:] - X[:-1]).detach()), J.detach()


@torch.no_grad()
def _concat_datasets(first, second, max_dataset_size=None):
    """Appends trial data, keeping the LAST `max_dataset_size` rows
    (pddp.py:248-267)."""
    if first is None:
        return second
    if second is None:
        return first
    out = tuple(torch.cat([a, b]) for a, b in zip(first, second))
    if max_dataset_size is not None:
        out = tuple(t[-max_dataset_size:] for t in out)
    return out
