"""BNN dynamics model (pddp_amd.models.bnn) against outputs of the reference's
BNNDynamicsModel captured with identical weights, dropout noise and particle
noise (tools/make_golden.py capture_bnn -> tests/golden/bnn_*.npz).  The model
is device-agnostic torch; checked on CPU tensors here and on the GPU through
the plugin path in test_gpu_parity."""
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR


def _load():
    return np.load(os.path.join(GOLDEN_DIR, "bnn_cartpole_default_f64.npz"))


def _model(g, device="cpu"):
    import pddp_amd
    from pddp_amd.examples.cartpole import CartpoleDynamicsModel as CM
    from pddp_amd.models.bnn import (bnn_dynamics_model_factory,
                                     load_reference_state)
    cls = bnn_dynamics_model_factory(4, 1, [32, 24], CM.angular_indices,
                                     CM.non_angular_indices)
    model = cls(n_particles=int(g["P"])).double().eval()
    state = {k[len("state/"):]: g[k] for k in g.files
             if k.startswith("state/")}
    load_reference_state(model, state)
    return model.to(device)


OPTS = {"use_predicted_std": False, "infer_noise_variables": True}


def test_bnn_single_state_rollout_matches_reference():
    from pddp_amd import StateEncoding
    g = _load()
    model = _model(g)
    z = torch.from_numpy(g["z0"])
    U = torch.from_numpy(g["U"])
    assert np.allclose(z.numpy(), g["single/Z"][0])
    for i in range(U.shape[0]):
        z = model(z, U[i], i, StateEncoding.DEFAULT, **OPTS).detach()
        assert np.allclose(z.numpy(), g["single/Z"][i + 1], rtol=1e-9,
                           atol=1e-11), i


def test_bnn_batched_rows_match_reference():
    """Different rows per call (what the line search feeds): shared eps and
    dropout masks, per-row re-whitening of the previous particles."""
    from pddp_amd import StateEncoding
    g = _load()
    model = _model(g)
    z = torch.from_numpy(g["rows/Z0"])
    U = torch.from_numpy(g["rows/U"])
    for i in range(U.shape[0]):
        z = model(z, U[i], i, StateEncoding.DEFAULT, **OPTS).detach()
        assert np.allclose(z.numpy(), g["rows/Z"][i], rtol=1e-8,
                           atol=1e-10), i


def test_bnn_jacobians_match_reference():
    """F_z, F_u (14 x 14, 14 x 1) through whitening, the particle MLP, the
    covariance and its Cholesky factor: the plugin path's replicated-input
    autograd against the reference's batch_eval_dynamics."""
    from pddp_amd import StateEncoding
    from pddp_amd.controllers.plugin import TorchProblem
    g = _load()
    model = _model(g)
    tp = TorchProblem(model, None, StateEncoding.DEFAULT, model_opts=OPTS)
    z = torch.from_numpy(g["z0"]).unsqueeze(0)
    U = torch.from_numpy(g["U"])
    for i in range(U.shape[0]):
        Fz, Fu = tp._dyn_derivs(z, U[i].unsqueeze(0), i)
        assert np.allclose(Fz[0].numpy(), g["jac/F_z"][i], rtol=1e-7,
                           atol=1e-9), i
        assert np.allclose(Fu[0].numpy(), g["jac/F_u"][i], rtol=1e-7,
                           atol=1e-9), i
        z = torch.from_numpy(g["jac/Z"][i + 1]).unsqueeze(0)


def test_bnn_fit_reduces_loss_and_keeps_api():
    """Training loop (modules.py:131-198): learns a linear map, normalisation
    buffers are set, resample() clears every cache."""
    from pddp_amd.models.bnn import (bnn_dynamics_model_factory,
                                     gaussian_log_likelihood)
    torch.manual_seed(0)
    cls = bnn_dynamics_model_factory(3, 1, [32, 32])
    model = cls(n_particles=10)
    X = torch.randn(256, 3)
    U = torch.randn(256, 1)
    dX = 0.1 * X + 0.2 * U
    def nll():
        model.eval()
        out = model.model((torch.cat([X, U], -1) - model.X_mean)
                          * model.X_std_inv)
        mean, log_std = out.split([3, 3], -1)
        mean = mean * model.dX_std + model.dX_mean
        log_std = log_std + model.dX_std.log()
        return float(-gaussian_log_likelihood(dX, mean, log_std.exp()).mean())
    model.fit(X, U, dX, n_iter=1, quiet=True)
    before = nll()
    model.fit(X, U, dX, n_iter=300, learning_rate=1e-2, quiet=True)
    assert nll() < before - 0.5
    assert model.X_mean.shape == (4,) and model.dX_std.shape == (3,)
    from pddp_amd import StateEncoding
    z = torch.cat([torch.zeros(3), 0.1 * torch.ones(3)])
    model.eval()
    z1 = model(z, torch.zeros(1), 0, StateEncoding.STANDARD_DEVIATION_ONLY)
    assert z1.shape == (6,) and 0 in model.eps_in and 0 in model.output
    model.resample()
    assert model.eps_in == {} and model.output == {}
    assert all(d.noise is None for d in model.model.drops)


def test_bnn_real_size_rollout_matches_reference_f64():
    """The torch model at [200, 200] x 100 particles, float64 on the CPU,
    against the reference's float64 rollout with the same weights and noise
    (tests/golden/bnn_cartpole_real_size.npz; the HIP kernels are held to the
    same fixture in test_gpu_parity)."""
    from pddp_amd import StateEncoding
    from pddp_amd.examples.cartpole import CartpoleDynamicsModel as CM
    from pddp_amd.models.bnn import (bnn_dynamics_model_factory,
                                     load_reference_state)
    g = np.load(os.path.join(GOLDEN_DIR, "bnn_cartpole_real_size.npz"))
    P, H = int(g["P"]), int(g["H"])
    model = bnn_dynamics_model_factory(
        4, 1, [H, H], CM.angular_indices, CM.non_angular_indices)(
            n_particles=P).double().eval()
    load_reference_state(model, {k[len("state/"):]: g[k] for k in g.files
                                 if k.startswith("state/")})
    for r in range(g["z0"].shape[0]):
        model.output = {}
        z = torch.from_numpy(g["z0"][r]).double()
        U = torch.from_numpy(g["U"][r]).double()
        ref = g["f64/%d/fwd/Z" % r]
        for i in range(U.shape[0]):
            z = model(z, U[i].clamp(-10.0, 10.0), i, StateEncoding.DEFAULT,
                      **OPTS).detach()
            assert np.allclose(z.numpy(), ref[i + 1], rtol=1e-8,
                               atol=1e-10), (r, i)
