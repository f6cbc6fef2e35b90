"""GP dynamics plugin (pddp_amd/models/gp.py; BASELINE.json north_star "GP
dynamics", configs[3]).  The reference has no GP: PARITY UNPINNED - the torch
model is held to the independent numpy restatement (oracle/gp_port.py) and to
sampling."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import pddp_amd  # noqa: E402
from pddp_amd import GaussianVariable, StateEncoding  # noqa: E402
from pddp_amd.models.gp import gp_dynamics_model_factory  # noqa: E402


def _toy(seed=0, M=14, angular=(2,)):
    """A small GP on a cartpole-shaped system (D = 4, one angle, m = 1)."""
    g = torch.Generator().manual_seed(seed)
    D, m = 4, 1
    X = torch.randn(M, D, generator=g, dtype=torch.float64)
    U = torch.randn(M, m, generator=g, dtype=torch.float64)
    dX = 0.3 * torch.sin(X @ torch.randn(D, D, generator=g, dtype=torch.float64)) \
        + 0.2 * U
    cls = gp_dynamics_model_factory(D, m, list(angular))
    model = cls().double()
    model.fit(X, U, dX)
    return model, X, U, dX


def _oracle_state(model):
    import oracle.gp_port as gp
    ell = model.log_ell.exp().detach().numpy()
    sf2 = (2 * model.log_sf).exp().detach().numpy()
    sn2 = (2 * model.log_sn).exp().detach().numpy()
    return gp, ell, sf2, sn2


def test_gp_conditioning_and_moments_vs_numpy_port():
    model, X, U, dX = _toy()
    gp, ell, sf2, sn2 = _oracle_state(model)
    Xt = model.Xt.numpy()
    Kinv, beta = gp.condition(Xt, dX.numpy(), ell, sf2, sn2)
    assert np.allclose(model.Kinv.numpy(), Kinv, rtol=1e-8, atol=1e-8)
    assert np.allclose(model.beta.numpy(), beta, rtol=1e-8, atol=1e-9)
    g = torch.Generator().manual_seed(1)
    d = Xt.shape[1]
    for trial in range(3):
        m = 0.5 * torch.randn(d, generator=g, dtype=torch.float64)
        A = 0.3 * torch.randn(d, d, generator=g, dtype=torch.float64)
        S = A @ A.t()
        S[-1, :] = 0.0
        S[:, -1] = 0.0  # the action is known exactly
        with torch.no_grad():
            mu, Sig, W = model.moments(m, S)
        mu_o, Sig_o, V_o = gp.moments(Xt, Kinv, beta, ell, sf2, sn2,
                                      m.numpy(), S.numpy())
        assert np.allclose(mu.numpy(), mu_o, rtol=1e-9, atol=1e-11)
        assert np.allclose(Sig.numpy(), Sig_o, rtol=1e-8, atol=1e-10)
        assert np.allclose((S @ W).numpy(), V_o, rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("encoding", [StateEncoding.DEFAULT,
                                      StateEncoding.FULL_COVARIANCE_MATRIX,
                                      StateEncoding.VARIANCE_ONLY])
def test_gp_forward_vs_numpy_port(encoding):
    from pddp_amd.utils.encoding import decode_covar, decode_mean
    model, X, U, dX = _toy(seed=2)
    gp, ell, sf2, sn2 = _oracle_state(model)
    Xt = model.Xt.numpy()
    Kinv, beta = gp.condition(Xt, dX.numpy(), ell, sf2, sn2)
    g = torch.Generator().manual_seed(3)
    rows = []
    for r in range(3):
        mean = torch.tensor([0.1, -0.2, 0.7, 0.05], dtype=torch.float64) + \
            0.1 * torch.randn(4, generator=g, dtype=torch.float64)
        A = 0.15 * torch.randn(4, 4, generator=g, dtype=torch.float64)
        covar = A @ A.t() + 1e-3 * torch.eye(4, dtype=torch.float64)
        rows.append(GaussianVariable(mean, covar=covar).encode(encoding))
    z = torch.stack(rows)
    u = 0.3 * torch.randn(3, 1, generator=g, dtype=torch.float64)
    with torch.no_grad():
        zn = model(z, u, 0, encoding)
    assert zn.shape == z.shape
    for r in range(3):
        mean = decode_mean(z[r], encoding, state_size=4).numpy()
        covar = decode_covar(z[r], encoding, state_size=4).numpy()
        Mn, Cn = gp.step(Xt, Kinv, beta, ell, sf2, sn2, mean, covar,
                         u[r].numpy(), [2], [0, 1, 3])
        got_m = decode_mean(zn[r], encoding, state_size=4).numpy()
        got_C = decode_covar(zn[r], encoding, state_size=4).numpy()
        assert np.allclose(got_m, Mn, rtol=1e-9, atol=1e-11)
        if encoding == StateEncoding.VARIANCE_ONLY:
            assert np.allclose(np.diag(got_C), np.diag(Cn), rtol=1e-8)
        else:
            assert np.allclose(got_C, Cn, rtol=1e-7, atol=1e-10)
    # a single (unbatched) state gives the same row
    with torch.no_grad():
        z0 = model(z[0], u[0], 0, encoding)
    assert torch.allclose(z0, zn[0], rtol=1e-12, atol=1e-14)


def test_gp_moments_vs_sampling():
    """The closed-form moments are those of sample means of the posterior
    mean function's outputs plus the posterior variance (law of total
    variance), up to Monte-Carlo error."""
    model, X, U, dX = _toy(seed=4, M=10)
    gp, ell, sf2, sn2 = _oracle_state(model)
    Xt = model.Xt
    d = Xt.shape[1]
    g = torch.Generator().manual_seed(5)
    m = 0.3 * torch.randn(d, generator=g, dtype=torch.float64)
    A = 0.25 * torch.randn(d, d, generator=g, dtype=torch.float64)
    S = A @ A.t()
    with torch.no_grad():
        mu, Sig, _ = model.moments(m, S)
    n = 200000
    xs = m + torch.randn(n, d, generator=g, dtype=torch.float64) @ \
        torch.linalg.cholesky(S + 1e-12 * torch.eye(d, dtype=torch.float64)).t()
    with torch.no_grad():
        K = model._kernel(xs, Xt)                       # [E, n, M]
        f = torch.einsum("enm,em->ne", K, model.beta)   # posterior means
        kxx = (2 * model.log_sf).exp()
        var = kxx - torch.einsum("enm,emk,enk->ne", K, model.Kinv, K) \
            + (2 * model.log_sn).exp()
    mu_s = f.mean(0)
    Sig_s = torch.cov(f.t()) + torch.diag(var.mean(0))
    assert torch.allclose(mu, mu_s, atol=5e-3)
    assert torch.allclose(Sig, Sig_s, atol=5e-3)


def test_gp_hyperparameter_steps_improve_likelihood():
    model, X, U, dX = _toy(seed=6, M=24)
    Xt = model.Xt
    with torch.no_grad():
        before = float(model._nlml(Xt, dX))
    model.fit(X, U, dX, n_iter=25, learning_rate=0.05)
    with torch.no_grad():
        after = float(model._nlml(model.Xt, dX))
    assert after < before


@pytest.mark.gpu
def test_gp_controller_on_gpu_matches_cpu_model():
    """The GP plugin under iLQRController on the GPU (plugin path: autograd
    Jacobians of the moment-matched step, HIP sweep / accept): the rollout
    equals the CPU model's, and a fit lowers the cost."""
    from pddp_amd.controllers import iLQRController
    from pddp_amd.examples.cartpole import CartpoleCost, CartpoleDynamicsModel
    CM = CartpoleDynamicsModel
    g = torch.Generator().manual_seed(7)
    true = CM(0.1).double()
    enc0 = StateEncoding.IGNORE_UNCERTAINTY
    X = torch.cat([torch.randn(60, 2, generator=g, dtype=torch.float64),
                   3.0 + 0.8 * torch.randn(60, 1, generator=g, dtype=torch.float64),
                   torch.randn(60, 1, generator=g, dtype=torch.float64)], -1)
    U = 3.0 * torch.randn(60, 1, generator=g, dtype=torch.float64)
    with torch.no_grad():
        dX = true(X, U, 0, enc0) - X
    cls = gp_dynamics_model_factory(4, 1, CM.angular_indices,
                                    CM.non_angular_indices)
    cpu = cls().double()
    cpu.fit(X, U, dX)
    gpu = cls().double().cuda()
    gpu.fit(X.cuda(), U.cuda(), dX.cuda())
    enc = StateEncoding.DEFAULT
    z0 = GaussianVariable(torch.tensor([0.0, 0.0, 3.0, 0.0], dtype=torch.float64),
                          var=1e-2 * torch.ones(4, dtype=torch.float64)).encode(enc)
    N = 8
    Us = 0.5 * torch.randn(N, 1, generator=g, dtype=torch.float64)
    z_c, z_g = z0, z0.cuda()
    with torch.no_grad():
        for t in range(N):
            z_c = cpu(z_c, Us[t], t, enc)
            z_g = gpu(z_g, Us[t].cuda(), t, enc)
    assert torch.allclose(z_g.cpu(), z_c, rtol=1e-7, atol=1e-9)
    ctrl = iLQRController(None, gpu, CartpoleCost().double().cuda())
    J = []
    Z, Uo, st = ctrl.fit(Us.cuda(), enc, n_iterations=4, z0=z0.cuda(),
                         u_min=torch.tensor([-10.0], dtype=torch.float64),
                         u_max=torch.tensor([10.0], dtype=torch.float64),
                         on_iteration=lambda i, s_, Z_, U_, J_: J.append(float(J_)))
    assert torch.isfinite(Z).all() and torch.isfinite(Uo).all()
    assert min(J) < J[0] or len(J) == 1
    assert ctrl._solver.plugin is not None


# ---- the HIP kernel of the moment-matched step (csrc/gp_step.hip) -------------
_SYSTEMS = {"pendulum": (2, 1, [0]), "cartpole": (4, 1, [2]),
            "double_cartpole": (6, 1, [1, 2])}


def _system_model(system, M, dtype, seed=0):
    D, m, ang = _SYSTEMS[system]
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(M, D, generator=g, dtype=torch.float64)
    U = torch.randn(M, m, generator=g, dtype=torch.float64)
    dX = 0.3 * torch.sin(X @ torch.randn(D, D, generator=g,
                                          dtype=torch.float64)) + 0.2 * U
    model = gp_dynamics_model_factory(D, m, ang)().double()
    model.fit(X, U, dX)
    return model.to(dtype).cuda(), (X, U, dX)


def _system_rows(system, R, encoding, dtype, seed=1, spread=0.2):
    from pddp_amd.utils.encoding import encode
    D, m, _ = _SYSTEMS[system]
    g = torch.Generator().manual_seed(seed)
    mean = 0.5 * torch.randn(R, D, generator=g, dtype=torch.float64)
    A = spread * torch.randn(R, D, D, generator=g, dtype=torch.float64)
    C = A @ A.transpose(-1, -2) + 1e-3 * torch.eye(D, dtype=torch.float64)
    z = encode(mean, C=C, encoding=encoding)
    u = torch.randn(R, m, generator=g, dtype=torch.float64)
    return z.to(dtype).cuda(), u.to(dtype).cuda()


def _torch_step(model, z, u, encoding, jacobian):
    """The torch module with the kernel switched off; Jacobians by autograd,
    one replica of the row per output (controllers/plugin.py _dyn_derivs)."""
    model.use_native = False
    try:
        if not jacobian:
            with torch.no_grad():
                return model(z, u, 0, encoding)
        R, n = z.shape
        m = u.shape[1]
        rep = torch.cat([z, u], -1).unsqueeze(1).expand(R, n, n + m) \
            .reshape(R * n, n + m).detach().clone().requires_grad_()
        zn = model(rep[:, :n], rep[:, n:], 0, encoding)
        eye = torch.eye(n, dtype=z.dtype, device=z.device).repeat(R, 1)
        J, = torch.autograd.grad(zn, rep, eye)
        J = J.reshape(R, n, n + m)
        return zn.reshape(R, n, n)[:, 0].detach(), J[:, :, :n], J[:, :, n:]
    finally:
        model.use_native = True


def _rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


@pytest.mark.gpu
@pytest.mark.parametrize("system", sorted(_SYSTEMS))
@pytest.mark.parametrize("encoding", [1, 2, 3, 4])
@pytest.mark.parametrize("dtype,tol_step,tol_jac", [
    (torch.float64, 1e-11, 1e-10), (torch.float32, 2e-5, 5e-4)])
def test_gp_step_kernel_vs_torch_module(system, encoding, dtype, tol_step,
                                        tol_jac):
    """pddp_gp_step_* - next encoded state and its Jacobian with respect to
    the encoded state and the action - against the torch module and autograd
    through it: three systems, four encodings, both dtypes; 66 training points
    (more than one tile of 64 lanes)."""
    enc = StateEncoding(encoding)
    model, _ = _system_model(system, 66, dtype)
    z, u = _system_rows(system, 19, enc, dtype)
    assert model.native_ok(z, enc, jacobian=True)
    ref, Fz_r, Fu_r = _torch_step(model, z, u, enc, True)
    out, Fz, Fu = model.native_step(z, u, enc, jacobian=True)
    plain = model.native_step(z, u, enc)
    if dtype == torch.float32:
        # float: both are held to the module in fp64 on the same (float)
        # inputs - the kernel may be as far from it as the float module is
        # (x4), or within the fixed bound
        import copy
        m64 = copy.deepcopy(model).double()
        m64._native_cache = {}
        r64, Fz64, Fu64 = _torch_step(m64, z.double(), u.double(), enc, True)
        for got, tor, exact, tol in ((out, ref, r64, tol_step),
                                     (plain, ref, r64, tol_step),
                                     (Fz, Fz_r, Fz64, tol_jac),
                                     (Fu, Fu_r, Fu64, tol_jac)):
            assert _rel(got.double(), exact) < max(
                tol, 4.0 * _rel(tor.double(), exact))
    else:
        assert _rel(out, ref) < tol_step and _rel(plain, ref) < tol_step
        assert _rel(Fz, Fz_r) < tol_jac and _rel(Fu, Fu_r) < tol_jac
    # and `forward` itself goes through the kernel when nobody can ask for
    # gradients
    with torch.no_grad():
        assert torch.equal(model(z, u, 0, enc), plain)
        assert torch.equal(model(z[0], u[0], 0, enc), plain[0])


@pytest.mark.gpu
def test_gp_step_kernel_vs_numpy_port():
    """The kernel against the independent numpy restatement
    (oracle/gp_port.py `step`), fp64, cartpole-shaped."""
    from pddp_amd.utils.encoding import decode_covar, decode_mean
    model, (X, U, dX) = _system_model("cartpole", 33, torch.float64, seed=4)
    gp, ell, sf2, sn2 = _oracle_state(model.cpu())
    Xt = model.Xt.cpu().numpy()
    Kinv, beta = gp.condition(Xt, dX.numpy(), ell, sf2, sn2)
    model = model.cuda()
    enc = StateEncoding.DEFAULT
    z, u = _system_rows("cartpole", 5, enc, torch.float64, seed=6)
    zn = model.native_step(z, u, enc).cpu()
    for r in range(5):
        mean = decode_mean(z[r].cpu(), enc, state_size=4).numpy()
        covar = decode_covar(z[r].cpu(), enc, state_size=4).numpy()
        Mn, Cn = gp.step(Xt, Kinv, beta, ell, sf2, sn2, mean, covar,
                         u[r].cpu().numpy(), [2], [0, 1, 3])
        assert np.allclose(decode_mean(zn[r], enc, state_size=4).numpy(), Mn,
                           rtol=1e-9, atol=1e-11)
        assert np.allclose(decode_covar(zn[r], enc, state_size=4).numpy(), Cn,
                           rtol=1e-7, atol=1e-10)


@pytest.mark.gpu
def test_gp_step_kernel_refuses_what_it_does_not_cover():
    """Full-covariance encoding and unbuilt shapes stay on the torch path
    (native_ok False); the C entry point itself returns PDDP_E_UNSUPPORTED."""
    from pddp_amd import _native
    model, _ = _system_model("cartpole", 20, torch.float64)
    enc = StateEncoding.FULL_COVARIANCE_MATRIX
    z, u = _system_rows("cartpole", 3, enc, torch.float64)
    assert not model.native_ok(z, enc)
    with pytest.raises(_native.NativeError):
        model.native_step(z, u, enc)
    with torch.no_grad():
        out = model(z, u, 0, enc)  # torch ops
    assert out.shape == z.shape and torch.isfinite(out).all()
    assert not model.native_ok(z.cpu(), StateEncoding.DEFAULT)


@pytest.mark.gpu
def test_gp_derivative_records_by_the_kernel_equal_autograd_records():
    """A round of the plugin path with the GP: the kernel's Jacobians
    (`last_derivs_path["dynamics"] == "hip"`) give the records, gains and
    accepted nominals that autograd through the torch module gives."""
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.examples.cartpole import CartpoleCost, CartpoleDynamicsModel
    CM = CartpoleDynamicsModel
    g = torch.Generator().manual_seed(7)
    true = CM(0.1).double()
    X = torch.cat([torch.randn(51, 2, generator=g, dtype=torch.float64),
                   3.0 + 0.8 * torch.randn(51, 1, generator=g,
                                           dtype=torch.float64),
                   torch.randn(51, 1, generator=g, dtype=torch.float64)], -1)
    U = 3.0 * torch.randn(51, 1, generator=g, dtype=torch.float64)
    with torch.no_grad():
        dX = true(X, U, 0, StateEncoding.IGNORE_UNCERTAINTY) - X
    model = gp_dynamics_model_factory(4, 1, CM.angular_indices,
                                      CM.non_angular_indices)().double().cuda()
    model.fit(X.cuda(), U.cuda(), dX.cuda())
    model.eval()
    enc = StateEncoding.DEFAULT
    B, N, n, m = 6, 12, 14, 1
    z0 = torch.stack([GaussianVariable(
        torch.tensor([0.0, 0.0, 3.0, 0.0], dtype=torch.float64) +
        0.05 * torch.randn(4, generator=g, dtype=torch.float64),
        var=1e-2 * torch.ones(4, dtype=torch.float64)).encode(enc)
        for _ in range(B)]).cuda()
    U0 = (0.3 * torch.randn(B, N, m, generator=g, dtype=torch.float64)).cuda()
    sol = []
    for native in (True, False):
        plugin = TorchProblem(model, CartpoleCost().double().cuda(), enc, {},
                              {})
        plugin.use_native_gp = native
        model.use_native = native
        s = ILQRSolver(None, B, N, torch.float64, "cuda",
                       torch.tensor([-10.0], dtype=torch.float64),
                       torch.tensor([10.0], dtype=torch.float64),
                       fit_alphas(torch.float64, "cuda"), plugin=plugin, n=n,
                       m=m)
        s.set_nominal(z0, U0)
        s.round(5e-6, 1e10, 1 << 30)
        assert plugin.last_derivs_path["dynamics"] == \
            ("hip" if native else "autograd")
        rec = s.rec.clone()
        s.round(5e-6, 1e10, 1 << 30)
        sol.append((rec, s.gains.clone(), s.Z.clone(), s.U.clone(),
                    s.J_opt.clone()))
    model.use_native = True
    for a, b in zip(*sol):
        assert torch.allclose(a, b, rtol=1e-7, atol=1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("encoding", [1, 2, 4])
def test_gp_line_search_with_batched_costs_equals_the_generic_one(encoding):
    """The GP line search (N launches of the kernel, stage costs of all
    candidate points in one batched evaluation from the augmented moments)
    against the plugin's generic line search (the cost module per time step,
    with its encode -> decode round trip): candidate states, actions, costs."""
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.examples.cartpole import CartpoleCost, CartpoleDynamicsModel
    from pddp_amd.utils.encoding import infer_encoded_state_size
    CM = CartpoleDynamicsModel
    enc = StateEncoding(encoding)
    g = torch.Generator().manual_seed(11)
    X = torch.cat([torch.randn(40, 2, generator=g, dtype=torch.float64),
                   3.0 + 0.8 * torch.randn(40, 1, generator=g,
                                           dtype=torch.float64),
                   torch.randn(40, 1, generator=g, dtype=torch.float64)], -1)
    U = 3.0 * torch.randn(40, 1, generator=g, dtype=torch.float64)
    with torch.no_grad():
        dX = CM(0.1).double()(X, U, 0, StateEncoding.IGNORE_UNCERTAINTY) - X
    model = gp_dynamics_model_factory(4, 1, CM.angular_indices,
                                      CM.non_angular_indices)().double().cuda()
    model.fit(X.cuda(), U.cuda(), dX.cuda())
    model.eval()
    B, N, m = 5, 9, 1
    n = infer_encoded_state_size(4, enc)
    z0 = torch.stack([GaussianVariable(
        torch.tensor([0.0, 0.0, 3.0, 0.0], dtype=torch.float64) +
        0.05 * torch.randn(4, generator=g, dtype=torch.float64),
        var=1e-2 * torch.ones(4, dtype=torch.float64)).encode(enc)
        for _ in range(B)]).cuda()
    U0 = (0.3 * torch.randn(B, N, m, generator=g, dtype=torch.float64)).cuda()
    got = []
    for fast in ("rollout", "per_step", False):
        plugin = TorchProblem(model, CartpoleCost().double().cuda(), enc, {},
                              {})
        # "rollout": pddp_gp_rollout_* (control law, clamp, step and stage
        # cost in the step's kernel); "per_step": torch ops around
        # pddp_gp_step_* + batched costs; False: the generic line search
        plugin.use_gp_rollout = fast == "rollout"
        s = ILQRSolver(None, B, N, torch.float64, "cuda",
                       torch.tensor([-10.0], dtype=torch.float64),
                       torch.tensor([10.0], dtype=torch.float64),
                       fit_alphas(torch.float64, "cuda"), plugin=plugin, n=n,
                       m=m)
        s.set_nominal(z0, U0)
        s.derivs()
        s.backward(active=s.active)
        assert plugin._gp_line_search_ok(s)
        if not fast:
            plugin._gp_line_search_ok = lambda s_: False
        s.line_search(active=s.active)
        got.append((s.Zc.clone(), s.Uc.clone(), s.Jc.clone()))
    for other in got[1:]:
        for a, b in zip(got[0], other):
            assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("system", ["pendulum", "double_cartpole"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_gp_rollout_kernel_vs_per_step_line_search(system, dtype):
    """pddp_gp_rollout_* on the other two systems (E = 2 and E = 6: eight
    augmented features, the full 8 x 8 lane map of the cost), DEFAULT encoding,
    with masked trajectories and failed sweeps: candidates, actions and costs
    against the per-step torch form; skipped rows untouched."""
    import pddp_amd.examples as ex
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    mod = getattr(ex, system)
    cost_cls = [getattr(mod, k) for k in dir(mod) if k.endswith("Cost") and
                k not in ("AugmentedQRCost", "QRCost")][0]
    enc = StateEncoding.DEFAULT
    # (the GP on the example model's own angular / non-angular indices: the
    # cost's augmented state must be the model's features)
    MC = [getattr(mod, k) for k in dir(mod) if k.endswith("DynamicsModel") and
          k != "DynamicsModel"][0]
    E, m = MC.state_size, 1
    g = torch.Generator().manual_seed(2)
    Md = 24
    Xd = torch.randn(Md, E, generator=g, dtype=torch.float64)
    Ud = torch.randn(Md, m, generator=g, dtype=torch.float64)
    dXd = 0.1 * torch.randn(Md, E, generator=g, dtype=torch.float64)
    model = gp_dynamics_model_factory(E, m, MC.angular_indices,
                                      MC.non_angular_indices)().double().cuda()
    model.fit(Xd.cuda(), Ud.cuda(), dXd.cuda())
    model = model.to(dtype).eval()
    n = E + E * (E + 1) // 2
    B, N = 7, 6
    z0 = torch.stack([GaussianVariable(
        0.3 * torch.randn(E, generator=g, dtype=torch.float64),
        var=1e-2 * torch.ones(E, dtype=torch.float64)).encode(enc)
        for _ in range(B)]).to(dtype).cuda()
    U0 = (0.3 * torch.randn(B, N, m, generator=g)).to(dtype).cuda()
    bound = torch.tensor([2.0], dtype=dtype)
    got = []
    for roll in (True, False):
        plugin = TorchProblem(model, cost_cls().to(dtype).cuda(), enc, {}, {})
        plugin.use_gp_rollout = roll
        s = ILQRSolver(None, B, N, dtype, "cuda", -bound, bound,
                       fit_alphas(dtype, "cuda"), plugin=plugin, n=n, m=m)
        s.set_nominal(z0, U0)
        s.derivs()
        s.mu.fill_(1.0)
        s.backward(active=s.active)
        assert plugin._gp_line_search_ok(s)
        s.active[2] = 0
        s.bwd_status[4] = 3
        s.Zc.fill_(-7.0)
        s.Jc.fill_(-7.0)
        s.line_search(active=s.active)
        got.append((s.Zc.clone(), s.Uc.clone(), s.Jc.clone()))
    live = torch.ones(B, dtype=torch.bool)
    live[2] = live[4] = False
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    for a, b in zip(*got):
        assert torch.allclose(a[live], b[live], rtol=tol, atol=tol), \
            float((a[live] - b[live]).abs().max())
    assert bool((got[0][0][~live] == -7.0).all())   # skipped rows: untouched
    assert bool((got[0][2][~live] == -7.0).all())


@pytest.mark.gpu
def test_pddp_controller_with_the_gp_plugin_on_gpu():
    """PDDPController.fit (pddp.py:61-206) with the GP plugin as the learned
    model, end to end on the GPU: exploration trials, conditioning the GPs on
    the collected data, iLQR on the learned model (derivative records and line
    search on pddp_gp_step), an MPC trial, re-conditioning."""
    from pddp_amd.examples import pendulum
    torch.manual_seed(0)
    np.random.seed(0)
    PM = pendulum.PendulumDynamicsModel
    env = pendulum.PendulumEnv(dt=0.1)
    cost = pendulum.PendulumCost().cuda()
    model = gp_dynamics_model_factory(2, 1, PM.angular_indices,
                                      PM.non_angular_indices)().cuda()
    ctrl = pddp_amd.controllers.PDDPController(
        env, model, cost, model_opts={}, training_opts={"n_iter": 0})
    N = 5
    U0 = 0.1 * torch.randn(N, 1, device="cuda")
    trials = []
    kw = dict(encoding=StateEncoding.DEFAULT, quiet=True, n_iterations=3,
              u_min=torch.tensor([-2.5]), u_max=torch.tensor([2.5]),
              on_trial=lambda t, X, U: trials.append((t, X.shape, U.shape)))
    ctrl.eval()
    Z, U, state = ctrl.fit(U0, **kw)
    assert Z.shape == (N + 1, 5) and U.shape == (N, 1)
    assert model.fitted and model.Xt.shape[0] == 2 * N
    plugin = ctrl._solver.plugin
    assert plugin.last_derivs_path["dynamics"] == "hip"
    ctrl.train()
    Z, U, state = ctrl.fit(U0, max_trials=3, **kw)
    assert trials[-1][1] == (2 * N, 2)             # MPC trial of horizon 2N
    assert torch.isfinite(U).all() and torch.isfinite(Z).all()


def test_gp_step_lds_bytes_query():
    """pddp_gp_step_lds_bytes (a host function: no GPU needed): what decides
    whether a training set fits the kernel - grows with M, more with the
    Jacobian, -1 for a shape that is not built; the limits DESIGN 3.11 quotes."""
    from pddp_amd import _native
    fn = _native.lib().pddp_gp_step_lds_bytes
    assert fn(5, 9, 60, 28, 0, 4) == -1
    a, b = fn(6, 9, 60, 28, 0, 4), fn(6, 9, 61, 28, 0, 4)
    assert 0 < a <= b < fn(6, 9, 61, 28, 1, 4)
    assert fn(6, 9, 60, 28, 1, 8) == 2 * fn(6, 9, 60, 28, 1, 4)
    limit = 160 * 1024
    assert fn(6, 9, 1278, 28, 0, 4) <= limit < fn(6, 9, 1282, 28, 0, 4)
    # with the Jacobian: the per-point g_i table is kept up to 318 (f32) / 74
    # (f64) training points and formed again in phase C beyond (round 5) -
    # the query steps DOWN there, and the form holds up to 890 / 208
    assert fn(6, 9, 318, 28, 1, 4) <= limit
    assert fn(6, 9, 322, 28, 1, 4) < fn(6, 9, 318, 28, 1, 4)
    assert fn(6, 9, 890, 28, 1, 4) <= limit < fn(6, 9, 894, 28, 1, 4)
    assert fn(6, 9, 74, 28, 1, 8) <= limit
    assert fn(6, 9, 208, 28, 1, 8) <= limit < fn(6, 9, 212, 28, 1, 8)


@pytest.mark.gpu
def test_gp_full_size_rounds_of_configs3():
    """BASELINE configs[3] as stated, at its full size (1024 double-cartpole
    trajectories, N = 150, n = 27, 60 training points): two rounds on the HIP
    path.  Size-independent properties: every accepted cost is below the
    nominal's, the nominal rollout reproduces itself (alpha = 1 candidate of
    zero gains = the nominal), the Jacobians of a sample of rows equal autograd
    through the torch module."""
    import bench
    import argparse
    args = argparse.Namespace(batch=1024, horizon=150, steps=1, warmup=0,
                              scaling="weak", no_cpu_baseline=True)
    out = bench.bench_gp(args, emit=False)
    assert out["config"]["gp_step_on"] == "hip"
    assert out["config"]["derivative_path"] == {"dynamics": "hip",
                                                "cost": "hip"}
    assert out["config"]["batch_per_gpu"] == 1024
    assert np.isfinite(out["value"]) and out["value"] > 0
    s = bench._last_gp_solver
    assert torch.isfinite(s.Z).all() and torch.isfinite(s.J_opt).all()
    J0 = s.J_opt.clone()
    s.round(5e-6, 1e10, 1 << 30)
    accepted = (s.state == 1) | (s.state == 5)
    assert bool((s.J_opt[accepted] < J0[accepted]).all())
    assert bool((s.J_opt[~accepted] == J0[~accepted]).all())
    # Jacobians of 6 random (trajectory, step) rows against autograd
    model = s.plugin.model
    g = torch.Generator().manual_seed(0)
    b = torch.randint(0, 1024, (6,), generator=g)
    t = torch.randint(0, 150, (6,), generator=g)
    z = s.Z[b.cuda(), t.cuda()].contiguous()
    u = s.U[b.cuda(), t.cuda()].clamp(-20.0, 20.0).contiguous()
    _, Fz, Fu = model.native_step(z, u, StateEncoding.DEFAULT, jacobian=True)
    import copy
    m64 = copy.deepcopy(model).double()
    _, Fz64, Fu64 = _torch_step(m64, z.double(), u.double(),
                                StateEncoding.DEFAULT, True)
    assert _rel(Fz.double(), Fz64) < 2e-3 and _rel(Fu.double(), Fu64) < 2e-3


@pytest.mark.gpu
def test_gp_rounds_replayed_as_hipgraphs_equal_eager_rounds():
    """`ILQRSolver.fit(graph=True)` with the GP plugin: a round is HIP launches
    and sync-free torch ops, so it is captured (plugin.capture_ok) - the
    replayed rounds leave the state the eager rounds leave."""
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.examples.cartpole import CartpoleCost, CartpoleDynamicsModel
    CM = CartpoleDynamicsModel
    g = torch.Generator().manual_seed(3)
    X = torch.cat([torch.randn(40, 2, generator=g),
                   3.0 + 0.8 * torch.randn(40, 1, generator=g),
                   torch.randn(40, 1, generator=g)], -1)
    U = 3.0 * torch.randn(40, 1, generator=g)
    with torch.no_grad():
        dX = CM(0.1)(X, U, 0, StateEncoding.IGNORE_UNCERTAINTY) - X
    model = gp_dynamics_model_factory(4, 1, CM.angular_indices,
                                      CM.non_angular_indices)().cuda()
    model.fit(X.cuda(), U.cuda(), dX.cuda())
    model.eval()
    enc = StateEncoding.DEFAULT
    B, N, n, m = 8, 10, 14, 1
    z0 = torch.stack([GaussianVariable(
        torch.tensor([0.0, 0.0, 3.0, 0.0]) + 0.05 * torch.randn(4, generator=g),
        var=1e-2 * torch.ones(4)).encode(enc) for _ in range(B)]).cuda()
    U0 = (0.3 * torch.randn(B, N, m, generator=g)).cuda()
    end = []
    for graph in (False, True):
        plugin = TorchProblem(model, CartpoleCost().cuda(), enc, {}, {})
        s = ILQRSolver(None, B, N, torch.float32, "cuda",
                       torch.tensor([-10.0]), torch.tensor([10.0]),
                       fit_alphas(torch.float32, "cuda"), plugin=plugin, n=n,
                       m=m)
        s.set_nominal(z0, U0)
        assert s.graph_ok()
        rounds = s.fit(n_iterations=4, graph=graph, max_rounds=6)
        end.append((rounds, s.Z.clone(), s.U.clone(), s.J_opt.clone(),
                    s.state.clone()))
    assert end[0][0] == end[1][0]
    for a, b in zip(end[0][1:], end[1][1:]):
        assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("M", [2, 3, 5, 63, 64, 65, 127])
def test_gp_step_kernel_training_set_sizes(M):
    """Edge sizes of the training set: fewer points than a step's four, odd
    counts (the zero-weight phantom partners), exactly one lane tile, one more,
    two tiles less one - step and Jacobian against the torch module, fp64."""
    enc = StateEncoding.DEFAULT
    model, _ = _system_model("cartpole", M, torch.float64, seed=M)
    z, u = _system_rows("cartpole", 7, enc, torch.float64, seed=M + 1)
    assert model.native_ok(z, enc, jacobian=True)
    ref, Fz_r, Fu_r = _torch_step(model, z, u, enc, True)
    out, Fz, Fu = model.native_step(z, u, enc, jacobian=True)
    plain = model.native_step(z, u, enc)
    assert _rel(out, ref) < 1e-10 and _rel(plain, ref) < 1e-10
    assert _rel(Fz, Fz_r) < 1e-9 and _rel(Fu, Fu_r) < 1e-9


@pytest.mark.gpu
def test_gp_step_kernel_at_the_outer_loops_dataset_size():
    """BASELINE configs[3]'s shape at the training-set size the outer loop of
    PDDPController.fit produces after its two initial trials (pddp.py:67-71,
    121-150: 2 x N = 300 rows): double cartpole, DEFAULT encoding (n = 27),
    M = 300 - the float kernel's step and Jacobian (what the bench's second GP
    line runs; M <= 318 fits a workgroup's LDS with the Jacobian) against the
    fp64 torch module and autograd on the same inputs with the float module's
    own distance as the yardstick, and the fp64 kernel's step (its Jacobian
    form stops at M = 74: the derivative rollout then goes through autograd)
    to 1e-10."""
    import copy
    enc = StateEncoding.DEFAULT
    model, _ = _system_model("double_cartpole", 300, torch.float32, seed=3)
    z, u = _system_rows("double_cartpole", 5, enc, torch.float32, seed=4)
    assert model.native_ok(z, enc, jacobian=True)
    ref, Fz_r, Fu_r = _torch_step(model, z, u, enc, True)
    out, Fz, Fu = model.native_step(z, u, enc, jacobian=True)
    m64 = copy.deepcopy(model).double()
    m64._native_cache = {}
    r64, Fz64, Fu64 = _torch_step(m64, z.double(), u.double(), enc, True)
    for got, tor, exact, tol in ((out, ref, r64, 2e-5), (Fz, Fz_r, Fz64, 5e-4),
                                 (Fu, Fu_r, Fu64, 5e-4)):
        assert _rel(got.double(), exact) < max(
            tol, 4.0 * _rel(tor.double(), exact))
    assert not m64.native_ok(z.double(), enc, jacobian=True)
    assert m64.native_ok(z.double(), enc)
    assert _rel(m64.native_step(z.double(), u.double(), enc), r64) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("M,dtype", [(319, torch.float32), (640, torch.float32),
                                     (890, torch.float32), (75, torch.float64),
                                     (208, torch.float64)])
def test_gp_jacobian_beyond_the_g_table(M, dtype):
    """The Jacobian form beyond the training-set size whose per-point g_i =
    G_a nu_i table fits a workgroup's LDS (double cartpole: 318 points in f32,
    74 in f64): the launcher drops the table and phase C forms g_i again
    (csrc/gp_step.hip `Lds::gstore`; round 5) - up to 890 / 208 points, the
    sizes PDDPController.fit's data set grows to (pddp.py:67-71: up to 1000
    rows).  Step and Jacobian against the fp64 torch module and autograd on
    the same inputs (f32: with the float module's own distance as the
    yardstick), the first size past the table and the last that fits."""
    import copy
    enc = StateEncoding.DEFAULT
    model, _ = _system_model("double_cartpole", M, dtype, seed=M)
    z, u = _system_rows("double_cartpole", 2, enc, dtype, seed=M + 1)
    assert model.native_ok(z, enc, jacobian=True)
    out, Fz, Fu = model.native_step(z, u, enc, jacobian=True)
    if dtype == torch.float64:
        ref, Fz_r, Fu_r = _torch_step(model, z, u, enc, True)
        assert _rel(out, ref) < 1e-10
        assert _rel(Fz, Fz_r) < 1e-9 and _rel(Fu, Fu_r) < 1e-9
        return
    ref, Fz_r, Fu_r = _torch_step(model, z, u, enc, True)
    m64 = copy.deepcopy(model).double()
    m64._native_cache = {}
    r64, Fz64, Fu64 = _torch_step(m64, z.double(), u.double(), enc, True)
    for got, tor, exact, tol in ((out, ref, r64, 2e-5), (Fz, Fz_r, Fz64, 5e-4),
                                 (Fu, Fu_r, Fu64, 5e-4)):
        assert _rel(got.double(), exact) < max(
            tol, 4.0 * _rel(tor.double(), exact))


@pytest.mark.gpu
def test_gp_kernel_view_follows_a_loaded_state():
    """The kernel's cached view of the model is keyed by the tensors it was
    made from: after `load_state_dict` (or a parameter changed in place) the
    kernel answers with the new model, not the cached one."""
    enc = StateEncoding.DEFAULT
    a, _ = _system_model("cartpole", 20, torch.float64, seed=1)
    b, _ = _system_model("cartpole", 20, torch.float64, seed=2)
    z, u = _system_rows("cartpole", 5, enc, torch.float64)
    out_a, out_b = a.native_step(z, u, enc), b.native_step(z, u, enc)
    assert not torch.allclose(out_a, out_b)
    a.load_state_dict(b.state_dict())
    assert torch.equal(a.native_step(z, u, enc), out_b)
    with torch.no_grad():
        a.log_ell.add_(0.1)
    ref = _torch_step(a, z, u, enc, False)
    assert _rel(a.native_step(z, u, enc), ref) < 1e-11


@pytest.mark.gpu
def test_gp_graphs_follow_a_refit():
    """A captured round / rollout graph holds raw pointers into the kernel's
    view of the GP (training points, beta, K^-1); `model.fit()` frees that view.
    iLQRController(graph=True): fit -> model.fit(new data) -> fit must equal
    what a fresh controller gets from the refitted model (nothing but the
    model's generation says the graphs are stale; the GP twin of
    test_bnn_graphs_follow_model_resample_and_refit)."""
    from pddp_amd.controllers import iLQRController
    from pddp_amd.examples.cartpole import CartpoleCost, CartpoleDynamicsModel
    from pddp_amd.models.bnn import generation
    CM = CartpoleDynamicsModel
    enc0, enc = StateEncoding.IGNORE_UNCERTAINTY, StateEncoding.DEFAULT

    def data(seed, M):
        g = torch.Generator().manual_seed(seed)
        X = torch.cat([torch.randn(M, 2, generator=g),
                       3.0 + 0.8 * torch.randn(M, 1, generator=g),
                       torch.randn(M, 1, generator=g)], -1)
        U = 3.0 * torch.randn(M, 1, generator=g)
        with torch.no_grad():
            dX = CM(0.1)(X, U, 0, enc0) - X
        return X.cuda(), U.cuda(), dX.cuda()
    cls = gp_dynamics_model_factory(4, 1, CM.angular_indices,
                                    CM.non_angular_indices)
    g = torch.Generator().manual_seed(9)
    B, N = 6, 8
    z0 = torch.stack([GaussianVariable(
        torch.tensor([0.0, 0.0, 3.0, 0.0]) + 0.05 * torch.randn(4, generator=g),
        var=1e-2 * torch.ones(4)).encode(enc) for _ in range(B)]).cuda()
    U0 = (0.3 * torch.randn(B, N, 1, generator=g)).cuda()
    kw = dict(n_iterations=3, z0=z0, u_min=torch.tensor([-10.0]),
              u_max=torch.tensor([10.0]), quiet=True)
    model = cls().cuda()
    model.fit(*data(3, 40))
    model.eval()
    ctrl = iLQRController(None, model, CartpoleCost().cuda(), graph=True)
    Z1, U1, _ = ctrl.fit(U0.clone(), enc, **kw)
    assert ctrl._solver._graph is not None
    gen = generation(model)
    model.fit(*data(4, 40))              # same M: same solver key, same sizes
    assert generation(model) > gen
    Z2, U2, _ = ctrl.fit(U0.clone(), enc, **kw)
    fresh = iLQRController(None, model, CartpoleCost().cuda(), graph=False)
    Zf, Uf, _ = fresh.fit(U0.clone(), enc, **kw)
    assert torch.isfinite(Z2).all()
    assert torch.equal(Z2, Zf) and torch.equal(U2, Uf)
    assert not torch.equal(Z1, Z2)       # (the model did change)


def test_cholesky_solve_helper_is_torch_cholesky_solve():
    """utils/linalg.py cholesky_solve (two triangular solves; what the GP
    conditioning and the torch BoxQP call instead of torch.cholesky_solve):
    the same numbers, lower and upper factors, batched and broadcast."""
    from pddp_amd.utils.linalg import cholesky_solve
    g = torch.Generator().manual_seed(0)
    A = torch.randn(3, 7, 7, generator=g, dtype=torch.float64)
    K = A @ A.transpose(-1, -2) + torch.eye(7, dtype=torch.float64)
    Bm = torch.randn(3, 7, 2, generator=g, dtype=torch.float64)
    eye = torch.eye(7, dtype=torch.float64).expand(3, 7, 7)
    L = torch.linalg.cholesky(K)
    U = torch.linalg.cholesky(K, upper=True)
    for rhs in (Bm, eye):
        assert torch.allclose(cholesky_solve(rhs, L),
                              torch.cholesky_solve(rhs, L), rtol=1e-12, atol=1e-12)
        assert torch.allclose(cholesky_solve(rhs, U, upper=True),
                              torch.cholesky_solve(rhs, U, upper=True),
                              rtol=1e-12, atol=1e-12)
    assert torch.allclose(cholesky_solve(Bm[0], L[0]),
                          torch.cholesky_solve(Bm[0], L[0]),
                          rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_gp_fit_on_the_device_leaves_other_tensors_alone():
    """GPDynamicsModel.fit on the device, among many small live tensors: none
    of them changes.  (torch.cholesky_solve's batched path on this ROCm build
    wrote outside its outputs: cached index tensors came back scaled by 1 / L_ii
    and a GPU exception followed many launches later -
    tools/dbg/torch_potrs_canary.py; the conditioning now runs on triangular
    solves, utils/linalg.py.)"""
    dev = torch.device("cuda", 0)
    keep = []
    for r in range(3000):
        n = (64, 30, 128, 256, 100, 512)[r % 6]
        keep.append(torch.full((n,), 3 + (r % 5), dtype=torch.int64, device=dev))
    keep = [t for i, t in enumerate(keep) if i % 3]   # holes between them
    g = torch.Generator().manual_seed(0)
    for it in range(40):
        for E, ai, ni, Md in ((2, [0], [1], 24), (4, [2], [0, 1, 3], 24),
                              (6, [1, 2], [0, 3, 4, 5], 24),
                              (6, [1, 2], [0, 3, 4, 5], 21), (4, [2], [0, 1, 3], 5)):
            X = torch.randn(Md, E, generator=g, dtype=torch.float64)
            U = torch.randn(Md, 1, generator=g, dtype=torch.float64)
            dX = 0.1 * torch.randn(Md, E, generator=g, dtype=torch.float64)
            model = gp_dynamics_model_factory(E, 1, ai, ni)().double().to(dev)
            model.fit(X.to(dev), U.to(dev), dX.to(dev))
    torch.cuda.synchronize()
    bad = [i for i, t in enumerate(keep)
           if not bool((t == t.flatten()[-1]).all()) or
           int(t.flatten()[-1]) not in (3, 4, 5, 6, 7)]
    assert not bad, bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_gp_step_with_a_row_mask(dtype):
    """pddp_gp_step_masked_*: groups of rows with a zero mask entry are skipped
    - their outputs (step and Jacobians) are left as they were - and the other
    rows are the unmasked launch's, bit for bit."""
    model, _ = _system_model("cartpole", 30, dtype)
    enc = StateEncoding.DEFAULT
    g = torch.Generator().manual_seed(5)
    groups, per = 5, 3
    R = groups * per - 1   # (a ragged last group)
    z = torch.stack([GaussianVariable(
        0.3 * torch.randn(4, generator=g, dtype=torch.float64),
        var=1e-2 * torch.ones(4, dtype=torch.float64)).encode(enc)
        for _ in range(R)]).to(dtype).cuda()
    u = (0.3 * torch.randn(R, 1, generator=g)).to(dtype).cuda()
    n = z.shape[1]
    full = [t.clone() for t in model.native_step(z, u, enc, jacobian=True)]
    mask = torch.tensor([1, 0, 1, 0, 1], dtype=torch.uint8, device="cuda")
    Fz = torch.full((R, n, n), 7.0, dtype=dtype, device="cuda")
    Fu = torch.full((R, n, 1), 7.0, dtype=dtype, device="cuda")
    out, Fz2, Fu2 = model.native_step(z, u, enc, jacobian=True, Fz=Fz, Fu=Fu,
                                      row_mask=mask, rows_per_mask=per)
    rows = torch.arange(R, device="cuda") // per
    live = mask[rows].bool()
    assert torch.equal(out[live], full[0][live])
    assert torch.equal(Fz2[live], full[1][live])
    assert torch.equal(Fu2[live], full[2][live])
    assert bool((Fz2[~live] == 7.0).all()) and bool((Fu2[~live] == 7.0).all())
    # forward only (another kernel: against its own unmasked launch)
    full_f = model.native_step(z, u, enc).clone()
    out_f = model.native_step(z, u, enc, row_mask=mask, rows_per_mask=per)
    assert torch.equal(out_f[live], full_f[live])
