"""The BNN hot path in double precision on the HIP kernels (the reference runs
in the dtype of its inputs: pddp/models/bnn/modules.py:287-386; a float64
controller used to fall to ~150 torch launches per time step):
pddp_bnn_mlp_f64 / pddp_bnn_mlp_jvp_rows_f64 (csrc/bnn_mlp_f64.hip, the f64
matrix cores), pddp_bnn_moment_step_f64, pddp_bnn_jvp_features_f64 /
_moments_f64, pddp_qr_cost_derivs_f64 - against the torch path, and at the
real size against the reference's own float64 outputs."""
import numpy as np
import pytest
import torch

from test_gpu_parity import BOUND, MEAN0, _bnn_real_size_run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H", [64, 128, 200])
@pytest.mark.parametrize("rows,P,in_dim,out_dim", [(1, 100, 6, 8), (37, 100, 6, 8),
                                                   (5, 7, 4, 4), (64, 33, 15, 16),
                                                   (301, 100, 6, 8),
                                                   (3, 100, 9, 12)])
def test_bnn_mlp_f64_kernel_vs_torch(rows, P, in_dim, out_dim, H):
    """pddp_bnn_mlp_f64 (fused fc -> mask -> ReLU x2 -> fc on
    v_mfma_f64_16x16x4_f64) against the same network layer by layer in torch
    float64 (modules.py:774-864): ragged row counts, particle counts that do
    not divide the 16-row tile, the widest supported input / output, more
    tiles than one workgroup pass."""
    from pddp_amd.models.bnn import BayesianMLP
    torch.manual_seed(H + rows)
    net = BayesianMLP(in_dim, out_dim, [H, H]).cuda().double().eval()
    x = torch.randn(rows, P, in_dim, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        assert net._native_ok(x, False)
        y = net(x)                       # native (draws the masks)
        net.use_native = False
        assert not net._native_ok(x, False)
        ref = net(x)                     # library GEMMs, same masks
    assert y.dtype == torch.float64 and y.shape == ref.shape
    err = float((y - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-13, err


def test_bnn_mlp_f64_live_rows_only():
    """`live_rows` (a device count): rows beyond it are neither computed nor
    written (pddp_bnn_mlp_rows_f64)."""
    from pddp_amd.models.bnn import BayesianMLP
    torch.manual_seed(3)
    net = BayesianMLP(6, 4, [200, 200]).cuda().double().eval()
    x = torch.randn(40, 100, 6, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        full = net._forward_native(x)
        live = torch.tensor([1700], dtype=torch.int32, device="cuda")
        part = net._forward_native(x, live_rows=live)
    # (the output is torch.empty: what was not written is whatever was there)
    assert torch.equal(part.reshape(-1, 4)[:1700], full.reshape(-1, 4)[:1700])


@pytest.mark.parametrize("G,live", [(8, None), (16, None), (8, 6), (8, 4),
                                    (8, 5), (8, 3)])
@pytest.mark.parametrize("H", [64, 200])
@pytest.mark.parametrize("groups,P,in_dim,out_dim", [(1, 100, 6, 4), (37, 100, 6, 4),
                                                     (203, 7, 4, 2), (64, 33, 15, 16)])
def test_bnn_mlp_jvp_f64_kernel_vs_torch(groups, P, in_dim, out_dim, H, G, live):
    """pddp_bnn_mlp_jvp_rows_f64: groups of 8 / 16 rows = one input and its
    tangent directions, biases only on the input row, ReLUs linearised at it -
    against the same forward-mode pass written out layer by layer
    (utils/evaluation.py:203-235 through modules.py:774-864)."""
    from pddp_amd.models.bnn import BayesianMLP
    torch.manual_seed(H + groups)
    net = BayesianMLP(in_dim, out_dim, [H, H]).cuda().double().eval()
    F = torch.randn(groups, G, in_dim, device="cuda", dtype=torch.float64)
    with torch.no_grad():
        Y = net._jvp_native(F.reshape(groups * G, in_dim).contiguous(), P,
                            out_dim, G, live=live).reshape(groups, G, out_dim)
        if live is not None:
            Y, F = Y[:, :live], F[:, :live]
        W1, b1 = net.hidden[0].weight, net.hidden[0].bias
        W2, b2 = net.hidden[1].weight, net.hidden[1].bias
        W3, b3 = net.out.weight, net.out.bias
        pidx = torch.arange(groups, device="cuda") % P
        m1 = net.drops[0]._mask(net.drops[0].noise)[pidx].unsqueeze(1)
        m2 = net.drops[1]._mask(net.drops[1].noise)[pidx].unsqueeze(1)
        h1 = F @ W1.T
        h1[:, :1] += b1
        a1 = torch.where(h1[:, :1] * m1 > 0, h1 * m1, torch.zeros_like(h1))
        h2 = a1 @ W2.T
        h2[:, :1] += b2
        a2 = torch.where(h2[:, :1] * m2 > 0, h2 * m2, torch.zeros_like(h2))
        ref = a2 @ W3.T
        ref[:, :1] += b3
    err = float((Y - ref).abs().max()) / float(ref.abs().max())
    assert err < 1e-13, err


def test_bnn_mlp_jvp_f64_refuses_groups_of_32():
    from pddp_amd import _native
    from pddp_amd.models.bnn import BayesianMLP
    net = BayesianMLP(6, 4, [64, 64]).cuda().double().eval()
    F = torch.randn(64, 6, device="cuda", dtype=torch.float64)
    with pytest.raises(_native.NativeError):
        net._jvp_native(F, 10, 4, 32)


def _problem_parts(problem, dtype=torch.float64):
    """(known-dynamics model class, its cost, D, m) of a sample problem."""
    import pddp_amd
    mod = getattr(pddp_amd.examples, problem)
    KM = [getattr(mod, k) for k in dir(mod) if k.endswith("DynamicsModel")
          and k != "DynamicsModel"][0]
    cost = [getattr(mod, k) for k in dir(mod) if k.endswith("Cost")
            and k != "AugmentedQRCost"][0]().to(dtype).cuda()
    D, m = KM.state_size, KM.action_size
    return KM, cost, D, m


@pytest.mark.parametrize("model_opts", [{"use_predicted_std": False},
                                        {"use_predicted_std": True}],
                         ids=["mean_only", "predicted_std"])
@pytest.mark.parametrize("problem,H,P", [("cartpole", 64, 30), ("cartpole", 200, 100),
                                         ("pendulum", 64, 40),
                                         ("double_cartpole", 128, 70)])
def test_bnn_f64_line_search_vs_torch_path(problem, H, P, model_opts):
    """The moment-matched line search in float64: N + 1
    pddp_bnn_moment_step_f64 launches with pddp_bnn_mlp_rows_f64 in between
    against the same rollout made of torch float64 ops
    (controllers/plugin.py:line_search; ilqr.py:677-723, 764-791 through
    modules.py:287-386): candidates' encoded states, controls and costs."""
    import pddp_amd
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.models.bnn import bnn_dynamics_model_factory
    dt = torch.float64
    torch.manual_seed(7)
    KM, cost, D, m = _problem_parts(problem)
    cls = bnn_dynamics_model_factory(D, m, [H, H], KM.angular_indices,
                                     KM.non_angular_indices)
    model = cls(n_particles=P).double().cuda().eval()
    with torch.no_grad():  # keep the learned dynamics gentle: dx ~ 1e-2
        model.model.out.weight.mul_(0.05)
        model.model.out.bias.mul_(0.05)
    enc = pddp_amd.StateEncoding.DEFAULT
    B, N = 5, 9
    n = D + D * (D + 1) // 2
    bound = BOUND[problem]
    res = []
    for native in (True, False):
        plugin = TorchProblem(model, cost, enc, dict(model_opts), {})
        plugin.use_native_bnn = native
        s = ILQRSolver(None, B, N, dt, "cuda",
                       torch.full((m,), -bound, dtype=dt),
                       torch.full((m,), bound, dtype=dt),
                       fit_alphas(dt, "cuda"), plugin=plugin, n=n, m=m)
        g = torch.Generator().manual_seed(1)
        mean = torch.tensor(MEAN0[problem], dtype=dt)
        z0 = torch.stack([pddp_amd.GaussianVariable(
            mean + 1e-2 * torch.randn(D, generator=g, dtype=dt),
            var=1e-2 * torch.ones(D, dtype=dt)).encode(enc)
            for _ in range(B)]).cuda()
        U = (0.1 * torch.randn(B, N, m, generator=g, dtype=dt)).cuda()
        s.set_nominal(z0, U)
        s.gains.copy_(1e-1 * torch.randn(s.gains.shape, generator=g,
                                         dtype=dt).cuda())
        with torch.no_grad():
            assert plugin._bnn_native_ok(s) == native
        s.line_search()
        res.append((s.Zc.clone(), s.Uc.clone(), s.Jc.clone(), s.Z.clone()))
    (Za, Ua, Ja, Zna), (Zb, Ub, Jb, Znb) = res
    assert float((Zna - Znb).abs().max()) / float(Znb.abs().max()) < 1e-10
    assert float((Znb[:, 1:] - Znb[:, :1]).abs().max()) > 1e-4
    assert torch.isfinite(Za).all() and torch.isfinite(Ja).all()
    for x, y, name in ((Za, Zb, "Zc"), (Ua, Ub, "Uc"), (Ja, Jb, "Jc")):
        err = float((x - y).abs().max()) / max(float(y.abs().max()), 1e-6)
        assert err < 1e-9, (name, err)


@pytest.mark.parametrize("model_opts", [
    {"use_predicted_std": False}, {"use_predicted_std": True},
    {"use_predicted_std": True, "independent_noise": True}],
    ids=["mean_only", "predicted_std", "predicted_std_independent"])
@pytest.mark.parametrize("problem,H,P,B", [("cartpole", 64, 30, 3),
                                           ("cartpole", 200, 100, 2),
                                           ("pendulum", 64, 40, 5),
                                           ("double_cartpole", 128, 60, 3)])
def test_bnn_f64_jacobians_and_cost_vs_autograd_path(problem, H, P, B, model_opts):
    """The packed derivative records of a whole nominal in float64: F_z, F_u in
    forward mode (pddp_bnn_jvp_features_f64 / pddp_bnn_mlp_jvp_rows_f64 /
    pddp_bnn_jvp_moments_f64) and the cost's value, gradient and Hessian by
    hyper-dual evaluation (pddp_qr_cost_derivs_f64) against autograd over the
    replicated input (controllers/plugin.py:_dyn_derivs, the reference's
    utils/evaluation.py:203-288)."""
    import pddp_amd
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.models.bnn import bnn_dynamics_model_factory
    dt = torch.float64
    torch.manual_seed(11)
    KM, cost, D, m = _problem_parts(problem)
    cls = bnn_dynamics_model_factory(D, m, [H, H], KM.angular_indices,
                                     KM.non_angular_indices)
    model = cls(n_particles=P).double().cuda().eval()
    with torch.no_grad():
        model.model.out.weight.mul_(0.05)
        model.model.out.bias.mul_(0.05)
    enc = pddp_amd.StateEncoding.DEFAULT
    N = 7
    n = D + D * (D + 1) // 2
    bound = BOUND[problem]
    recs = []
    for native in (True, False):
        plugin = TorchProblem(model, cost, enc, dict(model_opts), {})
        plugin.use_native_bnn_jvp = native
        plugin.use_native_cost = native
        s = ILQRSolver(None, B, N, dt, "cuda",
                       torch.full((m,), -bound, dtype=dt),
                       torch.full((m,), bound, dtype=dt),
                       fit_alphas(dt, "cuda"), plugin=plugin, n=n, m=m)
        g = torch.Generator().manual_seed(1)
        mean = torch.tensor(MEAN0[problem], dtype=dt)
        z0 = torch.stack([pddp_amd.GaussianVariable(
            mean + 1e-2 * torch.randn(D, generator=g, dtype=dt),
            var=1e-2 * torch.ones(D, dtype=dt)).encode(enc)
            for _ in range(B)]).cuda()
        U = (0.1 * torch.randn(B, N, m, generator=g, dtype=dt)).cuda()
        U[:, 2] = 2 * bound  # one clamped action: derivatives at the bound
        s.set_nominal(z0, U)
        model.output = {}
        s.derivs()
        torch.cuda.synchronize()
        assert plugin.last_derivs_path == (
            {"dynamics": "hip", "cost": "hip"} if native
            else {"dynamics": "autograd", "cost": "autograd"})
        recs.append((s.rec.clone(), s.Z.clone(), s.L.clone()))
    (ra, Za, La), (rb, Zb, Lb) = recs
    assert torch.equal(Za, Zb)
    assert torch.isfinite(ra).all()
    lay = s.lay
    for name, o, w in (("F_z", lay.o_Fz, n * n), ("F_u", lay.o_Fu, n * m),
                       ("L_zz", lay.o_Lzz, n * n), ("L_uz", lay.o_Luz, m * n),
                       ("L_z", lay.o_Lz, n), ("L_uu", lay.o_Luu, m * m),
                       ("L_u", lay.o_Lu, m)):
        a, b = ra[:, :, o:o + w], rb[:, :, o:o + w]
        scale = max(float(b.abs().max()), 1e-3)
        err = float((a - b).abs().max()) / scale
        assert err < 1e-8, (name, err)
    assert float((La - Lb).abs().max()) / float(Lb.abs().max()) < 1e-10


def test_bnn_f64_hip_kernels_vs_reference_real_size():
    """The float64 HIP BNN kernels at the size configs[2] runs them ([200, 200]
    hidden, 100 particles, cartpole DEFAULT encoding n = 14) DIRECTLY against
    the reference's float64 outputs (tests/golden/bnn_cartpole_real_size.npz,
    group f64/: `forward` ilqr.py:393-486 with modules.py:287-386 +
    evaluation.py:242-288, `_control_law` / `_trajectory_cost` ilqr.py:678-791):
    nominal rollout, F_z / F_u, the cost's derivatives, and the 10-candidate
    line search on the gains the fixture fed both of its runs.  Bar: 1e-12
    for everything (measured on the MI355X: rollouts, costs and the line search
    1e-17 .. 8e-16, the Jacobians - through the differential of a Cholesky
    factor - 3e-15 .. 9e-15)."""
    rows = _bnn_real_size_run(torch.float64)
    assert rows[-1]["path"] == {"dynamics": "hip", "cost": "hip"}
    seen = set()
    for r in rows[:-1]:
        assert r["hip_vs_f64"] <= 1e-12, r
        seen.add(r["what"])
    assert seen == {"Z", "F_z", "F_u", "L", "L_z", "L_u", "L_zz", "L_uu",
                    "Z_new", "U_new", "J"}


def test_bnn_f64_fit_runs_on_the_native_path():
    """A float64 iLQR fit under the BNN through the reference's controller
    API: every round goes through the HIP kernels (no torch fall-back), and it
    lands where the torch path does."""
    import pddp_amd
    from pddp_amd.examples import cartpole
    from pddp_amd.models.bnn import bnn_dynamics_model_factory
    dt = torch.float64
    torch.manual_seed(0)
    CM = cartpole.CartpoleDynamicsModel
    model = bnn_dynamics_model_factory(
        4, 1, [64, 64], CM.angular_indices, CM.non_angular_indices)(
            n_particles=30).double().cuda().eval()
    with torch.no_grad():
        model.model.out.weight.mul_(0.05)
        model.model.out.bias.mul_(0.05)
    cost = cartpole.CartpoleCost().double().cuda()
    ctrl = pddp_amd.controllers.iLQRController(
        None, model, cost,
        model_opts={"use_predicted_std": False, "infer_noise_variables": True})
    g = torch.Generator().manual_seed(2)
    B, N = 3, 8
    U = (0.1 * torch.randn(B, N, 1, generator=g, dtype=dt)).cuda()
    enc = pddp_amd.StateEncoding.DEFAULT
    z0 = torch.stack([pddp_amd.GaussianVariable(
        torch.tensor([0.0, 0.0, 3.0, 0.0], dtype=dt) +
        1e-2 * torch.randn(4, generator=g, dtype=dt),
        var=1e-2 * torch.ones(4, dtype=dt)).encode(enc)
        for _ in range(B)]).cuda()
    kw = dict(encoding=enc, n_iterations=4, tol=0.0, quiet=True, z0=z0,
              u_min=torch.tensor([-10.0], dtype=dt),
              u_max=torch.tensor([10.0], dtype=dt))
    Za, Ua, _ = ctrl.fit(U, **kw)
    plugin = ctrl._solver.plugin
    assert Za.dtype == dt
    assert plugin.last_derivs_path == {"dynamics": "hip", "cost": "hip"}
    with torch.no_grad():
        assert plugin._bnn_native_ok(ctrl._solver)
    Za, Ua = Za.clone(), Ua.clone()
    plugin.use_native_bnn = False
    plugin.use_native_bnn_jvp = False
    plugin.use_native_cost = False
    Zb, Ub, _ = ctrl.fit(U, **kw)
    assert plugin.last_derivs_path == {"dynamics": "autograd",
                                       "cost": "autograd"}
    assert float((Ua - Ub).abs().max()) < 1e-6 * max(1.0, float(Ub.abs().max()))
    assert float((Za - Zb).abs().max()) < 1e-6 * max(1.0, float(Zb.abs().max()))


@pytest.mark.parametrize("problem,B,N", [("cartpole", 4096, 100),
                                         ("double_cartpole", 1024, 150)])
def test_full_size_bnn_round_f64(problem, B, N):
    """BASELINE.json configs[2] and one GPU's shard of configs[3]'s problem at
    FULL size in float64, one round of the fit loop on the float64 HIP kernels
    (test_full_size_bnn_round is the float32 one): duplicated trajectories are
    bit-identical; a 64-trajectory sample re-run as its own batch is
    bit-identical; the same sample against the float64 TORCH path (autograd
    Jacobians, torch line search: the checker the float32 kernels are held to)
    - nominal rollout 1e-12, derivative records 1e-8, candidate costs under the
    HIP gains 1e-10, and the same best step size."""
    import copy
    from test_gpu_parity import _bnn_problem, rel_err
    model, cost, z0, U, solver = _bnn_problem(problem, B, N)
    dup = [(0, B - 1), (17, B // 2)]
    for a, b in dup:
        z0[b], U[b] = z0[a], U[a]
    m64 = copy.deepcopy(model).double()
    m64.eps_in = {k: v.double() for k, v in model.eps_in.items()}
    m64.output = {}
    c64 = copy.deepcopy(cost).double()
    z0, U = z0.double(), U.double()
    s = solver(m64, c64, B, torch.float64, native64=True)
    s.set_nominal(z0, U)
    with torch.no_grad():
        assert s.plugin._bnn_native_ok(s)
    s.round(n_iterations=50)
    torch.cuda.synchronize()
    assert s.plugin.last_derivs_path == {"dynamics": "hip", "cost": "hip"}
    assert int((s.bwd_status != 0).sum()) == 0
    for a, b in dup:
        for t in (s.rec, s.gains, s.Jc, s.J_opt, s.U, s.state, s.mu):
            assert torch.equal(t[a], t[b]), (a, b)
    assert torch.isfinite(s.Jc).all()
    sample = np.random.RandomState(2).choice(B, 64, replace=False)
    sample[:2] = (0, B - 1)
    idx = torch.from_numpy(sample).cuda()
    gains_full, Jc_full, rec_full = s.gains[idx].clone(), s.Jc[idx].clone(), s.rec[idx].clone()
    s2 = solver(m64, c64, 64, torch.float64, native64=True)
    s2.set_nominal(z0[idx], U[idx])
    Z_nom = s2.Z.clone()
    s2.derivs(mask=s2.fresh)
    rec_small = s2.rec.clone()
    s2.backward(active=s2.active)
    s2.line_search(active=s2.active)
    assert torch.equal(rec_small, rec_full)
    assert torch.equal(s2.gains, gains_full)
    assert torch.equal(s2.Jc, Jc_full)
    # ---- the float64 torch path on the sample
    # (a copy of the float64 model AFTER its dropout and particle noise were
    # drawn, with the HIP kernels switched off)
    t64, tc64 = copy.deepcopy(m64), copy.deepcopy(c64)
    t64.model.use_native = False
    t64.output = {}
    s3 = solver(t64, tc64, 64, torch.float64)
    s3.set_nominal(z0[idx], U[idx])
    assert rel_err(Z_nom.cpu().numpy(), s3.Z.cpu().numpy()) < 1e-12
    s3.derivs(mask=s3.fresh)
    assert s3.plugin.last_derivs_path == {"dynamics": "autograd",
                                          "cost": "autograd"}
    lay, n, m = s3.lay, s3.n, s3.m
    for name, blocks in (
            ("F_zu", ((lay.o_Fz, n * n), (lay.o_Fu, n * m))),
            ("L_zu", ((lay.o_Lz, n), (lay.o_Lu, m))),
            ("L_zuzu", ((lay.o_Lzz, n * n), (lay.o_Luz, m * n),
                        (lay.o_Luu, m * m)))):
        a = np.concatenate([rec_small[..., o:o + c].cpu().numpy()
                            for o, c in blocks], -1)
        b = np.concatenate([s3.rec[..., o:o + c].cpu().numpy()
                            for o, c in blocks], -1)
        assert rel_err(a, b) < 1e-8, (name, rel_err(a, b))
    s3.gains.copy_(s2.gains)
    s3.bwd_status.zero_()
    s3.line_search(active=s3.active)
    Ja, Jb = s2.Jc.cpu().numpy(), s3.Jc.cpu().numpy()
    assert rel_err(Ja, Jb) < 1e-10, rel_err(Ja, Jb)
    srt = np.sort(Jb, axis=1)
    clear = (srt[:, 1] - srt[:, 0]) > 1e-8 * np.abs(srt[:, 0])
    assert clear.sum() >= 16
    assert np.array_equal(Ja.argmin(1)[clear], Jb.argmin(1)[clear])
