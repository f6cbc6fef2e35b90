"""Iterative LQR controller on the MI355X.

Same public surface as the reference's pddp/controllers/ilqr.py - `iLQRState`
(:35-64), `iLQRController` (:67-390) and the module functions `forward`
(:393), `Q` (:490), `backward` (:530), `_control_law` (:678),
`_trajectory_cost` (:765) - but every tensor may carry a leading batch axis of
B independent trajectories, and the work runs in the HIP kernels of
libpddp_hip.so (see solver.py).  There is no CPU fallback: CPU tensors raise.
"""
import warnings
import weakref
from enum import IntEnum

import torch

from .base import Controller
from .plugin import TorchProblem
from .solver import (BRANCH_CHOLESKY, BRANCH_EIG, ILQRSolver, fit_alphas,
                     mpc_alphas)
from .. import _native
from ..utils.encoding import StateEncoding, decode_mean


class iLQRState(IntEnum):
    """ilqr.py:35-64"""
    UNDEFINED = 0
    ACCEPTED = 1
    REJECTED = 2
    NOT_PD = 3
    MAX_REG = 4
    CONVERGED = 5

    def should_retry(self):
        return self in (iLQRState.UNDEFINED, iLQRState.NOT_PD,
                        iLQRState.REJECTED)

    def is_terminal(self):
        return self in (iLQRState.CONVERGED, iLQRState.MAX_REG)


def _native_problem(model, cost, encoding):
    """ctypes PddpProblem when (model, cost, encoding) is a sample problem the
    HIP kernels evaluate in closed form, else None (plugin path)."""
    if hasattr(model, "native_problem"):
        return model.native_problem(encoding, cost)
    return None


def _make_solver(model, cost, encoding, B, N, n, dtype, device, u_min, u_max,
                 alphas, model_opts=None, cost_opts=None, force_plugin=False,
                 kernel_variant=0, exact=False):
    problem = None if force_plugin else _native_problem(model, cost, encoding)
    if problem is not None:
        s = ILQRSolver(problem, B, N, dtype, device, u_min, u_max, alphas)
    else:
        plugin = TorchProblem(model, cost, encoding, model_opts, cost_opts)
        s = ILQRSolver(None, B, N, dtype, device, u_min, u_max, alphas,
                       plugin=plugin, n=n, m=model.action_size)
    s.kernel_variant = s.exact_variant() if exact else int(kernel_variant)
    return s


def _as_batch(t, ndim_single):
    """Adds the batch axis when `t` is an un-batched reference-shaped tensor."""
    if t.dim() == ndim_single:
        return t.unsqueeze(0), False
    return t, True


class iLQRController(Controller):
    """ilqr.py:67-390.  `fit(U)` accepts U of shape (N, m) (one trajectory,
    reference semantics) or (B, N, m) (B independent problems; `z0` of shape
    (B, n) may be given, otherwise every trajectory starts from
    `env.get_state()`)."""

    def __init__(self, env, model, cost, model_opts={}, cost_opts={},
                 force_plugin=False, graph=False, exact=False,
                 kernel_variant=0, **kwargs):
        """`graph=True` replays each round as one captured hipGraph (see
        ILQRSolver.capture_round).  `exact=True` runs the fp32 backward sweep
        on the IEEE-division twin of the kernel `auto` picks (the default fp32
        kernels use v_rcp / v_sqrt; measured error identical, see
        profiles/r02_sweep_error_stats.json); `kernel_variant` selects a sweep
        kernel by number (include/pddp_hip.h)."""
        super(iLQRController, self).__init__()
        self._force_plugin = force_plugin
        self._graph = graph
        self._exact = bool(exact)
        self._kernel_variant = int(kernel_variant)
        self.env = env
        self.cost = cost
        self.model = model
        self._cost_opts = cost_opts
        self._model_opts = model_opts
        self._mu = 0.0
        self._mu_min = 1e-6
        self._delta_0 = 2.0
        self._delta = self._delta_0
        self._Z_nominal = None
        self._U_nominal = None
        self._K = None
        self._solver = None
        self._solvers = {}
        self._batched = False
        self._last_rounds = 0  # rounds of the last fit / MPC step

    # -- solver plumbing ------------------------------------------------------
    def _get_solver(self, B, N, n, dtype, device, encoding, u_min, u_max,
                    alphas):
        key = (B, N, n, dtype, torch.device(device), int(encoding),
               None if u_min is None else tuple(
                   torch.as_tensor(u_min).flatten().tolist()),
               None if u_max is None else tuple(
                   torch.as_tensor(u_max).flatten().tolist()),
               tuple(alphas.flatten().tolist()), self._kernel_variant,
               self._exact, self._force_plugin)
        # a small cache: fit() and forward(mpc=True) use different alpha
        # schedules, i.e. different solvers - both stay (with their captured
        # graphs; graph freshness against the model is the solver's business,
        # ILQRSolver._graphs_fresh)
        s = self._solvers.get(key)
        if s is None:
            s = _make_solver(self.model, self.cost, encoding, B, N, n, dtype,
                             device, u_min, u_max, alphas, self._model_opts,
                             self._cost_opts, self._force_plugin,
                             self._kernel_variant, self._exact)
            s._key = key
            s.graph_rollout = bool(self._graph)
            if len(self._solvers) >= 4:  # (oldest out)
                self._solvers.pop(next(iter(self._solvers)))
            self._solvers[key] = s
        self._solver = s
        return s

    def _export(self, s):
        # independent tensors cross the public API, like the reference's
        # `Z_new[:, amin].detach()` (ilqr.py:167-169): the solver (cached per
        # key) overwrites its buffers in the next fit / MPC step
        k, K = s.gain_views(accepted=True)
        if self._batched:
            Z, U, K = s.Z.clone(), s.U.clone(), K.clone()
        else:
            Z, U, K = s.Z[0].clone(), s.U[0].clone(), K[0].clone()
        self._Z_nominal, self._U_nominal, self._K = Z, U, K
        self._mu = float(s.mu[0])
        self._delta = float(s.delta[0])

    def _states(self, s):
        st = s.state.cpu()
        if self._batched:
            return st
        return iLQRState(int(st[0]))

    def _run(self, s, n_iterations, tol, max_reg, on_iteration):
        def on_round(r, s):
            if on_iteration is None:
                return
            if self._batched:
                on_iteration(s.iter.cpu() - 1, s.state.cpu(), s.Z.clone(),
                             s.U.clone(), s.J_opt.clone())
            else:
                # (a callback reads the regularisation of the attempt it is
                # told about, as in the reference: ilqr.py:166-181 update
                # _mu / _delta before on_iteration fires, :232-233)
                self._mu, self._delta = float(s.mu[0]), float(s.delta[0])
                it = int(s.iter[0]) - 1
                st = iLQRState(int(s.state[0]))
                if st == iLQRState.ACCEPTED and int(s.active[0]):
                    it -= 1  # the counter already points at the next step()
                on_iteration(it, st, s.Z[0].clone(), s.U[0].clone(),
                             s.J_opt[0].clone())
        # (nobody watches the attempts: several rounds per launch where the
        # one-launch round applies - cartpole f32 - and one look at the live
        # count per launch)
        self._last_rounds = s.fit(
            n_iterations, tol, max_reg,
            on_round if on_iteration is not None else None,
            graph=self._graph and s.graph_ok(), rounds_per_launch=8)

    # -- reference API ----------------------------------------------------------
    def fit(self, U, encoding=StateEncoding.DEFAULT, n_iterations=50, tol=5e-6,
            max_reg=1e10, batch_rollout=True, quiet=False, on_iteration=None,
            u_min=None, u_max=None, z0=None, **kwargs):
        """ilqr.py:237-316.  Returns (Z, U, state)."""
        _native.require_gpu(U)
        U = U.detach()
        Ub, self._batched = _as_batch(U, 2)
        B, N, m = Ub.shape
        opts = {"dtype": U.dtype, "device": U.device}
        if z0 is None:
            z0 = self.env.get_state().encode(encoding).detach().to(**opts)
        z0 = z0.to(**opts)
        if z0.dim() == 1:
            z0 = z0.unsqueeze(0).expand(B, -1)
        s = self._get_solver(B, N, z0.shape[-1], U.dtype, U.device, encoding,
                             u_min, u_max, fit_alphas(U.dtype, U.device))
        s.set_nominal(z0.contiguous(), Ub.contiguous())
        self._mu, self._delta = 0.0, self._delta_0
        self._run(s, n_iterations, tol, max_reg, on_iteration)
        self._export(s)
        states = self._states(s)
        if (s.state == int(iLQRState.MAX_REG)).any():
            warnings.warn("exceeded max regularization term")
        return self._Z_nominal, self._U_nominal, states

    def step(self, z0, U=None, i=0, encoding=StateEncoding.DEFAULT,
             batch_rollout=True, alphas=None, u_min=None, u_max=None,
             on_iteration=None, **kwargs):
        """ilqr.py:183-235: ONE optimisation step from `z0` around `U`
        (default: the stored nominal controls) - derivative rollout, then
        backward sweep / line search attempts until one is accepted, has
        converged or the regularisation is exhausted.  The regularisation
        state `_mu` / `_delta` carries over between calls exactly as in the
        reference (`fit` resets it once, `forward(mpc=True)` every control
        step); the accepted nominals and gains land in `_Z_nominal`,
        `_U_nominal`, `_K`.  Returns the iLQRState (a tensor of states for a
        batch).  `alphas` defaults to the signature default of the reference,
        10 ** linspace(0, -3, 11)."""
        if U is None:
            U = self._U_nominal
        _native.require_gpu(U)
        U = U.detach()
        Ub, self._batched = _as_batch(U, 2)
        B, N, m = Ub.shape
        z0 = z0.detach().to(dtype=U.dtype, device=U.device)
        if z0.dim() == 1:
            z0 = z0.unsqueeze(0).expand(B, -1)
        al = (mpc_alphas(U.dtype, U.device) if alphas is None
              else alphas.to(dtype=U.dtype, device=U.device))
        s = self._get_solver(B, N, z0.shape[-1], U.dtype, U.device, encoding,
                             u_min, u_max, al)
        s.set_nominal(z0.contiguous(), Ub.contiguous())
        s.mu.fill_(float(self._mu))        # (set_nominal reset them)
        s.delta.fill_(float(self._delta))

        def on_round(r, s):
            if on_iteration is None:
                return
            if self._batched:
                on_iteration(i, s.state.cpu(), s.Z.clone(), s.U.clone(),
                             s.J_opt.clone())
            else:
                self._mu, self._delta = float(s.mu[0]), float(s.delta[0])
                on_iteration(i, iLQRState(int(s.state[0])), s.Z[0].clone(),
                             s.U[0].clone(), s.J_opt[0].clone())
        self._last_rounds = s.fit(1, kwargs.get("tol", 5e-6),
                                  kwargs.get("max_reg", 1e10), on_round,
                                  graph=self._graph and s.graph_ok())
        accepted = (s.state == int(iLQRState.ACCEPTED)) | \
            (s.state == int(iLQRState.CONVERGED))
        if self._batched or bool(accepted[0]):
            self._export(s)        # (ilqr.py:167-169: stored on accept only)
        else:
            self._mu = float(s.mu[0])
            self._delta = float(s.delta[0])
        return self._states(s)

    def forward(self, z, i, encoding=StateEncoding.DEFAULT, mpc=False,
                ignore_uncertainty=True, u_min=None, u_max=None, **kwargs):
        """ilqr.py:318-362: feedback law, or one MPC re-optimisation step."""
        if not mpc:
            if self._U_nominal is None:
                raise RuntimeError(
                    "You need to either call fit or initialize _U_nominal")
            if self._Z_nominal is None:
                return self._U_nominal[..., i, :]
            if ignore_uncertainty:
                x = decode_mean(z, encoding)
                dx = x - decode_mean(self._Z_nominal[..., i, :], encoding)
                D = x.shape[-1]
                Ki = self._K[..., i, :, :D]
            else:
                dx = z - self._Z_nominal[..., i, :]
                Ki = self._K[..., i, :, :]
            du = (Ki @ dx.unsqueeze(-1)).squeeze(-1)
            return self._U_nominal[..., i, :] + du
        # MPC: _reset_reg, one step() from z, emit U[0], shift      (:356-362)
        U = self._U_nominal
        _native.require_gpu(U)
        Ub, self._batched = _as_batch(U, 2)
        B, N, m = Ub.shape
        z = z.to(dtype=U.dtype, device=U.device)
        if z.dim() == 1:
            z = z.unsqueeze(0).expand(B, -1)
        s = self._get_solver(B, N, z.shape[-1], U.dtype, U.device, encoding,
                             u_min, u_max, mpc_alphas(U.dtype, U.device))
        s.set_nominal(z.contiguous(), Ub.contiguous())
        self._run(s, 1, kwargs.get("tol", 5e-6), kwargs.get("max_reg", 1e10),
                  kwargs.get("on_iteration"))
        self._export(s)
        Un = self._U_nominal
        u = Un[..., 0, :].clone()
        self._U_nominal = torch.cat([Un[..., 1:, :], Un[..., -1:, :]], -2)
        return u


# ---------------------------------------------------------------------------
# module functions with the reference's signatures
# ---------------------------------------------------------------------------

_REC_OWNERS = weakref.WeakValueDictionary()  # rec.data_ptr() -> solver


def forward(z0, U, model, cost, encoding=StateEncoding.DEFAULT,
            batch_rollout=True, model_opts={}, cost_opts={}, u_min=None,
            u_max=None):
    """ilqr.py:393-486: nominal rollout with first / second derivatives.

    Returns (Z, F_z, F_u, L, L_z, L_u, L_zz, L_uz, L_uu); the derivative
    tensors are zero-copy views of the HBM record buffer that `backward`
    streams (so passing them on costs nothing)."""
    _native.require_gpu(U)
    Ub, batched = _as_batch(U.detach(), 2)
    B, N, m = Ub.shape
    z0b = z0.detach().to(dtype=U.dtype, device=U.device)
    if z0b.dim() == 1:
        z0b = z0b.unsqueeze(0).expand(B, -1)
    s = _make_solver(model, cost, encoding, B, N, z0b.shape[-1], U.dtype,
                     U.device, u_min, u_max, None, model_opts, cost_opts)
    s.z0.copy_(z0b)
    s.U.copy_(Ub)
    s.nominal_rollout()
    s.derivs(set_state=False)
    F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu = s.record_views()
    _REC_OWNERS[s.rec.data_ptr()] = s
    out = (s.Z, F_z, F_u, s.L, L_z, L_u, L_zz, L_uz, L_uu)
    if not batched:
        out = tuple(t[0] for t in out)
    out[1]._pddp_owner = s  # keeps the record buffer's owner alive with F_z
    return out


@torch.no_grad()
def Q(F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu, V_z, V_zz):
    """ilqr.py:489-526: derivatives of the Q-function for one step (helper;
    the sweep kernel does this in LDS / registers)."""
    Ft, Gt = F_z.transpose(-1, -2), F_u.transpose(-1, -2)
    Q_z = L_z + (Ft @ V_z.unsqueeze(-1)).squeeze(-1)
    Q_u = L_u + (Gt @ V_z.unsqueeze(-1)).squeeze(-1)
    Q_zz = L_zz + Ft @ V_zz @ F_z
    Q_zz = 0.5 * (Q_zz + Q_zz.transpose(-1, -2))
    Q_uz = L_uz + Gt @ V_zz @ F_z
    Q_uu = L_uu + Gt @ V_zz @ F_u
    Q_uu = 0.5 * (Q_uu + Q_uu.transpose(-1, -2))
    return Q_z, Q_u, Q_zz, Q_uz, Q_uu



def _views_of_record(owner, B, N, n, m, dtype, lay, tensors, batched):
    """True when ALL seven derivative tensors are the untouched views
    `forward()` handed out of `owner.rec` (same storage offset, shape and
    strides): only then may `backward` stream the record buffer as it is.  A
    caller who replaced any of them (say `L_uu + reg I`) gets the pack path."""
    rec = owner.rec
    if rec.shape != (B, N + 1, lay.stride) or rec.dtype != dtype:
        return False
    ref = owner.record_views()
    if not batched:
        ref = tuple(t[0] for t in ref)
    for got, want in zip(tensors, ref):
        if (got.dtype != want.dtype or got.device != want.device
                or got.data_ptr() != want.data_ptr()
                or tuple(got.shape) != tuple(want.shape)
                or tuple(got.stride()) != tuple(want.stride())):
            return False
    return True


_STATUS_MSG = {1: "non-positive definite matrix",
               2: "non-positive definite matrix (Cholesky failed)",
               3: "BoxQP failed"}


@torch.no_grad()
def backward(Z, F_z, F_u, L, L_z, L_u, L_zz, L_uz, L_uu, reg=0.0,
             V_zz_reg=False, u_min=None, u_max=None, U=None, quiet=False,
             return_status=False, kernel_variant=0):
    """ilqr.py:529-674: the backward Riccati sweep -> (k, K).

    Un-batched inputs raise RuntimeError on failure like the reference;
    batched inputs raise unless `return_status=True`, in which case the
    per-trajectory status vector is returned as a third value."""
    _native.require_gpu(F_z)
    F_zb, batched = _as_batch(F_z, 3)
    B, N, n, _ = F_zb.shape
    m = F_u.shape[-1]
    dtype, device = F_z.dtype, F_z.device
    lay = _native.record_layout(n, m)
    owner = _REC_OWNERS.get(F_z.data_ptr())
    zero_copy = owner is not None and _views_of_record(
        owner, B, N, n, m, dtype, lay,
        (F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu), batched)
    bounded = u_min is not None and u_max is not None
    if zero_copy:
        rec = owner.rec
        if bounded and U is not None:
            rec[:, :N, lay.o_U:lay.o_U + m] = U.reshape(B, N, m)
    else:
        rec = torch.empty(B, N + 1, lay.stride, dtype=dtype, device=device)
        c = lambda t, *shape: t.reshape(*shape).contiguous()
        Ub = None if U is None else c(U.to(dtype=dtype, device=device), B, N, m)
        _native.call("pddp_pack_records", dtype, B, N, n, m,
                     _native.ptr(c(F_z, B, N, n, n)),
                     _native.ptr(c(F_u, B, N, n, m)),
                     _native.ptr(c(L_z, B, N + 1, n)),
                     _native.ptr(c(L_u, B, N, m)),
                     _native.ptr(c(L_zz, B, N + 1, n, n)),
                     _native.ptr(c(L_uz, B, N, m, n)),
                     _native.ptr(c(L_uu, B, N, m, m)), _native.ptr(Ub),
                     _native.ptr(rec), _native.stream_handle(device))
    opts = dict(dtype=dtype, device=device)
    umin = None if not bounded else \
        torch.as_tensor(u_min).to(**opts).reshape(m).contiguous()
    umax = None if not bounded else \
        torch.as_tensor(u_max).to(**opts).reshape(m).contiguous()
    if torch.is_tensor(reg) and reg.numel() == B:
        regv = reg.to(dtype=torch.float64, device=device).contiguous()
    else:
        regv = torch.full((B,), float(reg), dtype=torch.float64, device=device)
    gains = torch.zeros(B, N, lay.gain_stride, **opts)
    status = torch.zeros(B, dtype=torch.int32, device=device)
    _native.call("pddp_riccati_backward_variant", dtype, B, N, n, m,
                 _native.ptr(rec), _native.ptr(umin), _native.ptr(umax),
                 _native.ptr(regv),
                 BRANCH_CHOLESKY if V_zz_reg else BRANCH_EIG, None,
                 _native.ptr(gains), _native.ptr(status),
                 _native.stream_handle(device), int(kernel_variant))
    k = gains[..., :m]
    K = gains[..., m:].unflatten(-1, (m, n))
    if not batched:
        k, K = k[0], K[0]
    if return_status:
        return k, K, (status if batched else int(status[0]))
    bad = status.nonzero()
    if bad.numel() > 0:
        code = int(status[bad[0, 0]])
        raise RuntimeError(_STATUS_MSG.get(code, "backward failed"))
    return k, K


@torch.no_grad()
def _control_law(model, Z, U, k, K, alpha, encoding=StateEncoding.DEFAULT,
                 model_opts={}, u_min=None, u_max=None, cost=None,
                 return_cost=False):
    """ilqr.py:677-723: candidate rollouts under u = U + alpha k + K dz.

    Returns Z_new (N+1, [B,] A, n) and U_new (N, [B,] A, m) (time-major, as the
    reference), plus J ([B,] A) when `return_cost` (needs `cost`): the HIP
    line-search kernel evaluates ilqr.py:764-791 in the same pass."""
    _native.require_gpu(U)
    if cost is None:
        cost = getattr(model, "_default_cost", None)
    Ub, batched = _as_batch(U, 2)
    B, N, m = Ub.shape
    n = Z.shape[-1]
    alpha = torch.as_tensor(alpha).flatten()
    if cost is None:  # candidates only: the plugin line search needs a cost
        cost = _ZeroCost()
    s = _make_solver(model, cost, encoding, B, N, n, U.dtype, U.device, u_min,
                     u_max, alpha, model_opts)
    s.Z.copy_(Z.reshape(B, N + 1, n))
    s.U.copy_(Ub)
    s.gains[..., :m] = k.reshape(B, N, m)
    s.gains[..., m:] = K.reshape(B, N, m * n)
    s.line_search(use_status=False)
    A = s.A
    Zn = s.Zc.permute(1, 0, 2, 3)  # (N+1, B, A, n): the reference's time-major
    Un = s.Uc.permute(1, 0, 2, 3)
    J = s.Jc
    if not batched:
        Zn, Un, J = Zn[:, 0], Un[:, 0], J[0]
    if alpha.numel() == 1:
        Zn, Un, J = Zn.squeeze(-2), Un.squeeze(-2), J.squeeze(-1)
    if return_cost:
        return Zn, Un, J
    return Zn, Un


class _ZeroCost(torch.nn.Module):
    """Stand-in when `_control_law` is asked for rollouts only."""

    def forward(self, z, u, i, terminal=False, encoding=None, **kwargs):
        return z.new_zeros(z.shape[:-1])


@torch.no_grad()
def _trajectory_cost(cost, Z, U, encoding=StateEncoding.DEFAULT,
                     cost_opts={}):
    """ilqr.py:764-791 through the plugin interface: one flattened `cost()`
    call over every (time, candidate) row plus the terminal rows.  (The
    controller itself gets these sums from the line-search kernel.)  Passes the
    true time index of each row (the reference's index vector is alpha-major,
    SURVEY appendix A.15)."""
    cost.eval()
    N = U.shape[0]
    batch_shape = Z.shape[1:-1]
    Zr = Z[:-1].reshape(-1, Z.shape[-1])
    Ur = U.reshape(-1, U.shape[-1])
    rows = Zr.shape[0] // N
    I = torch.arange(N, device=Z.device).repeat_interleave(rows)
    L = cost(Zr, Ur, I, terminal=False, encoding=encoding, **cost_opts)
    Ze = Z[-1].reshape(-1, Z.shape[-1])
    lf = cost(Ze, None, torch.full((Ze.shape[0],), N, device=Z.device),
              terminal=True, encoding=encoding, **cost_opts)
    J = L.reshape(N, rows).sum(0) + lf
    return J.reshape(batch_shape) if len(batch_shape) else J.reshape(())
