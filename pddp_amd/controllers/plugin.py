"""Plugin path: arbitrary `DynamicsModel` / `Cost` modules and any
`StateEncoding`, evaluated with PyTorch-ROCm ops on the GPU.

Sample problems run entirely in HIP kernels (solver.ILQRSolver).  Any other
plugin - a user model, an `AggregateCost`, the DEFAULT (Cholesky) encoding -
has no closed-form derivatives, so this module produces what the kernels need
the way the reference does, batched over B trajectories:

  * derivative rollout (ilqr.py:457-473) with the reference's replicate-the-
    input trick (utils/evaluation.py:203-235, 268-280): one autograd pass with
    an identity cotangent gives a Jacobian row per replicated input row;
  * the records are packed by `pddp_pack_records_*`, the backward sweep, the
    accept / regularisation state machine stay the HIP kernels;
  * the line search (ilqr.py:677-723, 764-791) calls the plugin on
    (B * A) rows per time step and writes candidates in the kernels' layout.

Nothing here runs on the CPU; tensors must be CUDA tensors.
"""
import torch

from .. import _native
from ..utils.constraint import clamp


def _augmented_forward():
    from ..examples._common import AugmentedQRCost
    return AugmentedQRCost.forward


class TorchProblem(object):
    """Adapter with the three problem-dependent operations of ILQRSolver."""

    def __init__(self, model, cost, encoding, model_opts=None, cost_opts=None):
        self.model, self.cost, self.encoding = model, cost, encoding
        self.model_opts = dict(model_opts or {})
        self.cost_opts = dict(cost_opts or {})

    # -- ilqr.py:457-468 ------------------------------------------------------
    @torch.no_grad()
    def rollout(self, s):
        self.model.eval()
        Z, U = s.Z, s.U
        if self._bnn_native_ok(s):
            # the nominal is the alpha = 0 "candidate" of zero gains
            opts = dict(dtype=s.dtype, device=s.device)
            Z.zero_()
            Z[:, 0] = s.z0
            Zc = torch.empty(s.B, s.N + 1, 1, s.n, **opts)
            Uc = torch.empty(s.B, s.N, 1, s.m, **opts)
            self._bnn_rollouts(s, 1, torch.zeros(1, **opts),
                               torch.zeros_like(s.gains), Zc, Uc, None, None)
            Z.copy_(Zc[:, :, 0])
            return
        Z[:, 0] = s.z0
        for t in range(s.N):
            u = U[:, t]
            if s.u_min is not None:
                u = clamp(u, s.u_min, s.u_max)
            Z[:, t + 1] = self.model(Z[:, t], u, t, self.encoding,
                                     **self.model_opts)

    # -- ilqr.py:464-473 + evaluation.py:134-288 --------------------------------
    def _cost_derivs(self, z, u, t, terminal):
        """z [B,n], u [B,m] or None -> l [B], l_z, l_u, l_zz, l_uz, l_uu."""
        B, n = z.shape
        m = 0 if terminal else u.shape[1]
        d = n + m
        zu = z if terminal else torch.cat([z, u], -1)
        rep = zu.unsqueeze(1).expand(B, d, d).reshape(B * d, d)
        rep = rep.detach().clone().requires_grad_()
        l_rep = self.cost(rep[:, :n], None if terminal else rep[:, n:], t,
                          terminal=terminal, encoding=self.encoding,
                          identical_inputs=True, **self.cost_opts)
        g, = torch.autograd.grad(l_rep, rep, torch.ones_like(l_rep),
                                 create_graph=True)
        eye = torch.eye(d, dtype=z.dtype, device=z.device).repeat(B, 1)
        H, = torch.autograd.grad(g, rep, eye, allow_unused=True)
        if H is None:
            H = torch.zeros_like(rep)
        H = H.reshape(B, d, d)
        g0 = g.detach().reshape(B, d, d)[:, 0]
        l = l_rep.detach().reshape(B, d)[:, 0]
        l_z, l_zz = g0[:, :n], H[:, :n, :n]
        if terminal:
            return l, l_z, None, l_zz, None, None
        return l, l_z, g0[:, n:], l_zz, H[:, n:, :n], H[:, n:, n:]

    def _dyn_derivs(self, z, u, t):
        """z [B,n], u [B,m] -> F_z [B,n,n], F_u [B,n,m]."""
        B, n = z.shape
        m = u.shape[1]
        zu = torch.cat([z, u], -1)
        rep = zu.unsqueeze(1).expand(B, n, n + m).reshape(B * n, n + m)
        rep = rep.detach().clone().requires_grad_()
        # `identical_inputs` tells a stateful model (the BNN's re-whitening,
        # modules.py:333-348) that all rows are ONE input; with B trajectories
        # in the batch that only holds for B = 1
        zn = self.model(rep[:, :n], rep[:, n:], t, self.encoding,
                        identical_inputs=(B == 1), **self.model_opts)
        eye = torch.eye(n, dtype=z.dtype, device=z.device).repeat(B, 1)
        J, = torch.autograd.grad(zn, rep, eye)
        J = J.reshape(B, n, n + m)
        return J[:, :, :n], J[:, :, n:]

    @torch.no_grad()
    def capture_ok(self, s):
        """True when a whole round is HIP launches and sync-free torch ops
        (native BNN rollout, forward-mode Jacobians, hyper-dual cost
        derivatives): the round can then be captured into hipGraphs
        (ILQRSolver.capture_round)."""
        if self._bnn_native_ok(s) and self._bnn_jvp_ok(s) and \
                self._qr_cost_native_ok(s):
            return True
        # GP plugin: records and line search on pddp_gp_step_*, batched costs
        return self._gp_line_search_ok(s) and self._qr_cost_native_ok(s)

    def derivs(self, s, mask=None, set_state=True, in_graph=False):
        """Fills s.rec, s.L, s.J_opt (and resets s.state) for masked rows.
        `in_graph`: no host synchronisation at all (stream capture): the
        caller knows some nominal is fresh; rows are blended by the mask."""
        if not in_graph and mask is not None and not bool(mask.any()):
            # a round of retries only (ilqr.py:125-139 with a larger mu): every
            # nominal is unchanged, so are its records.  One host sync, against
            # a derivative rollout of the whole batch.
            return
        self.model.eval()
        self.cost.eval()
        B, N, n, m = s.B, s.N, s.n, s.m
        opts = dict(dtype=s.dtype, device=s.device)
        F_z = torch.zeros(B, N, n, n, **opts)
        F_u = torch.zeros(B, N, n, m, **opts)
        L = torch.zeros(B, N + 1, **opts)
        L_z = torch.zeros(B, N + 1, n, **opts)
        L_u = torch.zeros(B, N, m, **opts)
        L_zz = torch.zeros(B, N + 1, n, n, **opts)
        L_uz = torch.zeros(B, N, m, n, **opts)
        L_uu = torch.zeros(B, N, m, m, **opts)
        with torch.no_grad():  # (the network's kernel refuses when autograd
            # could be asked to differentiate through it)
            native_dyn = self._bnn_native_ok(s, need_cost=False) and \
                self._bnn_jvp_ok(s)
            gp_dyn = not native_dyn and self._gp_native_ok(s)
        native_cost = self._qr_cost_native_ok(s)
        # which code produced the records (tests assert on it)
        if native_dyn:
            # (rows of the trajectories whose nominal stands are skipped)
            self._dyn_derivs_bnn(s, F_z, F_u, mask)
        elif gp_dyn:
            # GP plugin: every (trajectory, step) row of the nominal in ONE
            # launch of the moment-matched step with its Jacobian
            # (rows of the trajectories whose nominal stands are skipped:
            # their records are kept below)
            self._dyn_derivs_gp(s, F_z, F_u, mask)
            native_dyn = True
        self.last_derivs_path = {"dynamics": "hip" if native_dyn else "autograd",
                                 "cost": "hip" if native_cost else "autograd"}
        if native_cost:
            self._cost_derivs_qr(s, L, L_z, L_u, L_zz, L_uz, L_uu)
        with torch.enable_grad():
            for t in range(N):
                if native_dyn and native_cost:
                    break
                z = s.Z[:, t].detach()
                u = s.U[:, t].detach()
                if s.u_min is not None:  # derivatives AT the clamped action
                    u = clamp(u, s.u_min, s.u_max)
                if not native_cost:
                    (L[:, t], L_z[:, t], L_u[:, t], L_zz[:, t], L_uz[:, t],
                     L_uu[:, t]) = self._cost_derivs(z, u, t, False)
                if not native_dyn:
                    F_z[:, t], F_u[:, t] = self._dyn_derivs(z, u, t)
            # terminal cost, evaluated with the stale index N-1 (ilqr.py:471-473)
            if not native_cost:
                L[:, N], L_z[:, N], _, L_zz[:, N], _, _ = self._cost_derivs(
                    s.Z[:, N].detach(), None, N - 1, True)
        rec = torch.empty_like(s.rec)
        p = _native.ptr
        _native.call("pddp_pack_records", s.dtype, B, N, n, m, p(F_z), p(F_u),
                     p(L_z), p(L_u), p(L_zz), p(L_uz), p(L_uu),
                     p(s.U.contiguous()), p(rec), s._s())
        J = torch.empty(B, **opts)  # L.sum() (ilqr.py:484), in t order
        _native.call("pddp_sum_stage_costs", s.dtype, B, N + 1, p(L), p(J),
                     s._s())
        if mask is None:
            s.rec.copy_(rec)
            s.L.copy_(L)
            s.J_opt.copy_(J)
            if set_state:
                s.state.zero_()
        else:  # (torch.where: no nonzero(), so no host sync)
            sel = mask.bool()
            s.rec.copy_(torch.where(sel.view(B, 1, 1), rec, s.rec))
            s.L.copy_(torch.where(sel.view(B, 1), L, s.L))
            s.J_opt.copy_(torch.where(sel, J, s.J_opt))
            if set_state:
                s.state.copy_(torch.where(sel, torch.zeros_like(s.state),
                                          s.state))

    # -- GP plugin: csrc/gp_step.hip ----------------------------------------------
    def _gp_native_ok(self, s):
        mo = self.model
        if not (hasattr(mo, "native_step") and hasattr(mo, "native_ok")):
            return False
        if not getattr(self, "use_native_gp", True) or self.model_opts:
            return False
        return bool(mo.native_ok(s.Z[:, 0], self.encoding, jacobian=True))

    @torch.no_grad()
    def _dyn_derivs_gp(self, s, F_z, F_u, mask=None):
        """F_z [B N n n], F_u [B N n m] of the nominal: d z' / d (z, u) at the
        clamped actions (ilqr.py:443-470), by `pddp_gp_step_*`; with `mask`
        [B] only for its trajectories (`pddp_gp_step_masked_*`)."""
        B, N, n, m = s.B, s.N, s.n, s.m
        z = s.Z[:, :N].reshape(B * N, n)
        u = s.U
        if s.u_min is not None:
            u = clamp(u, s.u_min, s.u_max)
        rm = None
        if mask is not None:
            rm = mask.to(torch.uint8).contiguous()
        self.model.native_step(z, u.reshape(B * N, m), self.encoding,
                               jacobian=True, Fz=F_z, Fu=F_u, row_mask=rm,
                               rows_per_mask=N)

    def _gp_line_search_ok(self, s):
        """The line search as N launches of the GP step plus ONE batched
        evaluation of the stage costs: needs the sample problems' QR cost on
        the model's own angle-augmented state."""
        from ..examples._common import AugmentedQRCost
        co, mo = self.cost, self.model
        if not self._gp_native_ok(s) or self.cost_opts:
            return False
        if not isinstance(co, AugmentedQRCost) or co.model_class is None:
            return False
        mc = co.model_class
        return (list(mc.angular_indices) == list(mo.angular_indices) and
                list(mc.non_angular_indices) == list(mo.non_angular_indices)
                and mc.state_size == mo.state_size)

    @torch.no_grad()
    def _qr_costs_batched(self, s, Zc, Uc, chunk=8):
        """Sum over the horizon of the QR cost of every candidate rollout:
        Zc [B, N+1, A, n], Uc [B, N, A, m] -> [B, A].  The expectation
        E[(x~ - g)^T Q (x~ - g)] + u^T R u on the angle-augmented Gaussian state
        (costs/quadratic.py:60-99 after utils/angular.py:47-84), taken from
        the moments themselves: the reference's encode -> decode round trip of
        the augmented covariance (a Cholesky factorisation per time step) is
        the identity up to its 1e-12 jitter."""
        from ..utils.angular import (augment_moments, augment_moments_var,
                                     augment_state)
        from ..utils.encoding import (StateEncoding, decode_covar,
                                      decode_covar_sqrt, decode_mean,
                                      decode_var)
        co, enc = self.cost, self.encoding
        mc = co.model_class
        ai, ni, D = list(mc.angular_indices), list(mc.non_angular_indices), \
            mc.state_size
        B, N1, A, n = Zc.shape
        J = torch.zeros(B, A, dtype=Zc.dtype, device=Zc.device)
        for t0 in range(0, N1, chunk):
            t1 = min(N1, t0 + chunk)
            z = Zc[:, t0:t1]
            mean = decode_mean(z, enc, D)
            if enc == StateEncoding.IGNORE_UNCERTAINTY:
                M_, spread = augment_state(mean, ai, ni), None
            elif enc in (StateEncoding.FULL_COVARIANCE_MATRIX,
                         StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
                if enc == StateEncoding.UPPER_TRIANGULAR_CHOLESKY:
                    # U^T U as D outer products (a batched 6 x 6 GEMM over
                    # 80 000 rows takes 0.6 ms per chunk)
                    Uf = decode_covar_sqrt(z, enc, D)
                    Sx = Uf[..., 0, :, None] * Uf[..., 0, None, :]
                    for r in range(1, D):
                        Sx = Sx + Uf[..., r, :, None] * Uf[..., r, None, :]
                else:
                    Sx = decode_covar(z, enc, D)
                M_, C_ = augment_moments(mean, Sx, ai, ni)
                spread = None
            else:
                M_, spread = augment_moments_var(mean, decode_var(z, enc, D),
                                                 ai, ni)
                C_ = None
            dx = M_ - co.x_goal
            # (the last point of the last chunk is the terminal state: Q_term)
            parts = [(co.Q, 0, t1 - t0)]
            if t1 == N1:
                parts = [(co.Q, 0, t1 - t0 - 1), (co.Q_term, t1 - t0 - 1, t1 - t0)]
            for Q, lo, hi in parts:
                if lo >= hi:
                    continue
                # E[(x~ - g)(x~ - g)^T] = d d^T + C, contracted with Q
                # element-wise (a [rows, 8] x [8, 8] product goes to a GEMM
                # kernel that takes 0.6 ms per chunk at these shapes)
                d = dx[:, lo:hi]
                second = d.unsqueeze(-1) * d.unsqueeze(-2)
                if enc != StateEncoding.IGNORE_UNCERTAINTY:
                    if spread is None:
                        second = second + C_[:, lo:hi].transpose(-1, -2)
                    else:
                        second = second + torch.diag_embed(spread[:, lo:hi])
                J += (second * Q).sum((-2, -1)).sum(1)
        du = Uc - co.u_goal
        J += ((du @ co.R) * du).sum(-1).sum(1)
        return J

    use_gp_rollout = True  # pddp_gp_rollout_* (False: the per-step torch form)

    def cost_generation(self):
        """Identity and version of the cost's tensors whose converted copies
        captured launches read (`_line_search_gp`): a captured round is stale
        when this changes (ILQRSolver._graphs_fresh)."""
        co = self.cost
        ts = [getattr(co, k, None) for k in ("Q", "Q_term", "R", "x_goal",
                                             "u_goal")]
        if not hasattr(self, "_gp_cost_cache") or any(
                not torch.is_tensor(t) for t in ts):
            return None
        return tuple((t.data_ptr(), t._version) for t in ts)

    @torch.no_grad()
    def _line_search_gp(self, s, active=None, use_status=True):
        """ilqr.py:677-723 + :764-791 for every (trajectory, step size) as a
        device rollout: control law, clamp, moment-matched GP step and stage
        cost in the step's own kernel, N + 1 launches with nothing between
        them (`pddp_gp_rollout_*`, csrc/gp_step.hip)."""
        mo, co = self.model, self.cost
        import os
        if os.environ.get("PDDP_NO_GP_ROLLOUT"):
            return self._line_search_gp_torch(s)
        if not (self.use_gp_rollout and mo.state_size + 0 <= 6 and
                len(mo.non_angular_indices) + 2 * len(mo.angular_indices) <= 8
                and s.m <= 4):
            return self._line_search_gp_torch(s)
        import ctypes
        g = mo._native_model(s.dtype, s.device, self.encoding)
        key = (s.dtype, str(s.device)) + tuple(
            (t.data_ptr(), t._version)
            for t in (co.Q, co.Q_term, co.R, co.x_goal, co.u_goal))
        cc = getattr(self, "_gp_cost_cache", None)
        src = (co.Q, co.Q_term, co.R, co.x_goal, co.u_goal)
        if cc is not None and cc[0] != key and cc[0][:2] == key[:2] and all(
                d.shape == t.shape for d, t in zip(cc[1], src)):
            # new values INTO the converted copies: launches baked into a
            # hipGraph hold their addresses (a replaced tuple would free them
            # under the graph); the graphs themselves are dropped by
            # ILQRSolver._graphs_fresh through cost_generation()
            for d, t in zip(cc[1], src):
                d.copy_(t.detach())
            cc = self._gp_cost_cache = (key, cc[1])
        elif cc is None or cc[0] != key:
            conv = lambda t: t.detach().to(dtype=s.dtype, device=s.device,
                                           copy=True).contiguous()
            cc = self._gp_cost_cache = (key, tuple(conv(t) for t in src))
        Q, Qt, R, xg, ug = cc[1]
        p = _native.ptr
        r = _native.GpRollout()
        r.B, r.N, r.A = s.B, s.N, s.A
        for name, t in (("Z", s.Z), ("U", s.U), ("gains", s.gains),
                        ("alphas", s.alphas), ("u_min", s.u_min),
                        ("u_max", s.u_max), ("active", active),
                        ("bwd_status", s.bwd_status if use_status else None),
                        ("Zc", s.Zc), ("Uc", s.Uc), ("Jc", s.Jc), ("Q", Q),
                        ("Q_term", Qt), ("R", R), ("x_goal", xg),
                        ("u_goal", ug)):
            setattr(r, name, p(t))
        with torch.cuda.device(s.device):
            _native.call("pddp_gp_rollout", s.dtype, ctypes.byref(g),
                         ctypes.byref(r), _native.stream_handle(s.device))

    @torch.no_grad()
    def _line_search_gp_torch(self, s):
        """The same line search step by step in torch ops around
        `pddp_gp_step_*` with one batched evaluation of the costs: the checker
        of the rollout kernel (tests/test_gp.py)."""
        B, N, n, m, A = s.B, s.N, s.n, s.m, s.A
        k, K = s.gain_views()
        alpha = s.alphas.view(1, A, 1)
        z = s.Z[:, 0].unsqueeze(1).expand(B, A, n).contiguous()
        s.Zc[:, 0] = z
        for t in range(N):
            dz = z - s.Z[:, t].unsqueeze(1)
            u = s.U[:, t].unsqueeze(1) + alpha * k[:, t].unsqueeze(1) + \
                dz @ K[:, t].transpose(-1, -2)
            if s.u_min is not None:
                u = clamp(u, s.u_min, s.u_max)
            z = self.model.native_step(z.reshape(B * A, n),
                                       u.reshape(B * A, m),
                                       self.encoding).reshape(B, A, n)
            s.Uc[:, t] = u
            s.Zc[:, t + 1] = z
        s.Jc.copy_(self._qr_costs_batched(s, s.Zc, s.Uc))

    # -- fused BNN rollout: csrc/bnn_rollout.hip + csrc/bnn_mlp.hip -------------
    def _qr_cost_native_ok(self, s, co=None):
        """QR cost on the angle-augmented state, DEFAULT encoding, f32: value,
        gradient and Hessian of every (trajectory, step) in one launch
        (include/pddp_hip.h pddp_qr_cost_derivs_f32).  An AggregateCost
        (costs/base.py: op(first, second) built by the arithmetic overloads)
        qualifies when every leaf does and every op is one of + - * / ** (the
        last with a constant exponent): its derivatives follow from the
        leaves' by the product / quotient / power rules
        (`_cost_derivs_tree`), no autograd pass per time step."""
        from ..costs.base import AggregateCost
        from ..costs.quadratic import QRCost
        from ..utils.encoding import StateEncoding
        top = co is None
        co = self.cost if co is None else co
        if isinstance(co, AggregateCost):
            if not getattr(self, "use_native_cost", True) or self.cost_opts:
                return False
            if co.op not in (torch.add, torch.sub, torch.mul, torch.div,
                             torch.pow):
                return False
            kids = [c for c in (co.first, co.second)
                    if isinstance(c, torch.nn.Module)]
            if not kids or (co.op is torch.pow and
                            isinstance(co.second, torch.nn.Module)):
                return False
            consts = [c for c in (co.first, co.second)
                      if not isinstance(c, torch.nn.Module)]
            if any(isinstance(c, torch.Tensor) and c.numel() != 1
                   for c in consts):
                return False
            return all(self._qr_cost_native_ok(s, c) for c in kids)
        del top
        mc = getattr(co, "model_class", None)
        if not getattr(self, "use_native_cost", True) or self.cost_opts:
            return False
        if s.dtype not in (torch.float32, torch.float64) or \
                type(co).forward is not _augmented_forward() or mc is None:
            return False
        if int(self.encoding) != int(StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
            return False
        D = mc.state_size
        return (D in (2, 4, 6) and s.m <= 2 and len(mc.angular_indices) <= 2
                and s.n == D + D * (D + 1) // 2
                and isinstance(co, QRCost))

    @torch.no_grad()
    def _cost_derivs_tree(self, s, co):
        """(L [B,N+1], L_z, L_u, L_zz, L_uz, L_uu) of a cost expression tree:
        QR leaves through the HIP kernel, inner nodes by calculus on the
        batched tensors.  With x = (z, u), c = a op b:
            +, -:  linear
            *:     c' = a' b + a b',  c'' = a'' b + a b'' + a' b'^T + b' a'^T
            /:     c = a r, r = 1 / b: r' = -b' / b^2,
                   r'' = -b'' / b^2 + 2 b' b'^T / b^3
            ** e:  c' = e a^(e-1) a',  c'' = e a^(e-1) a'' + e (e-1) a^(e-2) a' a'^T
        (blocks zz, uz, uu; the terminal step has the z blocks only)."""
        from ..costs.base import AggregateCost
        B, N, n, m = s.B, s.N, s.n, s.m
        opts = dict(dtype=s.dtype, device=s.device)

        def zeros():
            return (torch.zeros(B, N + 1, **opts),
                    torch.zeros(B, N + 1, n, **opts),
                    torch.zeros(B, N, m, **opts),
                    torch.zeros(B, N + 1, n, n, **opts),
                    torch.zeros(B, N, m, n, **opts),
                    torch.zeros(B, N, m, m, **opts))
        if not isinstance(co, torch.nn.Module):  # a constant
            out = zeros()
            out[0].fill_(float(co))
            return out
        if not isinstance(co, AggregateCost):
            out = zeros()
            self._cost_derivs_qr(s, *out, co=co)
            return out
        a = self._cost_derivs_tree(s, co.first)
        op = co.op

        def scale(t, f):
            """t * f with f [B, N+1] (or its first N steps) broadcast."""
            ff = f[:, :t.shape[1]]
            return t * ff.reshape(B, t.shape[1], *([1] * (t.dim() - 2)))

        def outer(p_, q_):
            """Blocks of p q^T + q p^T for gradients p = (pz, pu), q."""
            (pz, pu), (qz, qu) = p_, q_
            zz = pz.unsqueeze(-1) * qz.unsqueeze(-2)
            zz = zz + zz.transpose(-1, -2)
            uz = pu.unsqueeze(-1) * qz[:, :N].unsqueeze(-2) + \
                qu.unsqueeze(-1) * pz[:, :N].unsqueeze(-2)
            uu = pu.unsqueeze(-1) * qu.unsqueeze(-2)
            uu = uu + uu.transpose(-1, -2)
            return zz, uz, uu

        def mul(a, b):
            (la, az, au, azz, auz, auu), (lb, bz, bu, bzz, buz, buu) = a, b
            zz, uz, uu = outer((az, au), (bz, bu))
            return (la * lb, scale(az, lb) + scale(bz, la),
                    scale(au, lb) + scale(bu, la),
                    scale(azz, lb) + scale(bzz, la) + zz,
                    scale(auz, lb) + scale(buz, la) + uz,
                    scale(auu, lb) + scale(buu, la) + uu)

        def power(a, e):
            la, az, au, azz, auz, auu = a
            f1 = e * la ** (e - 1.0)
            f2 = 0.5 * e * (e - 1.0) * la ** (e - 2.0)  # (outer() doubles)
            zz, uz, uu = outer((az, au), (az, au))
            return (la ** e, scale(az, f1), scale(au, f1),
                    scale(azz, f1) + scale(zz, f2),
                    scale(auz, f1) + scale(uz, f2),
                    scale(auu, f1) + scale(uu, f2))
        if op is torch.pow:
            return power(a, float(co.second))
        b = self._cost_derivs_tree(s, co.second)
        if op is torch.add:
            return tuple(x + y for x, y in zip(a, b))
        if op is torch.sub:
            return tuple(x - y for x, y in zip(a, b))
        if op is torch.mul:
            return mul(a, b)
        return mul(a, power(b, -1.0))  # torch.div

    @torch.no_grad()
    def _cost_derivs_qr(self, s, L, L_z, L_u, L_zz, L_uz, L_uu, co=None):
        import ctypes
        from ..costs.base import AggregateCost
        co = self.cost if co is None else co
        if isinstance(co, AggregateCost):
            for dst, src in zip((L, L_z, L_u, L_zz, L_uz, L_uu),
                                self._cost_derivs_tree(s, co)):
                dst.copy_(src)
            return
        mc = co.model_class
        ang = [int(i) for i in mc.angular_indices]
        non = [int(i) for i in mc.non_angular_indices]
        na, m = len(non) + 2 * len(ang), s.m
        opts = dict(dtype=s.dtype, device=s.device)
        vec = lambda t, k: torch.as_tensor(t).detach().to(**opts).expand(
            k).contiguous()
        mat = lambda t: t.detach().to(**opts).contiguous()
        keep = [mat(co.Q), mat(co.Q_term), mat(co.R), vec(co.x_goal, na),
                vec(co.u_goal, m), s.Z.contiguous(), s.U.contiguous()]
        st = _native.QrCost()
        st.B, st.N, st.D, st.m = s.B, s.N, mc.state_size, m
        st.n_ang, st.n_non = len(ang), len(non)
        for i, v in enumerate(ang):
            st.ang[i] = v
        for i, v in enumerate(non):
            st.non[i] = v
        p = _native.ptr
        for name, t in (("Q", keep[0]), ("Q_term", keep[1]), ("R", keep[2]),
                        ("x_goal", keep[3]), ("u_goal", keep[4]),
                        ("Z", keep[5]), ("U", keep[6]), ("u_min", s.u_min),
                        ("u_max", s.u_max), ("L", L), ("L_z", L_z),
                        ("L_u", L_u), ("L_zz", L_zz), ("L_uz", L_uz),
                        ("L_uu", L_uu)):
            setattr(st, name, p(t))
        _native.call("pddp_qr_cost_derivs", s.dtype, ctypes.byref(st),
                     _native.stream_handle(s.device))

    def _bnn_jvp_ok(self, s):
        """The forward-mode kernels cover D <= 6 with D + m <= 7 network
        tangent rows and up to 31 directions of (z | u) (include/pddp_hip.h
        pddp_bnn_jvp_group)."""
        mo = self.model
        return (getattr(self, "use_native_bnn_jvp", True) and
                _native.lib().pddp_bnn_jvp_group(mo.state_size, s.m) != 0)

    @torch.no_grad()
    def _dyn_derivs_bnn(self, s, F_z, F_u, mask=None):
        """F_z, F_u of the whole nominal in forward mode: per time step one
        feature launch, the fused network in JVP mode on B P 8 rows, one
        moment launch (csrc/bnn_jvp.hip, csrc/bnn_mlp.hip) - instead of
        autograd over n replicated inputs (utils/evaluation.py:203-235)."""
        import ctypes
        from ..utils.encoding import decode_covar_sqrt, decode_mean
        mo = self.model
        B, N, n, m = s.B, s.N, s.n, s.m
        D, P = mo.state_size, mo.n_particles
        ang, non = mo.angular_indices_, mo.non_angular_indices_
        in_dim = len(non) + 2 * len(ang) + m
        opts = dict(dtype=s.dtype, device=s.device)
        vec = lambda t, k: torch.as_tensor(t).detach().to(**opts).expand(
            k).contiguous()
        z0 = s.Z[:, 0]
        if 0 not in mo.eps_in:
            e = torch.randn(P, D, **opts)
            mo.eps_in[0] = (e - e.mean(0)) / e.std(0)
        Xp = (decode_mean(z0, self.encoding).unsqueeze(-2) +
              mo.eps_in[0] @ decode_covar_sqrt(z0, self.encoding)).contiguous()
        Xn = torch.empty_like(Xp)
        eps = torch.empty_like(Xp)
        G = 8  # network rows per (state, particle): input + D + m tangents
        F = torch.empty(B * P * G, in_dim, **opts)
        keep = [vec(mo.X_mean, in_dim), vec(mo.X_std_inv, in_dim),
                vec(mo.dX_mean, D), vec(mo.dX_std, D), s.Z.contiguous(),
                s.U.contiguous()]
        st = _native.BnnJvp()
        st.B, st.P, st.D, st.m, st.N = B, P, D, m, N
        st.n_ang, st.n_non = len(ang), len(non)
        for i, v in enumerate(ang):
            st.ang[i] = v
        for i, v in enumerate(non):
            st.non[i] = v
        ups = self._predicted_std()
        out_rows = 2 * D if ups else D
        st.in_dim, st.out_dim = in_dim, out_rows
        st.independent_noise = int(bool(
            self.model_opts.get("independent_noise", False)))
        eps_keep = [self._eps_out(i, P, D, opts) for i in range(N)] if ups \
            else None
        p = _native.ptr
        for name, t in (("X_mean", keep[0]), ("X_std_inv", keep[1]),
                        ("dX_mean", keep[2]), ("dX_std", keep[3]),
                        ("Z", keep[4]), ("U", keep[5]), ("u_min", s.u_min),
                        ("u_max", s.u_max), ("eps", eps), ("F", F),
                        ("F_z", F_z), ("F_u", F_u)):
            setattr(st, name, p(t))
        # with a mask: only its trajectories, their network rows packed to the
        # front and the network on that many rows (a device count; sync-free)
        live_rows = slot = None
        if mask is not None:
            run = mask != 0
            rank = torch.cumsum(run.to(torch.int32), 0, dtype=torch.int32)
            slot = torch.where(run, rank - 1,
                               torch.full_like(rank, -1)).contiguous()
            live_rows = (rank[-1:] * (P * G)).to(torch.int32)
            st.slot = p(slot)
        lib, stream = _native.lib(), _native.stream_handle(s.device)
        for t in range(N):
            st.t = t
            st.Xp, st.Xp_next = p(Xp), p(Xn)
            _native.call("pddp_bnn_jvp_features", s.dtype, ctypes.byref(st),
                         stream)
            Y = mo.model._jvp_native(F, P, out_rows, G, live=1 + D + s.m,
                                     live_rows=live_rows)
            st.net_out = p(Y)
            st.eps_out = p(eps_keep[t]) if ups else None
            _native.call("pddp_bnn_jvp_moments", s.dtype, ctypes.byref(st),
                         stream)
            Xp, Xn = Xn, Xp
        mo.output = {}  # the particle caches of a torch-path rollout: stale

    def _bnn_native_ok(self, s, need_cost=True):
        """True when the line search can run as N + 1 moment-step launches
        with the fused network kernel in between (include/pddp_hip.h:
        pddp_bnn_moment_step_f32, pddp_bnn_mlp_f32) instead of ~150 torch
        launches per time step."""
        from ..costs.quadratic import QRCost
        from ..utils.encoding import StateEncoding
        mo, co = self.model, self.cost
        if not getattr(self, "use_native_bnn", True):
            return False
        if s.dtype not in (torch.float32, torch.float64) or self.cost_opts:
            return False
        if int(self.encoding) != int(StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
            return False
        if not (hasattr(mo, "eps_in") and hasattr(mo, "n_particles")
                and hasattr(mo, "angular_indices_")):
            return False
        opts = dict(infer_noise_variables=True,
                    sample_input_distribution=True, resample=False)
        free = ("use_predicted_std", "independent_noise")  # either value
        for k, v in self.model_opts.items():
            if k in free:
                continue
            if k not in opts or bool(v) != opts[k]:
                return False
        mc = getattr(co, "model_class", None)
        if need_cost:
            if not isinstance(co, QRCost) or mc is None:
                return False
            if tuple(int(i) for i in mc.angular_indices) != \
                    mo.angular_indices_ or \
                    tuple(int(i) for i in mc.non_angular_indices) != \
                    mo.non_angular_indices_:
                return False
        D, m, P = mo.state_size, mo.action_size, mo.n_particles
        if D > 8 or m > 2 or P > 128 or P < 2 or len(mo.angular_indices_) > 2:
            return False
        probe = torch.empty(1, P, mo.model.hidden[0].in_features,
                            dtype=s.dtype, device=s.device)
        return mo.model._native_ok(probe, False)

    def _predicted_std(self):
        return bool(self.model_opts.get("use_predicted_std", False))

    def _eps_out(self, i, P, D, opts):
        """Standardised normals of time index i for the predicted std, drawn
        and cached exactly as the model's own forward does
        (models/bnn.py forward; modules.py:242-252)."""
        mo = self.model
        if i not in mo.eps_out:
            eps = torch.randn(P, D, **opts)
            mo.eps_out[i] = (eps - eps.mean(0)) / eps.std(0)
        return mo.eps_out[i].contiguous()

    @torch.no_grad()
    def _line_search_bnn(self, s, active, use_status):
        # (costs straight into s.Jc: rows of skipped trajectories stay as they
        # are, like their candidates)
        self._bnn_rollouts(s, s.A, s.alphas, s.gains, s.Zc, s.Uc, active,
                           s.bwd_status if use_status else None,
                           Jc_out=s.Jc.view(-1))

    @torch.no_grad()
    def _bnn_rollouts(self, s, A, alphas, gains, Zc, Uc, active, status,
                      Jc_out=None):
        """A moment-matched rollouts per trajectory under the control law
        u = clamp(U + alpha k + K (z - Z)): N + 1 moment-step launches with the
        fused network kernel in between.  Returns the costs [B A]."""
        import ctypes
        from ..utils.encoding import decode_covar_sqrt, decode_mean
        mo, co = self.model, self.cost
        B, N, n, m = s.B, s.N, s.n, s.m
        D, P = mo.state_size, mo.n_particles
        ang, non = mo.angular_indices_, mo.non_angular_indices_
        na = len(non) + 2 * len(ang)
        ups = self._predicted_std()
        # (without the predicted std the log-std rows of fc_out are unused)
        in_dim, out_dim = na + m, (2 * D if ups else D)
        opts = dict(dtype=s.dtype, device=s.device)
        vec = lambda t, k: torch.as_tensor(t).detach().to(**opts).expand(
            k).contiguous()
        mat = lambda t: t.detach().to(**opts).contiguous()
        # particles of step 0: mean + eps L, cached standardised normals
        # (modules.py:312-330; the same draw the model itself would make)
        z0 = s.Z[:, 0]
        mean0 = decode_mean(z0, self.encoding)
        if 0 not in mo.eps_in:
            e = torch.randn(P, D, **opts)
            mo.eps_in[0] = (e - e.mean(0)) / e.std(0)
        X0 = mean0.unsqueeze(-2) + mo.eps_in[0] @ decode_covar_sqrt(
            z0, self.encoding)
        Xp = X0.unsqueeze(1).expand(B, A, P, D).contiguous()
        F = torch.zeros(B * A, P, in_dim, **opts)
        J = torch.zeros(B * A, **opts)
        Jc = torch.zeros(B * A, **opts) if Jc_out is None else Jc_out
        keep = [vec(mo.X_mean, in_dim), vec(mo.X_std_inv, in_dim),
                vec(mo.dX_mean, D), vec(mo.dX_std, D), mat(co.Q),
                mat(co.Q_term), mat(co.R), vec(co.x_goal, na),
                vec(co.u_goal, m)]
        st = _native.BnnStep()
        st.B, st.A, st.P, st.D, st.m, st.N = B, A, P, D, m, N
        st.n_ang, st.n_non = len(ang), len(non)
        for i, v in enumerate(ang):
            st.ang[i] = v
        for i, v in enumerate(non):
            st.non[i] = v
        st.in_dim, st.out_dim = in_dim, out_dim
        p = _native.ptr
        for name, t in (("Z", s.Z), ("U", s.U), ("gains", gains),
                        ("alphas", alphas), ("u_min", s.u_min),
                        ("u_max", s.u_max), ("active", active),
                        ("bwd_status", status),
                        ("X_mean", keep[0]), ("X_std_inv", keep[1]),
                        ("dX_mean", keep[2]), ("dX_std", keep[3]),
                        ("Q", keep[4]), ("Q_term", keep[5]), ("R", keep[6]),
                        ("x_goal", keep[7]), ("u_goal", keep[8]), ("Xp", Xp),
                        ("F", F), ("Zc", Zc), ("Uc", Uc), ("J", J),
                        ("Jc", Jc)):
            setattr(st, name, p(t))
        # The live trajectories' candidates are packed to the front of the
        # network's rows (slot = rank among the live ones) and the network runs
        # on those rows only - a count it reads on the device: a round of
        # retries with three of 256 restarts alive costs 3 / 256 of the
        # network time (sync-free: capturable)
        live_rows = slot = None
        if active is not None or status is not None:
            alive = torch.ones(B, dtype=torch.bool, device=s.device)
            if active is not None:
                alive &= active != 0
            if status is not None:
                alive &= status == 0
            rank = torch.cumsum(alive.to(torch.int32), 0, dtype=torch.int32)
            slot = (rank - 1).contiguous()
            live_rows = (rank[-1:] * (A * P)).to(torch.int32)
            st.slot = p(slot)
        lib, stream = _native.lib(), _native.stream_handle(s.device)
        out = None
        eps_keep = [self._eps_out(i, P, D, opts) for i in range(N)] if ups \
            else None
        for t in range(N + 1):
            st.t = t
            st.net_out = p(out)
            st.eps_out = p(eps_keep[t - 1]) if ups and t > 0 else None
            _native.call("pddp_bnn_moment_step", s.dtype, ctypes.byref(st),
                         stream)
            if t < N:
                out = mo.model._forward_native(F, out_dim, live_rows=live_rows)
        mo.output = {}  # the particle caches of a torch-path rollout: stale
        return Jc

    # -- ilqr.py:677-723, 764-791 ---------------------------------------------
    @torch.no_grad()
    def line_search(self, s, active=None, use_status=True):
        self.model.eval()
        self.cost.eval()
        if self._bnn_native_ok(s):
            return self._line_search_bnn(s, active, use_status)
        if self._gp_line_search_ok(s):
            return self._line_search_gp(s, active, use_status)
        B, N, n, m, A = s.B, s.N, s.n, s.m, s.A
        k, K = s.gain_views()
        alpha = s.alphas.view(1, A, 1)
        z = s.Z[:, 0].unsqueeze(1).expand(B, A, n).contiguous()
        J = torch.zeros(B, A, dtype=s.dtype, device=s.device)
        s.Zc[:, 0] = z
        for t in range(N):
            dz = z - s.Z[:, t].unsqueeze(1)
            du = alpha * k[:, t].unsqueeze(1) + dz @ K[:, t].transpose(-1, -2)
            u = s.U[:, t].unsqueeze(1) + du
            if s.u_min is not None:
                u = clamp(u, s.u_min, s.u_max)
            zf, uf = z.reshape(B * A, n), u.reshape(B * A, m)
            J += self.cost(zf, uf, t, terminal=False, encoding=self.encoding,
                           **self.cost_opts).reshape(B, A)
            z = self.model(zf, uf, t, self.encoding,
                           **self.model_opts).reshape(B, A, n)
            s.Uc[:, t] = u
            s.Zc[:, t + 1] = z
        J += self.cost(z.reshape(B * A, n), None, N, terminal=True,
                       encoding=self.encoding, **self.cost_opts).reshape(B, A)
        s.Jc.copy_(J)
