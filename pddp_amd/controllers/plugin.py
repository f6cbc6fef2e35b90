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
        zn = self.model(rep[:, :n], rep[:, n:], t, self.encoding,
                        identical_inputs=True, **self.model_opts)
        eye = torch.eye(n, dtype=z.dtype, device=z.device).repeat(B, 1)
        J, = torch.autograd.grad(zn, rep, eye)
        J = J.reshape(B, n, n + m)
        return J[:, :, :n], J[:, :, n:]

    def derivs(self, s, mask=None, set_state=True):
        """Fills s.rec, s.L, s.J_opt (and resets s.state) for masked rows."""
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
        with torch.enable_grad():
            for t in range(N):
                z = s.Z[:, t].detach()
                u = s.U[:, t].detach()
                if s.u_min is not None:  # derivatives AT the clamped action
                    u = clamp(u, s.u_min, s.u_max)
                (L[:, t], L_z[:, t], L_u[:, t], L_zz[:, t], L_uz[:, t],
                 L_uu[:, t]) = self._cost_derivs(z, u, t, False)
                F_z[:, t], F_u[:, t] = self._dyn_derivs(z, u, t)
            # terminal cost, evaluated with the stale index N-1 (ilqr.py:471-473)
            L[:, N], L_z[:, N], _, L_zz[:, N], _, _ = self._cost_derivs(
                s.Z[:, N].detach(), None, N - 1, True)
        rec = torch.empty_like(s.rec)
        p = _native.ptr
        _native.call("pddp_pack_records", s.dtype, B, N, n, m, p(F_z), p(F_u),
                     p(L_z), p(L_u), p(L_zz), p(L_uz), p(L_uu),
                     p(s.U.contiguous()), p(rec), s._s())
        J = L.sum(-1)
        if mask is None:
            s.rec.copy_(rec)
            s.L.copy_(L)
            s.J_opt.copy_(J)
            if set_state:
                s.state.zero_()
        else:
            sel = mask.bool()
            s.rec[sel] = rec[sel]
            s.L[sel] = L[sel]
            s.J_opt[sel] = J[sel]
            if set_state:
                s.state[sel] = 0

    # -- ilqr.py:677-723, 764-791 ---------------------------------------------
    @torch.no_grad()
    def line_search(self, s, active=None, use_status=True):
        self.model.eval()
        self.cost.eval()
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
