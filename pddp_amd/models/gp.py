"""Gaussian-process dynamics with exact moment matching - BASELINE.json's
"GP dynamics" (north_star; configs[3]).

The reference ships NO Gaussian process (pddp/models/__init__.py:17-20 exports
the BNN only): this plugin is the build's own,
behind the reference's `DynamicsModel` contract (models/base.py:24-83:
`forward(z, u, i, encoding, ...)`, `fit(X, U, dX)`), so that it drops into
`iLQRController` / `PDDPController` like the BNN does.  PARITY UNPINNED: there
is no reference output to hold it to; its checker is the independent numpy
restatement oracle/gp_port.py (tests/test_gp.py) plus a Monte-Carlo check of
the moments.

Model: one GP per state increment, squared-exponential kernel with automatic
relevance determination on the inputs x~ = [x_non-angular, sin a, cos a, u]
(the reference's BNN features, modules.py:313-330).  Prediction at an
uncertain input x~ ~ N(m, S) is the exact first and second moment of the GP
posterior (Deisenroth & Rasmussen 2011, "PILCO", eqs. 14-23; Quinonero-Candela
et al. 2003):

    q_a[i]   = sf_a^2 |S L_a^-1 + I|^-1/2 exp(-1/2 nu_i^T (S + L_a)^-1 nu_i)
    mu_a     = beta_a^T q_a                                  nu_i = x~_i - m
    Q_ab[ij] = k_a(x~_i, m) k_b(x~_j, m) |R|^-1/2 exp(1/2 z_ij^T R^-1 S z_ij)
               R = S (L_a^-1 + L_b^-1) + I,  z_ij = L_a^-1 nu_i + L_b^-1 nu_j
    S_ab     = beta_a^T Q_ab beta_b - mu_a mu_b
               + [a = b] (sf_a^2 - tr(K_a^-1 Q_aa) + sn_a^2)
    cov[x~, D_a] = S (S + L_a)^-1 sum_i beta_a[i] q_a[i] nu_i

and the next state x' = x + D has mean mu_x + mu, covariance S_x + S_D + C +
C^T with C = cov[x, x~] (S + L_a)^-1 sum_i beta q nu (the state's covariance
with its own trigonometric features is exact by Stein's lemma:
cov[x, sin a_k] = S_x[:, k] E[cos a_k]).  The module below is batched torch
(einsum / solve / det) and differentiable: the definition, and the checker of
the HIP kernel `pddp_gp_step_f32 / _f64` (csrc/gp_step.hip),
which computes the same step and its Jacobian with respect to (z, u) in one
launch for any number of rows.  `forward` goes through the kernel whenever
nobody can ask autograd for gradients (`native_ok`, `native_step`); the
controllers' plugin path takes F_z, F_u of a whole nominal from one launch
(controllers/plugin.py `_dyn_derivs_gp`) and runs the line search on it
(`_line_search_gp`).
"""
import math

import torch

from .base import DynamicsModel
from ..utils.angular import augment_moments, augment_state
from ..utils.classproperty import classproperty
from ..utils.encoding import StateEncoding, decode_covar, decode_mean, encode
from ..utils.linalg import cholesky_solve


def gp_dynamics_model_factory(state_size, action_size, angular_indices=(),
                              non_angular_indices=None):
    """Class of GP dynamics models for one system (mirrors
    `bnn_dynamics_model_factory`, modules.py:44-52)."""
    ai = list(angular_indices)
    ni = (list(non_angular_indices) if non_angular_indices is not None
          else [k for k in range(state_size) if k not in ai])
    d_in = len(ni) + 2 * len(ai) + action_size

    class GPDynamicsModel(DynamicsModel):

        def __init__(self, jitter=1e-8):
            super(GPDynamicsModel, self).__init__()
            E = state_size
            self.jitter = float(jitter)
            self.register_buffer("Xt", torch.zeros(1, d_in))       # inputs
            self.register_buffer("beta", torch.zeros(E, 1))        # K^-1 y
            self.register_buffer("Kinv", torch.zeros(E, 1, 1))
            self.log_ell = torch.nn.Parameter(torch.zeros(E, d_in))
            self.log_sf = torch.nn.Parameter(torch.zeros(E))
            self.log_sn = torch.nn.Parameter(torch.full((E,), -2.0))
            self.fitted = False
            self._native_cache = {}

        @classproperty
        def action_size(cls):
            return action_size

        @classproperty
        def state_size(cls):
            return state_size

        @classproperty
        def angular_indices(cls):
            return ai

        @classproperty
        def non_angular_indices(cls):
            return ni

        # -- training ---------------------------------------------------------
        def _kernel(self, A, B_, a=None):
            """k_a(A, B) for every output a: [E, |A|, |B|]."""
            ell = self.log_ell.exp()
            d = (A.unsqueeze(-2) - B_.unsqueeze(-3)).unsqueeze(0) / \
                ell[:, None, None, :]
            return (2.0 * self.log_sf).exp()[:, None, None] * \
                torch.exp(-0.5 * (d ** 2).sum(-1))

        def _nlml(self, Xt, Y):
            """Negative log marginal likelihood, summed over the outputs."""
            M = Xt.shape[0]
            K = self._kernel(Xt, Xt)
            K = K + ((2.0 * self.log_sn).exp() + self.jitter)[:, None, None] * \
                torch.eye(M, dtype=Xt.dtype, device=Xt.device)
            L = torch.linalg.cholesky(K)
            a_ = cholesky_solve(Y.t().unsqueeze(-1), L).squeeze(-1)
            return (0.5 * (Y.t() * a_).sum() +
                    torch.diagonal(L, dim1=-2, dim2=-1).log().sum() +
                    0.5 * M * Y.shape[1] * math.log(2 * math.pi))

        def fit(self, X, U, dX, n_iter=0, learning_rate=0.05, quiet=True,
                max_points=None, **kwargs):
            """Conditions the GPs on (X, U) -> dX.  Hyper-parameters start at
            the usual data scales (length = input std, sf = target std, sn =
            sf / 10) and take `n_iter` Adam steps on the marginal likelihood.
            `max_points`: keep the last rows only (a dense GP is O(M^3))."""
            Xt = torch.cat([augment_state(X, ai, ni) if ai else X, U],
                           dim=-1).detach()
            Y = dX.detach()
            if max_points is not None:
                Xt, Y = Xt[-max_points:], Y[-max_points:]
            with torch.no_grad():
                sd = Xt.std(0).clamp_min(1e-3)
                self.log_ell.copy_(sd.log().expand(state_size, -1))
                sy = Y.std(0).clamp_min(1e-6)
                self.log_sf.copy_(sy.log())
                self.log_sn.copy_((0.1 * sy).log())
            if n_iter > 0:
                opt = torch.optim.Adam([self.log_ell, self.log_sf, self.log_sn],
                                       learning_rate)
                for _ in range(n_iter):
                    opt.zero_grad()
                    loss = self._nlml(Xt, Y)
                    loss.backward()
                    opt.step()
            self.condition(Xt, Y)
            return self

        @torch.no_grad()
        def condition(self, Xt, Y):
            """Posterior weights for the current hyper-parameters."""
            M = Xt.shape[0]
            K = self._kernel(Xt, Xt)
            K = K + ((2.0 * self.log_sn).exp() + self.jitter)[:, None, None] * \
                torch.eye(M, dtype=Xt.dtype, device=Xt.device)
            # The factorisation of the E kernel matrices on the HOST, in
            # float64 (once per fit; M <= 1000: milliseconds): on this image's
            # PyTorch 2.10 / ROCm 7.0 build the batched device potrf fails
            # outright on a [6, 300, 300] float batch ("unspecified launch
            # failure", bench.py --gp-points 300) - after the batched potrs
            # that writes outside its outputs (utils/linalg.py), the second
            # defect of that library path.  The weights go back in the model's
            # dtype; nothing on the controller's path runs here.
            Kh, Yh = K.double().cpu(), Y.double().cpu()
            L = torch.linalg.cholesky(Kh)
            eye = torch.eye(M, dtype=torch.float64).expand(state_size, M, M)
            self.Kinv = cholesky_solve(eye, L).to(dtype=Xt.dtype,
                                                  device=Xt.device)
            self.beta = cholesky_solve(Yh.t().unsqueeze(-1), L).squeeze(-1).to(
                dtype=Xt.dtype, device=Xt.device)
            self.Xt = Xt.clone()
            self.fitted = True
            self._drop_native_view()

        def _drop_native_view(self):
            """The kernel's view (and every hipGraph captured around its raw
            pointers: ILQRSolver._graphs_fresh looks at the generation BEFORE
            it replays) dies with the tensors it was made from - eagerly, not
            at the next eager native_step, which a graph-only controller never
            makes."""
            from .bnn import bump_generation
            self._native_cache = {}
            bump_generation(self)

        def _load_from_state_dict(self, *args, **kwargs):
            super(GPDynamicsModel, self)._load_from_state_dict(*args, **kwargs)
            self._drop_native_view()

        def _apply(self, fn, *args, **kwargs):
            # (.to / .cuda / .double replace parameters and buffers)
            out = super(GPDynamicsModel, self)._apply(fn, *args, **kwargs)
            self._drop_native_view()
            return out

        # -- prediction ---------------------------------------------------------
        def moments(self, m, S, max_bytes=1 << 29):
            """Exact moments of the increments at x~ ~ N(m, S): m [..., d],
            S [..., d, d] -> mu [..., E], Sig [..., E, E], W [..., d, E] with
            cov[x~, D_a] = S W[:, a]  (W_a = (S + L_a)^-1 sum_i beta q nu).
            Rows are processed in chunks: Q is E^2 M^2 numbers per row."""
            lead = m.shape[:-1]
            mf = m.reshape(-1, m.shape[-1])
            Sf = S.reshape(-1, *S.shape[-2:])
            M_ = self.Xt.shape[0]
            per_row = 6 * state_size ** 2 * M_ * M_ * m.element_size()
            step = max(1, int(max_bytes // max(per_row, 1)))
            outs = [self._moments(mf[r:r + step], Sf[r:r + step])
                    for r in range(0, mf.shape[0], step)]
            mu = torch.cat([o[0] for o in outs]).reshape(*lead, -1)
            Sig = torch.cat([o[1] for o in outs]).reshape(
                *lead, state_size, state_size)
            W = torch.cat([o[2] for o in outs]).reshape(
                *lead, d_in, state_size)
            return mu, Sig, W

        def _moments(self, m, S):
            E, d = state_size, d_in
            dt, dev = m.dtype, m.device
            Xt, beta, Kinv = self.Xt.to(dt), self.beta.to(dt), self.Kinv.to(dt)
            ell2 = (2.0 * self.log_ell).exp().to(dt)          # [E, d]
            sf2 = (2.0 * self.log_sf).exp().to(dt)            # [E]
            sn2 = (2.0 * self.log_sn).exp().to(dt)
            eye = torch.eye(d, dtype=dt, device=dev)
            nu = Xt - m.unsqueeze(-2)                         # [..., M, d]
            # ---- mean and input-output covariance
            SL = S.unsqueeze(-3) + torch.diag_embed(ell2)     # [..., E, d, d]
            # (S + L)^-1 nu^T  -> [..., E, d, M]
            sol = torch.linalg.solve(SL, nu.transpose(-1, -2).unsqueeze(-3))
            quad = (nu.transpose(-1, -2).unsqueeze(-3) * sol).sum(-2)  # [.., E, M]
            det = torch.linalg.det(S.unsqueeze(-3) / ell2.unsqueeze(-2) + eye)
            q = sf2.unsqueeze(-1) * det.unsqueeze(-1).rsqrt() * \
                torch.exp(-0.5 * quad)                        # [..., E, M]
            bq = beta * q                                     # [..., E, M]
            mu = bq.sum(-1)                                   # [..., E]
            W = (sol * bq.unsqueeze(-2)).sum(-1).transpose(-1, -2)  # [..., d, E]
            # ---- covariance of the increments
            iL = 1.0 / ell2                                   # [E, d]
            # log k_a(x_i, m) = log sf2_a - 1/2 nu_i^T L_a^-1 nu_i
            lk = sf2.log().unsqueeze(-1) - 0.5 * torch.einsum(
                "...md,ed->...em", nu ** 2, iL)               # [..., E, M]
            zi = nu.unsqueeze(-3) * iL.unsqueeze(-2)          # [..., E, M, d]
            Rm = S.unsqueeze(-3).unsqueeze(-3) * \
                (iL.unsqueeze(1) + iL.unsqueeze(0)).unsqueeze(-2) + eye
            # R[a, b] = S (L_a^-1 + L_b^-1) + I        [..., E, E, d, d]
            RiS = torch.linalg.solve(Rm, S.unsqueeze(-3).unsqueeze(-3)
                                     .expand(Rm.shape))       # R^-1 S
            detR = torch.linalg.det(Rm)                       # [..., E, E]
            # z_ij^T R^-1 S z_ij = za_i' T za_i + 2 za_i' T zb_j + zb_j' T zb_j
            T = RiS
            Ta = torch.einsum("...amd,...abde->...abme", zi, T)   # [.., E,E,M,d]
            aa = torch.einsum("...abme,...ame->...abm", Ta, zi)   # za' T za
            Tb = torch.einsum("...bmd,...abde->...abme", zi, T)
            bb = torch.einsum("...abme,...bme->...abm", Tb, zi)   # zb' T zb
            ab = torch.einsum("...abie,...bje->...abij", Ta, zi)  # za_i' T zb_j
            expo = lk.unsqueeze(-2).unsqueeze(-1) + lk.unsqueeze(-3).unsqueeze(-2) \
                + 0.5 * (aa.unsqueeze(-1) + bb.unsqueeze(-2)) + ab
            Q = expo.exp() * detR.rsqrt().unsqueeze(-1).unsqueeze(-1)  # [..,E,E,M,M]
            Sig = torch.einsum("ai,...abij,bj->...ab", beta, Q, beta)
            Sig = Sig - mu.unsqueeze(-1) * mu.unsqueeze(-2)
            Qaa = torch.diagonal(Q, dim1=-4, dim2=-3).movedim(-1, -3)  # [..,E,M,M]
            tr = (Kinv * Qaa.transpose(-1, -2)).sum((-1, -2))          # tr(K^-1 Q)
            Sig = Sig + torch.diag_embed(sf2 - tr + sn2)
            Sig = 0.5 * (Sig + Sig.transpose(-1, -2))
            return mu, Sig, W

        def __getstate__(self):
            # (the kernel's cached view holds raw device pointers: not part of
            # a copy or a pickle)
            state = self.__dict__.copy()
            state["_native_cache"] = {}
            return state

        # -- the HIP kernel (csrc/gp_step.hip) ------------------------------------
        use_native = True

        def native_ok(self, z, encoding, jacobian=False):
            """True when `pddp_gp_step_*` covers this call: device tensors of
            f32 / f64, an encoding other than the full covariance matrix, one of
            the built (state_size, feature + action size) pairs, a training set
            that fits the workgroup's LDS."""
            if not (self.use_native and self.fitted and z.is_cuda):
                return False
            if z.dtype not in (torch.float32, torch.float64):
                return False
            if int(encoding) not in (1, 2, 3, 4):
                return False
            if len(ai) > 4 or len(ni) > 8:
                return False
            E = state_size
            n = {1: E + E * (E + 1) // 2, 2: 2 * E, 3: 2 * E, 4: E}[
                int(encoding)]
            if n + action_size > 64:
                return False
            from .. import _native
            fn = _native.lib().pddp_gp_step_lds_bytes
            need = fn(E, d_in, int(self.Xt.shape[0]), n + action_size,
                      int(bool(jacobian)), z.element_size())
            return 0 <= need <= 160 * 1024

        def _native_model(self, dtype, device, encoding):
            """The kernel's view of the conditioned GPs (cached per dtype; the
            cache is dropped by `condition`)."""
            from .. import _native
            # (keyed by what the view was made from: a loaded state or a
            # parameter changed in place makes a new one)
            src = (self.Xt, self.beta, self.Kinv, self.log_ell, self.log_sf,
                   self.log_sn)
            key = (dtype, str(device)) + tuple(
                (t.data_ptr(), t._version) for t in src)
            hit = self._native_cache.get(key)
            if hit is None:
                self._native_cache.clear()
                # (hipGraphs captured around the old view hold its pointers:
                # ILQRSolver._graphs_fresh drops them on a new generation)
                from .bnn import bump_generation
                bump_generation(self)
                conv = lambda t: t.detach().to(dtype=dtype, device=device) \
                    .contiguous()
                # pairs of training points for the kernel's scalar loads
                M_ = int(self.Xt.shape[0])
                mp, ps = ((M_ + 3) // 4) * 2, (2 * d_in + 3) & ~3
                xp = torch.zeros(2 * mp, d_in, dtype=torch.float64,
                                 device=self.Xt.device)
                xp[:M_] = self.Xt.detach().double()
                xp = torch.nn.functional.pad(
                    xp.view(mp, 2, d_in).transpose(1, 2).reshape(mp, 2 * d_in),
                    (0, ps - 2 * d_in))
                bp = torch.zeros(state_size, 2 * mp, dtype=torch.float64,
                                 device=self.Xt.device)
                bp[:, :M_] = self.beta.detach().double()
                keep = dict(
                    Xt=conv(self.Xt), beta=conv(self.beta),
                    Xt_pairs=conv(xp), beta_pairs=conv(bp),
                    Kinv=conv(0.5 * (self.Kinv + self.Kinv.transpose(-1, -2))),
                    inv_ell2=conv((-2.0 * self.log_ell.double()).exp()),
                    sf2=conv((2.0 * self.log_sf.double()).exp()),
                    sn2=conv((2.0 * self.log_sn.double()).exp()))
                g = _native.GpModel()
                g.state_size, g.action_size = state_size, action_size
                g.M = int(self.Xt.shape[0])
                g.n_ang, g.n_non = len(ai), len(ni)
                for k_, v in enumerate(ai):
                    g.ang[k_] = int(v)
                for k_, v in enumerate(ni):
                    g.non[k_] = int(v)
                for name, t in keep.items():
                    setattr(g, name, _native.ptr(t))
                hit = self._native_cache[key] = (g, keep)
            hit[0].encoding = int(encoding)
            return hit[0]

        @torch.no_grad()
        def native_step(self, z, u, encoding=StateEncoding.DEFAULT,
                        jacobian=False, Fz=None, Fu=None, row_mask=None,
                        rows_per_mask=1):
            """`forward` on the HIP kernel for rows z [R, n], u [R, m]; with
            `jacobian` also d z' / d z [R, n, n] and d z' / d u [R, n, m]
            (written into Fz, Fu when given: contiguous, R n n / R n m
            elements).  `row_mask` (uint8, one entry per `rows_per_mask`
            consecutive rows): rows of a zero entry are skipped, their outputs
            left as they are."""
            import ctypes
            from .. import _native
            z = z.detach().contiguous()
            u = u.detach().reshape(z.shape[0], -1).to(z.dtype).contiguous()
            R, n = z.shape
            g = self._native_model(z.dtype, z.device, encoding)
            out = torch.empty_like(z)
            if jacobian:
                if Fz is None:
                    Fz = torch.empty(R, n, n, dtype=z.dtype, device=z.device)
                    Fu = torch.empty(R, n, action_size, dtype=z.dtype,
                                     device=z.device)
                assert Fz.is_contiguous() and Fu.is_contiguous() and \
                    Fz.numel() == R * n * n and Fz.dtype == z.dtype and \
                    Fu.numel() == R * n * action_size and Fu.dtype == z.dtype
            else:
                Fz = Fu = None
            p = _native.ptr
            with torch.cuda.device(z.device):
                if row_mask is None:
                    _native.call("pddp_gp_step", z.dtype, ctypes.byref(g), R,
                                 p(z), p(u), p(out), p(Fz), p(Fu),
                                 _native.stream_handle(z.device))
                else:
                    assert row_mask.dtype == torch.uint8 and \
                        row_mask.is_contiguous() and row_mask.numel() * \
                        int(rows_per_mask) >= R
                    _native.call("pddp_gp_step_masked", z.dtype,
                                 ctypes.byref(g), R, p(z), p(u), p(out), p(Fz),
                                 p(Fu), p(row_mask), int(rows_per_mask),
                                 _native.stream_handle(z.device))
            return (out, Fz, Fu) if jacobian else out

        def forward(self, z, u, i, encoding=StateEncoding.DEFAULT,
                    identical_inputs=False, **kwargs):
            """Encoded state distribution and action -> next encoded state
            distribution (models/base.py:63-83)."""
            if not self.fitted:
                raise RuntimeError("GPDynamicsModel: call fit() first")
            single = z.dim() == 1
            if single:
                z, u = z.unsqueeze(0), u.reshape(1, -1)
            # nobody can ask autograd to differentiate this call: the kernel
            if z.dim() == 2 and not (torch.is_grad_enabled() and (
                    z.requires_grad or u.requires_grad)) and \
                    self.native_ok(z, encoding):
                out = self.native_step(z, u, encoding)
                return out[0] if single else out
            D = state_size
            mx = decode_mean(z, encoding, state_size=D)
            Sx = decode_covar(z, encoding, state_size=D)
            # moment-matched features [x_na, sin a, cos a] and the action
            ma, Sa = augment_moments(mx, Sx, ai, ni) if ai else (mx, Sx)
            na = ma.shape[-1]
            m = torch.cat([ma, u], dim=-1)
            S = z.new_zeros(*z.shape[:-1], d_in, d_in)
            S[..., :na, :na] = Sa
            mu, Sig, W = self.moments(m, S)
            # cov[x, x~]: exact (Stein): the state with its own features
            Cxf = z.new_zeros(*z.shape[:-1], D, d_in)
            if ni:
                Cxf[..., :, :len(ni)] = Sx[..., :, ni]
            for k, a_ in enumerate(ai):
                es = ma[..., len(ni) + 2 * k]       # E[sin a_k]
                ec = ma[..., len(ni) + 2 * k + 1]   # E[cos a_k]
                Cxf[..., :, len(ni) + 2 * k] = Sx[..., :, a_] * ec.unsqueeze(-1)
                Cxf[..., :, len(ni) + 2 * k + 1] = -Sx[..., :, a_] * \
                    es.unsqueeze(-1)
            C = Cxf @ W                                   # cov[x, D]  [..., D, E]
            Mn = mx + mu
            Cn = Sx + Sig + C + C.transpose(-1, -2)
            Cn = 0.5 * (Cn + Cn.transpose(-1, -2))
            if encoding in (StateEncoding.FULL_COVARIANCE_MATRIX,
                            StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
                out = encode(Mn, C=Cn, encoding=encoding)
            else:
                var = torch.diagonal(Cn, dim1=-2, dim2=-1).clamp_min(1e-12)
                out = encode(Mn, V=var, encoding=encoding)
            return out[0] if single else out

    return GPDynamicsModel
