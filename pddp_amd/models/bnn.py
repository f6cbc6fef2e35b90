"""Dropout Bayesian-neural-network dynamics model with particle moment
matching (reference: pddp/models/bnn/modules.py and losses.py; same factory,
constructor and `forward / fit / resample` contract, written independently).

    z --decode--> (mean, L)          L^T L = covariance        (encoding.py:304)
    particles X = mean + eps L       P standardised normals     (modules.py:317-358)
        eps at i > 0 re-whitens the PREVIOUS call's output particles
        (`infer_noise_variables`, modules.py:333-348): state carried across calls
    MLP on [augment(X), u]: fc -> concrete-dropout mask (fixed per particle,
        shared by the batch) -> ReLU ... -> fc_out = [dx, log_std]
                                                              (modules.py:200-264)
    output particles X + dx  --> mean, unbiased covariance --> encode
                                                              (modules.py:369-386)

This is a plugin model: it runs on PyTorch-ROCm (the MLP is two rocBLAS GEMMs
per call) and reaches the controllers through the plugin path
(controllers/plugin.py).  A hand-written MFMA rollout kernel is the next row of
SURVEY.md 8(f).
"""
import math

import torch
from torch.nn import Parameter

from .base import DynamicsModel
from ..utils.angular import augment_state, infer_augmented_state_size
from ..utils.classproperty import classproperty
from ..utils.encoding import (StateEncoding, decode_covar_sqrt, decode_mean,
                              encode)


def gaussian_log_likelihood(targets, pred_means, pred_stds=None):
    """losses.py:20-38"""
    d = pred_means - targets
    if pred_stds is None:
        return -0.5 * (d ** 2).sum(-1)
    return (-0.5 * ((d / pred_stds) ** 2).sum(-1) - pred_stds.log().sum(-1)
            - 0.5 * math.log(2 * math.pi))


def particles_covar(x):
    """Unbiased covariance over the leading particle axis
    (utils/particles.py:136-149): x [P, D] -> [D, D]; [P, B, D] -> [B, D, D]."""
    d = x - x.mean(dim=0)
    if d.dim() == 3:
        d = d.permute(1, 0, 2)
    return d.transpose(-1, -2) @ d / (x.shape[0] - 1)



def set_network_precision(mode):
    """Arithmetic of the fused network kernel's hidden-to-hidden contraction
    (pddp_bnn_mlp_precision, include/pddp_hip.h): 0 exact f32 (default), 3 the
    bf16-split twin (f32 to rounding, ~1.5x faster at [200, 200]).  Process-wide;
    returns the previous mode."""
    from .. import _native
    return int(_native.lib().pddp_bnn_mlp_precision(int(mode)))


def set_network_deal(deal):
    """How the exact-f32 network kernel deals its layer-2 contraction out over
    a workgroup at [200, 200] (pddp_bnn_mlp_deal, include/pddp_hip.h; DESIGN.md
    3.6): -1 the default (2 for inference, 1 for forward mode), 0 / 1 / 2 one
    deal for both - 2 in forward mode is 5 % faster and sums in another order
    (a ReLU linearised within rounding of zero may take the other sign).
    Process-wide; returns the previous value."""
    from .. import _native
    return int(_native.lib().pddp_bnn_mlp_deal(int(deal)))


def bump_generation(model):
    """Marks every tensor a captured hipGraph may hold a pointer to
    (normalisation buffers, dropout masks, cached noise) as replaced: the
    solver drops its graphs when this number changes
    (controllers/solver.py `_graphs_fresh`)."""
    model.__dict__["_pddp_generation"] = generation(model) + 1


def generation(model):
    return getattr(model, "_pddp_generation", 0)


class ConcreteDropout(torch.nn.Module):
    """Concrete (continuous-relaxation) dropout with a mask that is sampled
    once per (particle, unit) and then held fixed until `resample()` - the
    Bayesian weights of one particle (modules.py:494-600 `CDropout`,
    `mask_dims=2`).  `binary=True` gives plain Bernoulli masks (`BDropout`,
    modules.py:413-491)."""

    def __init__(self, rate=0.5, temperature=0.1, reg=1.0, binary=False):
        super(ConcreteDropout, self).__init__()
        self.binary = binary
        self.register_buffer("reg", torch.tensor(float(reg)))
        self.register_buffer("rate", torch.tensor(float(rate)))
        self.temperature = Parameter(torch.tensor(float(temperature)),
                                     requires_grad=False)
        keep = 1.0 - rate
        self.logit_p = Parameter(torch.tensor(math.log(keep / (1.0 - keep))),
                                 requires_grad=not binary)
        self.noise = None  # uniform (concrete) or Bernoulli (binary) draws

    @property
    def p(self):
        return self.logit_p.sigmoid()

    def resample(self):
        self.noise = None

    def _mask(self, noise):
        if self.binary:
            return noise
        logits = self.logit_p + noise.log() - (1 - noise).log()
        return (logits / self.temperature).sigmoid()

    def _draw(self, like):
        if self.binary:
            return torch.bernoulli(self.p.detach().expand(like.shape)).to(like)
        return torch.rand_like(like)

    def regularization(self, weight, bias):
        """modules.py:434-447,517-530.  As in the reference the keep
        probability that enters the regulariser is the INITIAL 1 - rate (its
        BDropout.regularization resets p = 1 - rate), so `logit_p` only learns
        through the likelihood."""
        p = 1.0 - self.rate
        reg = self.reg * (p * (weight ** 2).sum() +
                          ((bias ** 2).sum() if bias is not None else 0.0))
        if not self.binary:
            reg = reg - (-(1 - p) * (1 - p).log() - p * p.log())
        return reg

    def forward(self, x, resample=False):
        shape = x.shape[-2:]
        if resample:  # fresh draw for every element, nothing cached
            mask = self._mask(self._draw(x))
            return x * (mask if self.training else mask.detach())
        if self.noise is None or self.noise.shape != shape:
            self.noise = self._draw(x.reshape(-1, *shape)[0]).detach()
        mask = self._mask(self.noise)
        return x * (mask if self.training else mask.detach())


class CDropout(ConcreteDropout):
    """The reference's name and constructor order for concrete dropout
    (modules.py:494-515)."""

    def __init__(self, temperature=0.1, rate=0.5, reg=1.0, **kwargs):
        super(CDropout, self).__init__(rate=rate, temperature=temperature,
                                       reg=reg, binary=False)


class BDropout(ConcreteDropout):
    """Bernoulli dropout with a fixed rate, masks held per particle
    (modules.py:413-491)."""

    def __init__(self, rate=0.1, reg=1.0, **kwargs):
        super(BDropout, self).__init__(rate=rate, reg=reg, binary=True)


class BayesianMLP(torch.nn.Module):
    """fc -> dropout -> ReLU, ..., fc_out (modules.py:792-864
    `bayesian_model`): Xavier-normal weights with ReLU gain, biases
    U(-0.1, 0.1), concrete dropout with initial keep probability 0.5."""

    def __init__(self, in_features, out_features, hidden_features,
                 initial_p=0.5, binary_dropout=False, dropout_layers=None):
        """`dropout_layers`: the reference's keyword (modules.py:803,818-830) -
        a dropout class or instance, or a list of them, one per hidden layer;
        default: concrete dropout with keep probability `initial_p`."""
        super(BayesianMLP, self).__init__()
        dims = [in_features] + list(hidden_features)
        self.hidden = torch.nn.ModuleList(
            [torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        if dropout_layers is None:
            drops = [ConcreteDropout(rate=initial_p, binary=binary_dropout)
                     for _ in hidden_features]
        else:
            if not isinstance(dropout_layers, (list, tuple)):
                dropout_layers = [dropout_layers] * len(hidden_features)
            drops = [d() if isinstance(d, type) else d for d in dropout_layers]
        self.drops = torch.nn.ModuleList(drops)
        self.out = torch.nn.Linear(dims[-1], out_features)
        gain = torch.nn.init.calculate_gain("relu")
        for lin in list(self.hidden) + [self.out]:
            torch.nn.init.xavier_normal_(lin.weight, gain=gain)
            torch.nn.init.uniform_(lin.bias, -0.1, 0.1)

    def resample(self):
        for d in self.drops:
            d.resample()

    def regularization(self):
        """Each dropout regularises the layer that FOLLOWS it
        (modules.py:753-771)."""
        nxt = list(self.hidden[1:]) + [self.out]
        return sum(d.regularization(l.weight, l.bias)
                   for d, l in zip(self.drops, nxt))

    # -- fused inference kernel (csrc/bnn_mlp.hip) ------------------------------
    _NATIVE_H = (64, 128, 200)

    def _native_ok(self, x, resample):
        if resample or not x.is_cuda or x.dtype not in (
                torch.float32, torch.float64):
            return False
        if self.hidden[0].weight.dtype != x.dtype:
            return False
        if x.dim() < 2 or len(self.hidden) != 2:
            return False
        if torch.is_grad_enabled() and (
                x.requires_grad or self.out.weight.requires_grad):
            return False  # the derivative rollout differentiates through us
        h = self.hidden[0].out_features
        return (h in self._NATIVE_H and self.hidden[1].out_features == h
                and self.hidden[1].in_features == h
                and self.hidden[0].in_features <= 15
                and self.out.out_features <= 16
                and getattr(self, "use_native", True))

    def _mask_t(self, k, P, like):
        """Layer k's mask [P, H] for the kernel (include/pddp_hip.h).  Same
        draw as ConcreteDropout.forward on first use; cached until the noise
        or the dropout parameters change."""
        drop = self.drops[k]
        H = self.hidden[k].out_features
        if drop.noise is None or drop.noise.shape != (P, H):
            drop.noise = drop._draw(like.new_empty(P, H)).detach()
        key = (id(drop.noise), drop.noise._version, drop.logit_p._version,
               drop.temperature._version, like.dtype)
        cache = self.__dict__.setdefault("_mask_cache", {})
        if cache.get(k, (None, None))[0] != key:
            m = drop._mask(drop.noise).detach()
            cache[k] = (key, m.to(like.dtype).contiguous())
        return cache[k][1]

    def _forward_native(self, x, out_rows=None, live_rows=None):
        """`out_rows`: only the first rows of fc_out (the mean increments when
        the predicted std is not used, modules.py:262).  `live_rows`: an int32
        device scalar - only that many leading rows are computed (the kernel
        reads it: no host synchronisation), the others are left alone."""
        from .. import _native
        P, in_dim = x.shape[-2], x.shape[-1]
        H = self.hidden[0].out_features
        out_dim = self.out.out_features if out_rows is None else int(out_rows)
        xc = x.detach().contiguous()
        R = xc.numel() // in_dim
        y = torch.empty(*x.shape[:-1], out_dim, dtype=x.dtype, device=x.device)
        m1, m2 = self._mask_t(0, P, xc), self._mask_t(1, P, xc)
        c = lambda t: t.detach().contiguous()
        p = _native.ptr
        fn = getattr(_native.lib(),
                     "pddp_bnn_mlp_rows_" + _native.suffix(x.dtype))
        rc = fn(
            R, P, in_dim, H, out_dim, p(xc), p(c(self.hidden[0].weight)),
            p(c(self.hidden[0].bias)), p(m1), p(c(self.hidden[1].weight)),
            p(c(self.hidden[1].bias)), p(m2), p(c(self.out.weight[:out_dim])),
            p(c(self.out.bias[:out_dim])), p(y), p(live_rows),
            _native.stream_handle(x.device))
        _native.check(rc, "pddp_bnn_mlp_rows_" + _native.suffix(x.dtype))
        return y

    def _jvp_native(self, F, P, out_rows, group=16, live=None, live_rows=None):
        """Forward-mode pass of csrc/bnn_mlp.hip: F [(states P) group, in_dim],
        groups of 8 / 16 / 32 rows = primal input + tangent rows; returns the
        first `out_rows` outputs per row (include/pddp_hip.h
        pddp_bnn_mlp_jvp_live_f32).  `live`: rows of a group in use (default:
        all) - the rest is neither read nor written."""
        from .. import _native
        in_dim = F.shape[-1]
        H = self.hidden[0].out_features
        R = F.shape[0]
        Y = torch.empty(R, out_rows, dtype=F.dtype, device=F.device)
        m1, m2 = self._mask_t(0, P, F), self._mask_t(1, P, F)
        c = lambda t: t.detach().contiguous()
        p = _native.ptr
        # (`live_rows`: an int32 device scalar - only that many leading rows)
        fn = getattr(_native.lib(),
                     "pddp_bnn_mlp_jvp_rows_" + _native.suffix(F.dtype))
        rc = fn(
            R, P, int(group), int(group if live is None else live), in_dim, H,
            out_rows, p(F),
            p(c(self.hidden[0].weight)),
            p(c(self.hidden[0].bias)), p(m1), p(c(self.hidden[1].weight)),
            p(c(self.hidden[1].bias)), p(m2), p(c(self.out.weight[:out_rows])),
            p(c(self.out.bias[:out_rows])), p(Y), p(live_rows),
            _native.stream_handle(F.device))
        _native.check(rc, "pddp_bnn_mlp_jvp_rows_" + _native.suffix(F.dtype))
        return Y

    def forward(self, x, resample=False):
        if self._native_ok(x, resample):
            return self._forward_native(x)
        for lin, drop in zip(self.hidden, self.drops):
            x = torch.relu(drop(lin(x), resample=resample))
        return self.out(x)


def bayesian_model(in_features, out_features, hidden_features, **kwargs):
    """fc -> dropout -> ReLU, ..., fc_out with the reference's initialisation
    (modules.py:792-864); accepts its `dropout_layers` keyword."""
    return BayesianMLP(in_features, out_features, hidden_features, **kwargs)


BSequential = BayesianMLP  # the container bayesian_model returns (modules.py:744)


def bnn_dynamics_model_factory(state_size, action_size, hidden_features,
                               angular_indices=None, non_angular_indices=None,
                               constrain_min=None, constrain_max=None,
                               particles=False, **kwargs):
    """modules.py:44-398.  Returns a `BNNDynamicsModel` class (or the
    particle-level model when `particles=True`).  `constrain_min / _max`: the
    action is squashed into the box before it enters the network
    (modules.py:78,118-121)."""
    should_constrain = constrain_min is not None and constrain_max is not None
    angular = angular_indices is not None and non_angular_indices is not None
    aug_size = (infer_augmented_state_size(angular_indices,
                                           non_angular_indices)
                if angular else state_size)

    _ang = tuple(int(i) for i in angular_indices) if angular else ()
    _non = (tuple(int(i) for i in non_angular_indices) if angular
            else tuple(range(state_size)))

    class ParticlesBNNDynamicsModel(DynamicsModel):

        # state layout, for the fused rollout (controllers/plugin.py)
        angular_indices_ = _ang
        non_angular_indices_ = _non

        def __init__(self):
            super(ParticlesBNNDynamicsModel, self).__init__()
            self.model = BayesianMLP(aug_size + action_size, 2 * state_size,
                                     hidden_features, **kwargs)
            for name, v in (("X_mean", 0.0), ("X_std", 1.0),
                            ("X_std_inv", 1.0), ("dX_mean", 0.0),
                            ("dX_std", 1.0), ("dX_std_inv", 1.0)):
                self.register_buffer(name, torch.tensor(v))
            self.eps_out = {}

        @classproperty
        def action_size(cls):
            return action_size

        @classproperty
        def state_size(cls):
            return state_size

        def resample(self):
            self.eps_out = {}
            self.model.resample()
            bump_generation(self)

        def _features(self, X, u):
            if should_constrain:
                from ..utils.constraint import constrain
                u = constrain(u, constrain_min, constrain_max)
            Xa = (augment_state(X, angular_indices, non_angular_indices)
                  if angular else X)
            P = X.shape[-2]
            ue = u.unsqueeze(-2).expand(*u.shape[:-1], P, u.shape[-1])
            return (torch.cat([Xa, ue], dim=-1) - self.X_mean) * self.X_std_inv

        def fit(self, X, U, dX, n_iter=500, batch_size=128, reg_scale=1.0,
                learning_rate=1e-4, likelihood=gaussian_log_likelihood,
                resample=True, normalize=True, quiet=False, graph=None, **kw):
            """Maximum-likelihood training with the dropout regulariser
            (modules.py:131-198): Adam(amsgrad), shuffled mini-batches.

            `graph` (default: on for CUDA data): one training step - gather of
            the mini-batch, forward with freshly drawn concrete-dropout masks,
            likelihood + regulariser, backward, Adam - is captured once into a
            hipGraph and replayed per full mini-batch; a trailing partial
            batch of an epoch (another shape) runs eagerly on the same
            optimizer state.  The network is small: a step is ~60 launches of
            a few microseconds each, i.e. launch-bound."""
            Xa = (augment_state(X, angular_indices, non_angular_indices)
                  if angular else X)
            X_ = torch.cat([Xa, U], dim=-1).detach()
            dX = dX.detach()
            N = X_.shape[0]
            if normalize:
                self.X_mean = X_.mean(0)
                self.X_std = X_.std(0)
                self.X_std_inv = self.X_std.reciprocal()
                self.dX_mean = dX.mean(0)
                self.dX_std = dX.std(0)
                self.dX_std_inv = self.dX_std.reciprocal()
            params = [p for p in self.parameters() if p.requires_grad]
            use_graph = X_.is_cuda if graph is None else bool(graph)
            use_graph = use_graph and X_.is_cuda
            if not resample and N % min(batch_size, N) != 0:
                # held masks are re-drawn (python side) whenever the batch
                # shape changes: a trailing partial batch would leave the
                # captured step on the masks of the previous epoch
                use_graph = False
            opt = torch.optim.Adam(params, learning_rate, amsgrad=True,
                                   capturable=use_graph)
            self.train()
            log_dX_std = self.dX_std.log()

            def step(idx):
                opt.zero_grad(set_to_none=True)
                out = self.model((X_[idx] - self.X_mean) * self.X_std_inv,
                                 resample=resample)
                mean, log_std = out.split([state_size, state_size], -1)
                mean = mean * self.dX_std + self.dX_mean
                log_std = log_std + log_dX_std
                loss = -likelihood(dX[idx], mean, log_std.exp()).mean()
                loss = loss + reg_scale * self.model.regularization() / N
                loss.backward()
                opt.step()

            full = min(batch_size, N)
            static_idx, captured, warm = None, None, 0
            if use_graph:
                static_idx = torch.zeros(full, dtype=torch.long,
                                         device=X_.device)
            it = 0
            while it < n_iter:
                perm = torch.randperm(N, device=X_.device)
                for s0 in range(0, N, batch_size):
                    idx = perm[s0:s0 + batch_size]
                    if not use_graph or idx.shape[0] != full:
                        step(idx)
                    elif captured is None and warm < 3:
                        # the first full batches run eagerly on a side stream
                        # (allocator / autograd warm-up before the capture;
                        # they are ordinary training steps)
                        side = torch.cuda.Stream(device=X_.device)
                        side.wait_stream(torch.cuda.current_stream(X_.device))
                        with torch.cuda.stream(side):
                            static_idx.copy_(idx)
                            step(static_idx)
                        torch.cuda.current_stream(X_.device).wait_stream(side)
                        warm += 1
                    else:
                        static_idx.copy_(idx)
                        if captured is None:
                            # a likelihood / layer that synchronises with the
                            # host (`.item()`, printing, a data-dependent
                            # branch) cannot be captured: the reference API
                            # accepts it, so train eagerly instead
                            try:
                                captured = torch.cuda.CUDAGraph()
                                with torch.cuda.graph(captured):
                                    step(static_idx)
                            except RuntimeError as err:
                                # capture-specific failures only (an illegal
                                # operation while the stream is capturing);
                                # OOM, shape errors and the like are bugs and
                                # propagate
                                msg = str(err).lower()
                                if not ("captur" in msg or
                                        "streamcapture" in msg):
                                    raise
                                import warnings
                                warnings.warn(
                                    "BNN fit: the training step cannot be "
                                    "captured into a hipGraph (%s); training "
                                    "eagerly" % str(err).splitlines()[0])
                                captured, use_graph = None, False
                                torch.cuda.synchronize(X_.device)
                                step(idx)
                                it += 1
                                if it >= n_iter:
                                    break
                                continue
                        captured.replay()
                    it += 1
                    if it >= n_iter:
                        break
            self.last_fit_used_graph = captured is not None
            # graph replays update logit_p in place without touching the
            # version counters the mask cache is keyed by; new normalisation
            # buffers / masks: captured solver graphs are stale from here on
            self.model.__dict__.pop("_mask_cache", None)
            bump_generation(self)
            return self

        def forward(self, X, u, i, resample=False, use_predicted_std=False,
                    independent_noise=False, **kw):
            """X [..., P, D] particles, u [..., m] -> next particles."""
            out = self.model(self._features(X, u), resample=resample)
            dx, log_std = out.split([state_size, state_size], -1)
            dx = dx * self.dX_std + self.dX_mean
            if use_predicted_std:
                log_std = log_std + self.dX_std.log()
                if resample or i not in self.eps_out:
                    eps = torch.randn_like(dx.reshape(-1, *dx.shape[-2:])[0])
                    self.eps_out[i] = (eps - eps.mean(0)) / eps.std(0)
                std = log_std.exp()
                if independent_noise:
                    std = std.detach()
                dx = dx + std * self.eps_out[i]
            return X + dx

    class BNNDynamicsModel(ParticlesBNNDynamicsModel):

        def __init__(self, n_particles=100):
            super(BNNDynamicsModel, self).__init__()
            self.n_particles = n_particles
            self.eps_in = {}   # time index -> standardised normals [P, D]
            self.output = {}   # time index -> last output particles [..., P, D]

        def resample(self):
            self.eps_in = {}
            self.output = {}
            super(BNNDynamicsModel, self).resample()  # (bumps the generation)

        def forward(self, z, u, i, encoding=StateEncoding.DEFAULT,
                    identical_inputs=False, resample=False,
                    sample_input_distribution=True,
                    infer_noise_variables=True, quiet=False, **kw):
            """modules.py:287-386.  z [n] or [rows, n]; `i` is the time index
            the noise caches are keyed by."""
            i = int(i)
            P = self.n_particles
            mean = decode_mean(z, encoding)
            X = mean.unsqueeze(-2).expand(*mean.shape[:-1], P, mean.shape[-1])
            if sample_input_distribution:
                if resample or i not in self.eps_in:
                    e = torch.randn(P, mean.shape[-1], dtype=z.dtype,
                                    device=z.device)
                    self.eps_in[i] = (e - e.mean(0)) / e.std(0)
                L = decode_covar_sqrt(z, encoding)  # [..., D, D], L^T L = cov
                eps = self.eps_in[i]
                if infer_noise_variables and i > 0 and (i - 1) in self.output:
                    # re-whiten the previous step's particles: eps L = delta
                    prev = self.output[i - 1]
                    if identical_inputs and z.dim() == 2:
                        delta = (prev.reshape(-1, P, prev.shape[-1])[0]
                                 - mean[0])
                        eps = torch.linalg.solve_triangular(
                            L[0], delta, upper=True, left=False).detach()
                    elif prev.shape[:-2] == mean.shape[:-1]:
                        delta = prev - mean.unsqueeze(-2)
                        eps = torch.linalg.solve(
                            L.transpose(-1, -2),
                            delta.transpose(-1, -2)).transpose(-1, -2).detach()
                X = X + eps @ L
            out = super(BNNDynamicsModel, self).forward(X, u, i,
                                                        resample=resample,
                                                        **kw)
            if infer_noise_variables:
                self.output[i] = out.detach()
            M = out.mean(dim=-2)
            if encoding in (StateEncoding.FULL_COVARIANCE_MATRIX,
                            StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
                d = out - M.unsqueeze(-2)
                C = d.transpose(-1, -2) @ d / (P - 1)
                try:
                    return encode(M, C=C, encoding=encoding)
                except RuntimeError:
                    pass  # not positive definite: fall back to the std
            return encode(M, S=out.std(dim=-2), encoding=encoding)

    return ParticlesBNNDynamicsModel if particles else BNNDynamicsModel


def load_reference_state(model, state):
    """Copies weights, normalisation buffers and cached noise captured from a
    reference model (tools/make_golden.py `capture_bnn`) into `model`; used by
    the parity tests."""
    mlp = model.model
    t = lambda a: torch.as_tensor(a)
    with torch.no_grad():
        for k, lin in enumerate(mlp.hidden):
            lin.weight.copy_(t(state["fc_%d.weight" % k]))
            lin.bias.copy_(t(state["fc_%d.bias" % k]))
            mlp.drops[k].logit_p.copy_(t(state["drop_%d.logit_p" % k]))
            mlp.drops[k].temperature.copy_(t(state["drop_%d.temperature" % k]))
            if ("drop_%d.noise" % k) in state:
                mlp.drops[k].noise = t(state["drop_%d.noise" % k]).to(
                    lin.weight)
        mlp.out.weight.copy_(t(state["fc_out.weight"]))
        mlp.out.bias.copy_(t(state["fc_out.bias"]))
        for name in ("X_mean", "X_std", "X_std_inv", "dX_mean", "dX_std",
                     "dX_std_inv"):
            setattr(model, name, t(state[name]).to(mlp.out.weight))
    if hasattr(model, "eps_in"):
        model.eps_in = {int(k.split("/")[1]): t(v).to(mlp.out.weight)
                        for k, v in state.items() if k.startswith("eps_in/")}
        model.output = {}
    model.eps_out = {}
    mlp.__dict__.pop("_mask_cache", None)
    bump_generation(model)
    return model

