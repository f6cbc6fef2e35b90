"""Device-resident batched iLQR engine: owns the HBM buffers of B independent
trajectories and drives the HIP kernels of libpddp_hip.so through the C ABI.

One `round()` = what the reference does in one pass of the retry loop of
`iLQRController.step` (pddp/controllers/ilqr.py:183-235) for EVERY live
trajectory at once:

    derivatives (only trajectories whose nominal changed, ilqr.py:198-209)
    -> backward Riccati sweep (ilqr.py:125-139)
    -> line search over A step sizes + costs (ilqr.py:148-160)
    -> accept / reject, mu schedule, fit-loop bookkeeping (ilqr.py:161-181,
       298-314)

with per-trajectory masks instead of Python control flow, so a round is a
fixed launch sequence with no host synchronisation.
"""
import ctypes

import torch

from .. import _native
from ..utils.encoding import StateEncoding

BRANCH_EIG, BRANCH_CHOLESKY = 0, 1


def _on_device(fn):
    """The C ABI takes a stream handle and launches on the CURRENT device: a
    solver that lives on another GPU makes its device current for the call."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        if torch.cuda.current_device() == self.device.index:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(self.device):
            return fn(self, *args, **kwargs)
    return wrapped


def fit_alphas(dtype, device):
    """ilqr.py:282 (the schedule `fit` actually uses):
    `1.025**(-torch.arange(10.0)**2).to(**tensor_opts)` - the `.to` binds to
    the parenthesised exponent, so the (integer-valued) exponents are cast and
    the power is taken in the RUN's dtype: a float64 run sees float64 step
    sizes.  Formed on the host so that every device sees the same bits."""
    return (1.025 ** (-torch.arange(10.0) ** 2).to(dtype)).to(device)


def mpc_alphas(dtype, device):
    """ilqr.py:116,189 default of `step` (used by forward(mpc=True))."""
    return (10.0 ** torch.linspace(0, -3, 11)).to(dtype=dtype, device=device)


class ILQRSolver(object):

    def __init__(self, problem, B, N, dtype, device, u_min=None, u_max=None,
                 alphas=None, branch=BRANCH_EIG, plugin=None, n=None, m=None,
                 kernel_variant=0):
        """`problem`: ctypes PddpProblem of a sample problem (everything in
        HIP), or None together with `plugin` (plugin.TorchProblem) and the
        encoded state / action sizes `n`, `m`: derivatives and the line search
        then come from the plugin modules, the sweep / accept stay HIP."""
        self.problem = problem
        self.plugin = plugin
        self.B, self.N = int(B), int(N)
        if problem is not None:
            self.n, self.m = problem.encoded_size, problem.action_size
        else:
            self.n, self.m = int(n), int(m)
        self.dtype, self.device = dtype, torch.device(device)
        if self.device.type != "cuda":
            raise _native.NativeError(
                "ILQRSolver needs a GPU device (no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lay = _native.record_layout(self.n, self.m)
        self.branch = branch
        # backward-sweep kernel of round(): 0 = auto (f32: approximate
        # v_rcp / v_sqrt kernels), `exact_variant()` = the IEEE-division twins
        self.kernel_variant = int(kernel_variant)
        opts = dict(dtype=dtype, device=self.device)
        B, N, n, m = self.B, self.N, self.n, self.m
        S, gs = self.lay.stride, self.lay.gain_stride
        self.u_min = None if u_min is None else \
            torch.as_tensor(u_min).to(**opts).reshape(m).contiguous()
        self.u_max = None if u_max is None else \
            torch.as_tensor(u_max).to(**opts).reshape(m).contiguous()
        self.alphas = (fit_alphas(dtype, self.device) if alphas is None
                       else torch.as_tensor(alphas).to(**opts).contiguous())
        A = self.A = self.alphas.numel()
        self.z0 = torch.zeros(B, n, **opts)
        self.Z = torch.zeros(B, N + 1, n, **opts)
        self.U = torch.zeros(B, N, m, **opts)
        self._rec = torch.zeros(B, N + 1, S, **opts)
        self.L = torch.zeros(B, N + 1, **opts)
        self.J_opt = torch.zeros(B, **opts)
        self.gains = torch.zeros(B, N, gs, **opts)
        self.gains_acc = torch.zeros(B, N, gs, **opts)
        # candidates, time-major like the reference's Z_new / U_new
        self.Zc = torch.zeros(B, N + 1, A, n, **opts)
        self.Uc = torch.zeros(B, N, A, m, **opts)
        self.Jc = torch.zeros(B, A, **opts)
        i32 = dict(dtype=torch.int32, device=self.device)
        u8 = dict(dtype=torch.uint8, device=self.device)
        f64 = dict(dtype=torch.float64, device=self.device)
        self.bwd_status = torch.zeros(B, **i32)
        self.state = torch.zeros(B, **i32)
        self.iter = torch.zeros(B, **i32)
        self.mu = torch.zeros(B, **f64)
        self.delta = torch.full((B,), 2.0, **f64)
        self.active = torch.zeros(B, **u8)
        self.fresh = torch.zeros(B, **u8)
        # PDDP_LIVE_SHARDS counters: running total of "still live after its
        # attempt" since the last reset (bench.py's unit accounting; the fit
        # loop itself looks at `active`)
        self.n_live = torch.zeros(256, **i32)
        self._graph = None  # (key, graph of a round [, round without derivs])
        self._rollout_graph = None
        self._model_gen = None  # model generation the plugin graphs captured
        self.graph_rollout = False  # nominal rollout of a plugin as a hipGraph
        self._fused = None  # None: untried, True / False: fused kernel applies
        # the whole round in one launch (round_nominal): None untried, True /
        # False applies / does not (set False to force the two launches)
        self._one_launch = None
        self._round_args = None
        # int64 [ceil(B / 16)][2] or None: pddp_round_nominal_f32's phase clock
        self.phase_ticks = None
        # False after a search launch that dropped the candidates (large
        # batches without records, pddp_search_candidates): `Zc`, `Uc` are then
        # scratch - only `Jc` and the nominal are results of that round.  (In
        # the record-free rounds of the sample problems `Uc` is never written:
        # the winner's actions are re-evaluated, include/pddp_hip.h)
        self.candidates_kept = True
        self._derivs_due = True
        # The sweep that evaluates the derivative records itself, from the
        # nominal (pddp_sweep_nominal_f32): None untried, then True / False.
        # With it `rec` is not kept up to date by round(); `sync_records()`
        # brings it up to date for whoever reads it.
        self._nominal_sweep = None if (self._nominal_sweep_possible() and
                                       self._nominal_sweep_pays()) else False
        self._rec_stale = False
        self._pp = None if problem is None else ctypes.addressof(problem)

    def _nominal_sweep_possible(self):
        """pddp_sweep_nominal_*'s domain (include/pddp_hip.h).  At every
        batch: from 12288 trajectories on the quad sweep on records is the
        faster SWEEP, but the round without records is still shorter (no 79 MB
        of records written by the line search and read back)."""
        # include/pddp_problem.h: PDDP_MODEL_CARTPOLE = 1,
        # PDDP_ENC_IGNORE_UNCERTAINTY = 4
        if not (self.plugin is None and self.problem is not None and
                self.m == 1 and self.problem.encoding == 4 and
                self.kernel_variant == 0):
            return False
        if self.problem.model == 1:  # cartpole: csrc/riccati_n4_elem.hpp
            # (f32, and - round 5 - the same mapping in f64)
            return (self.n == 4 and
                    self.u_min is not None and self.u_max is not None and
                    self.branch == BRANCH_EIG)
        # pendulum (3), double cartpole (2): csrc/riccati_mfma16_nominal.hpp -
        # f32 and f64, both branches, bounded or not
        return self.problem.model in (2, 3) and \
            (self.u_min is None) == (self.u_max is None)

    def _nominal_sweep_pays(self):
        """Where round() takes the sweep from the nominal by itself (measured,
        tools/nominal_round_time.py, 4096 trajectories): cartpole f32 1.37x
        the round on records, pendulum f32 1.04x; pendulum f64 0.95x and the
        double cartpole 0.65x (f32) / 0.74x (f64) - its record is ~1500
        instructions against a step of ~150, and a block of them in LDS leaves
        one workgroup per CU.  `sweep_nominal()` itself works wherever
        `_nominal_sweep_possible()` says so; set `_nominal_sweep = None` to
        make round() use it regardless."""
        return self.problem.model == 1 or (self.problem.model == 3 and
                                           self.dtype == torch.float32)

    @property
    def rec(self):
        """The derivative records [B][N+1][S] of the nominal, up to date."""
        self.sync_records()
        return self._rec

    @_on_device
    def sync_records(self):
        """Brings `rec` / `L` up to date with the nominal when round() left
        them behind (the sweep from the nominal writes no records)."""
        if self._rec_stale:
            p = _native.ptr
            _native.call("pddp_derivs", self.dtype, self._pp, self.B, self.N,
                         p(self.Z), p(self.U), p(self.u_min), p(self.u_max),
                         None, p(self._rec), p(self.L), p(self._J_scratch()),
                         None, self._s())
            self._rec_stale = False

    def _J_scratch(self):
        if getattr(self, "_jscr", None) is None:
            self._jscr = torch.zeros_like(self.J_opt)
        return self._jscr

    # -- views in the reference's tensor layout -----------------------------
    def record_views(self):
        """(F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu) as zero-copy views of the
        record buffer, shaped like ilqr.py:445-455 with a leading batch."""
        l, n, m, N = self.lay, self.n, self.m, self.N
        r = self.rec
        B = self.B
        F_z = r[:, :N, l.o_Fz:l.o_Fz + n * n].unflatten(-1, (n, n))
        F_u = r[:, :N, l.o_Fu:l.o_Fu + n * m].unflatten(-1, (n, m))
        L_z = r[:, :, l.o_Lz:l.o_Lz + n]
        L_u = r[:, :N, l.o_Lu:l.o_Lu + m]
        L_zz = r[:, :, l.o_Lzz:l.o_Lzz + n * n].unflatten(-1, (n, n))
        L_uz = r[:, :N, l.o_Luz:l.o_Luz + m * n].unflatten(-1, (m, n))
        L_uu = r[:, :N, l.o_Luu:l.o_Luu + m * m].unflatten(-1, (m, m))
        return F_z, F_u, L_z, L_u, L_zz, L_uz, L_uu

    def gain_views(self, accepted=False):
        g = self.gains_acc if accepted else self.gains
        m, n = self.m, self.n
        return g[..., :m], g[..., m:].unflatten(-1, (m, n))

    def exact_variant(self, branch=None, bounded=True):
        """The variant number (include/pddp_hip.h) of the kernel `auto` would
        pick, with IEEE division / square root instead of v_rcp / v_sqrt."""
        branch = self.branch if branch is None else branch
        bounded = bounded and self.u_min is not None
        if self.dtype != torch.float32:
            return 0  # the f64 kernels are IEEE throughout
        if self.n == 4 and self.m == 1:
            return 16 if self.B >= 12288 else 6
        if self.m == 1 and self.n <= 30:
            return 14
        return 0  # generic kernel: IEEE throughout

    # -- kernels --------------------------------------------------------------
    def _s(self):
        return _native.stream_handle(self.device)

    @_on_device
    def set_nominal(self, z0, U):
        """ilqr.py:274-277: new nominal controls, regularisation reset."""
        self.z0.copy_(z0.reshape(self.B, self.n))
        self.U.copy_(U.reshape(self.B, self.N, self.m))
        self.reset_controller_state()
        self.nominal_rollout()

    def reset_controller_state(self):
        self._derivs_due = True  # every nominal is new: records at round start
        self.n_live.zero_()
        self.mu.zero_()          # _reset_reg ilqr.py:364-367
        self.delta.fill_(2.0)
        self.state.zero_()       # UNDEFINED
        self.iter.fill_(1)       # first step() call is under way
        self.active.fill_(1)
        self.fresh.fill_(1)

    def _graphs_fresh(self):
        """Plugin graphs hold raw pointers to model-owned tensors
        (normalisation buffers, dropout masks, cached noise); `model.fit()`,
        `resample()` and loading a state replace those tensors and bump the
        model's generation (models/bnn.py).  A graph captured under another
        generation is dropped here and re-captured by its user."""
        if self.plugin is None:
            return
        from ..models.bnn import generation
        gen = generation(self.plugin.model)
        # (and the cost's tensors where launches read converted copies of
        # them: the GP line search, plugin.cost_generation)
        cg = getattr(self.plugin, "cost_generation", lambda: None)()
        if gen != self._model_gen or cg != getattr(self, "_cost_gen", cg):
            self._graph = None
            self._rollout_graph = None
            self._model_gen = gen
        self._cost_gen = cg

    @_on_device
    def nominal_rollout(self, mask=None):
        if self.plugin is not None:
            self._graphs_fresh()
            if self.graph_rollout and self.plugin.capture_ok(self):
                # the N + 1 moment-step / network launch pairs of the nominal
                # rollout as one hipGraph (z0, U, Z are solver-owned buffers)
                if self._rollout_graph is None:
                    self.plugin.rollout(self)  # warm: caches, attributes
                    torch.cuda.synchronize(self.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self.plugin.rollout(self)
                    self._rollout_graph = g
                self._rollout_graph.replay()
                return
            return self.plugin.rollout(self)
        p = _native.ptr
        _native.call("pddp_nominal_rollout", self.dtype, self._pp, self.B,
                     self.N, p(self.z0), p(self.U), p(self.u_min),
                     p(self.u_max), p(mask), p(self.Z), self._s())

    @_on_device
    def derivs(self, mask=None, set_state=True, in_graph=False):
        if self.plugin is not None:
            return self.plugin.derivs(self, mask, set_state, in_graph)
        if mask is not None:
            self.sync_records()  # (the rows outside the mask)
        self._rec_stale = False
        p = _native.ptr
        _native.call("pddp_derivs", self.dtype, self._pp, self.B, self.N,
                     p(self.Z), p(self.U), p(self.u_min), p(self.u_max),
                     p(mask), p(self._rec), p(self.L), p(self.J_opt),
                     p(self.state) if set_state else None, self._s())

    @_on_device
    def backward(self, active=None, reg=None, branch=None, bounded=True,
                 variant=0, events=None):
        """variant: 0 auto, 1 generic kernel, 2 / 3 specialised n=4 kernel
        (IEEE / approximate division), 6 / 7 closed-form BoxQP, 8 / 9 split
        over two wavefronts, see include/pddp_hip.h.  `events`: a
        (start, stop) pair of pddp_event handles to attach to the dispatch."""
        self.sync_records()
        p = _native.ptr
        reg = self.mu if reg is None else reg
        branch = self.branch if branch is None else branch
        umin = self.u_min if bounded else None
        umax = self.u_max if bounded else None
        args = (self.B, self.N, self.n, self.m, p(self._rec), p(umin), p(umax),
                p(reg), int(branch), p(active), p(self.gains),
                p(self.bwd_status), self._s(), int(variant))
        if events is None:
            _native.call("pddp_riccati_backward_variant", self.dtype, *args)
        else:
            _native.call("pddp_riccati_backward_timed", self.dtype, *args,
                         events[0], events[1])

    @_on_device
    def sweep_nominal(self, events=None):
        """Backward sweep straight from the nominal (pddp_sweep_nominal_f32):
        derivative records evaluated in the workgroups, stage costs to `L`,
        J_opt of the fresh nominals.  False when it does not apply."""
        p = _native.ptr
        if events is not None:
            _native.lib().pddp_attach_events(*events)
        rc = _native.call_rc(
            "pddp_sweep_nominal", self.dtype, self._pp, self.B, self.N,
            p(self.Z), p(self.U), p(self.u_min), p(self.u_max), p(self.mu),
            int(self.branch), p(self.active), p(self.fresh), p(self.gains),
            p(self.bwd_status), p(self.L), p(self.J_opt), self._s())
        if rc == _native.E_UNSUPPORTED:
            if events is not None:
                _native.lib().pddp_attach_events(None, None)
            self._nominal_sweep = False
            return False
        _native.check(rc, "pddp_sweep_nominal")
        self._nominal_sweep = True
        self._rec_stale = True
        return True

    @_on_device
    def round_nominal(self, tol, max_reg, n_iterations, events=None,
                      rounds=1):
        """A whole round in ONE launch (pddp_round_nominal_f32,
        csrc/round_n4.hip): the sweep from the nominal, then line search,
        accept and regularisation schedule in the same workgroups - the two
        launches' results (the sweep's bit for bit, the search's to rounding).
        `rounds` > 1: that many rounds back to back in the one launch (a
        workgroup owns its trajectories; no launch boundary between their
        attempts).  False when it does not apply (the caller then makes the two
        calls)."""
        if self.dtype != torch.float32 or self.plugin is not None or \
                self.u_min is None or self.u_max is None:
            self._one_launch = False
            return False
        if events is not None:
            _native.lib().pddp_attach_events(*events)
        # (the buffers are the solver's own for its lifetime: their addresses
        # are looked up once - a launch of this entry point is the whole host
        # side of a round, and 27 data_ptr() calls were a third of it)
        key = (self._rec.data_ptr(), self.Z.data_ptr(), self.U.data_ptr(),
               self.gains.data_ptr(), int(self.branch))
        if self._round_args is None or self._round_args[0] != key:
            p = _native.ptr
            self._round_args = (key, (
                self._pp, self.B, self.N, self.A, p(self.Z), p(self.U),
                p(self.alphas), p(self.u_min), p(self.u_max), int(self.branch),
                p(self.active), p(self.fresh), p(self.gains),
                p(self.bwd_status), p(self.L), p(self.J_opt), p(self.Zc),
                p(self.Uc), p(self.Jc)), (
                p(self.gains_acc), p(self.mu), p(self.delta), p(self.state),
                p(self.iter), p(self.n_live), p(self._rec)))
        _, head, tail = self._round_args
        rc = _native.lib().pddp_round_nominal_f32(
            *head, float(tol), float(max_reg), int(n_iterations), *tail,
            int(rounds), _native.ptr(self.phase_ticks), self._s())
        if rc == _native.E_UNSUPPORTED:
            if events is not None:
                _native.lib().pddp_attach_events(None, None)
            self._one_launch = False
            return False
        _native.check(rc, "pddp_round_nominal_f32")
        self._one_launch = True
        self._nominal_sweep = True
        self._fused = True
        self._rec_stale = True
        self._derivs_due = False
        return True

    @_on_device
    def line_search(self, active=None, use_status=True):
        if self.plugin is not None:
            return self.plugin.line_search(self, active, use_status)
        p = _native.ptr
        _native.call("pddp_line_search", self.dtype, self._pp, self.B, self.N,
                     self.A, p(self.Z), p(self.U), p(self.gains),
                     p(self.alphas), p(self.u_min), p(self.u_max), p(active),
                     p(self.bwd_status) if use_status else None, p(self.Zc),
                     p(self.Uc), p(self.Jc), self._s())

    @_on_device
    def accept(self, tol, max_reg, n_iterations):
        p = _native.ptr
        _native.call("pddp_accept", self.dtype, self.B, self.N, self.n, self.m,
                     self.A, p(self.Zc), p(self.Uc), p(self.Jc), p(self.gains),
                     p(self.bwd_status), float(tol), float(max_reg),
                     int(n_iterations), p(self.Z), p(self.U),
                     p(self.gains_acc), p(self.J_opt), p(self.mu),
                     p(self.delta), p(self.state), p(self.iter),
                     p(self.active), p(self.fresh), p(self.n_live), self._s())

    @_on_device
    def search_accept(self, tol, max_reg, n_iterations, events=None,
                      records=True):
        """Line search + accept + derivative records of the new nominals in
        one launch (pddp_search_accept_*).  False when the fused kernel does
        not apply; the caller then makes the separate calls.  `events`: a
        (start, stop) pair attached to THIS launch (bench.py); when the fused
        kernel does not apply nothing is attached and `last_search_timed` says
        so."""
        self.last_search_timed = None
        if self.plugin is not None or self._fused is False:
            return False
        p = _native.ptr
        if events is not None:
            _native.lib().pddp_attach_events(*events)
            self.last_search_timed = "search_accept"
        rc = _native.call_rc(
            "pddp_search_accept", self.dtype, self._pp, self.B, self.N, self.A,
            p(self.Z), p(self.U), p(self.gains), p(self.alphas), p(self.u_min),
            p(self.u_max), p(self.active), p(self.bwd_status), p(self.Zc),
            p(self.Uc), p(self.Jc), float(tol), float(max_reg),
            int(n_iterations), p(self.gains_acc), p(self.J_opt), p(self.mu),
            p(self.delta), p(self.state), p(self.iter), p(self.fresh),
            p(self.n_live), p(self._rec), p(self.L) if records else None,
            self._s())
        self._fused = rc == 0
        if self._fused:
            # (pddp_search_candidates: without records the candidates are
            # dropped beyond 200 MB - Zc / Uc are scratch after such a launch,
            # only Jc and the nominal are results)
            mode = _native.lib().pddp_search_candidates(-1)
            nbytes = float(self.B) * self.A * (
                (self.N + 1) * self.n + self.N * self.m) * \
                self.Z.element_size()
            self.candidates_kept = records or self._rec is None or not (
                mode == 2 or (mode == 0 and nbytes > 200e6))
        if not self._fused and events is not None:
            _native.lib().pddp_attach_events(None, None)  # nothing launched
            self.last_search_timed = None
        return self._fused

    def rounds(self, count, tol=5e-6, max_reg=1e10, n_iterations=50,
               events=None):
        """`count` rounds; in one launch where pddp_round_nominal_f32 applies
        (`events` are then attached to that launch), `count` calls of round()
        otherwise."""
        if count > 1 and self._one_launch is not False and \
                self.kernel_variant == 0 and \
                self._nominal_sweep is not False and \
                self._fused is not False and \
                self.round_nominal(tol, max_reg, n_iterations, events=events,
                                   rounds=count):
            return
        for _ in range(count):
            self.round(tol, max_reg, n_iterations)

    def round(self, tol=5e-6, max_reg=1e10, n_iterations=50, variant=None,
              backward_events=None, always_derivs=False, search_events=None):
        """One attempt of every live trajectory (no host sync): derivative
        records of the trajectories whose nominal is new, backward sweep, and
        - fused into one launch where the problem allows - line search,
        accept / regularisation schedule and the records of the accepted
        nominals (so the first call is a no-op from the second round on)."""
        if variant is None:
            variant = self.kernel_variant
        if self._one_launch is not False and variant == 0 and \
                self._nominal_sweep is not False and \
                self._fused is not False and search_events is None and \
                self.round_nominal(tol, max_reg, n_iterations,
                                   events=backward_events):
            return
        if self._nominal_sweep is not False and variant == 0 and \
                self._fused is not False and \
                self.sweep_nominal(events=backward_events):
            # records evaluated inside the sweep; the fused launch then writes
            # none (`fresh` stays set until the next sweep has summed the
            # stage costs of the new nominal into J_opt)
            self._derivs_due = False
            if self.search_accept(tol, max_reg, n_iterations,
                                  events=search_events, records=False):
                return
            # (the fused launch does not apply: > 16 step sizes) the separate
            # calls; records by the masked derivs launch from now on
            self._nominal_sweep = False
            self.line_search(active=self.active)
            self.accept(tol, max_reg, n_iterations)
            self._derivs_due = True
            return
        if self._rec_stale:
            self.sync_records()
        if self._derivs_due or not self._fused or always_derivs:
            self.derivs(mask=self.fresh)
            self._derivs_due = False
        self.backward(active=self.active, variant=variant,
                      events=backward_events)
        if not self.search_accept(tol, max_reg, n_iterations,
                                  events=search_events):
            if search_events is not None and self.plugin is None:
                # the separate line search is what gets timed then
                _native.lib().pddp_attach_events(*search_events)
                self.last_search_timed = "line_search"
            self.line_search(active=self.active)
            self.accept(tol, max_reg, n_iterations)

    def graph_ok(self):
        """A round can be captured: native sample problem, or a plugin whose
        round is all HIP launches and sync-free torch ops (the BNN path)."""
        return self.plugin is None or self.plugin.capture_ok(self)

    _STATE = ("Z", "U", "_rec", "L", "J_opt", "gains", "gains_acc", "Jc",
              "bwd_status", "state", "iter", "mu", "delta", "active", "fresh",
              "n_live")

    @_on_device
    def capture_round(self, tol=5e-6, max_reg=1e10, n_iterations=50):
        """Captures round() - a fixed launch sequence on device-resident state
        - into a hipGraph; `replay_round()` then issues it with one launch.
        For the launch-bound regime: small batches and the receding-horizon
        loop (BASELINE.json configs[4]).

        Native sample problems: one graph (the masked records launch is part
        of it).  Plugin (BNN) rounds: two graphs sharing one memory pool - the
        round with the derivative rollout (every record recomputed, rows
        blended by the `fresh` mask) and the round of retries only (ilqr.py
        :125-139 with a larger mu: nominals, hence records, unchanged); fit()
        picks one per round from the counts it reads back anyway."""
        if not self.graph_ok():
            raise _native.NativeError(
                "graph capture needs a round without host synchronisation: "
                "this plugin runs autograd / torch fallbacks inside a round")
        self._graphs_fresh()
        key = (float(tol), float(max_reg), int(n_iterations),
               self.kernel_variant)
        if self._graph is not None and self._graph[0] == key:
            return self._graph[1]
        torch.cuda.synchronize(self.device)
        if self.plugin is None:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                # the masked records launch is always part of the graph (a
                # no-op when no nominal is fresh): a replay after
                # set_nominal() must not sweep the previous nominal's records
                self.round(tol, max_reg, n_iterations, always_derivs=True)
            self._graph = (key, graph, None)
            self._graph_nominal = self._nominal_sweep is True
            return graph
        # warm-up outside the capture (noise caches, masks, per-kernel
        # attributes, allocator) on a snapshot of the solver's state
        snap = {k: getattr(self, k).clone() for k in self._STATE}
        self._plugin_round(tol, max_reg, n_iterations, True, False)
        for k, v in snap.items():
            getattr(self, k).copy_(v)
        torch.cuda.synchronize(self.device)
        g_full = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_full):
            self._plugin_round(tol, max_reg, n_iterations, True, True)
        g_retry = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_retry, pool=g_full.pool()):
            self._plugin_round(tol, max_reg, n_iterations, False, True)
        self._graph = (key, g_full, g_retry)
        return g_full

    def _plugin_round(self, tol, max_reg, n_iterations, with_derivs, in_graph):
        if with_derivs:
            self.derivs(mask=self.fresh, in_graph=in_graph)
        self.backward(active=self.active, variant=self.kernel_variant)
        self.line_search(active=self.active)
        self.accept(tol, max_reg, n_iterations)

    @_on_device
    def replay_round(self, with_derivs=True):
        g = self._graph[1] if (with_derivs or self._graph[2] is None) \
            else self._graph[2]
        if getattr(self, "_graph_nominal", False):
            self._rec_stale = True  # the captured sweep writes no records
        g.replay()

    def fit(self, n_iterations=50, tol=5e-6, max_reg=1e10, on_round=None,
            max_rounds=None, graph=False, rounds_per_sync=1,
            rounds_per_launch=1):
        """Runs rounds until every trajectory left the fit loop
        (ilqr.py:298-314). Returns the number of rounds.  With `graph=True`
        rounds are hipGraph replays and the host looks at the live count only
        every `rounds_per_sync` rounds (a round with nothing live is a no-op
        on the device, so the result does not depend on it).
        `rounds_per_launch` > 1 (no `on_round`, no graph): that many rounds per
        launch where pddp_round_nominal_f32 applies (`rounds()`), the live
        count read after each launch - the same results, up to
        rounds_per_launch - 1 no-op rounds more."""
        if graph:
            self.capture_round(tol, max_reg, n_iterations)
            if self.plugin is not None:
                rounds_per_sync = 1  # which graph comes next is read back
        rpl = 1 if (graph or on_round is not None) else max(1, rounds_per_launch)
        rounds = 0
        need_derivs = True
        while True:
            if graph:
                self.replay_round(need_derivs)
            elif rpl > 1 and self._one_launch is True:
                # (only where the one-launch round has applied: elsewhere
                # rounds() is a loop of round() calls that would run past the
                # last live trajectory)
                c = rpl if max_rounds is None else min(rpl, max_rounds - rounds)
                self.rounds(c, tol, max_reg, n_iterations)
                rounds += c - 1
            else:
                self.round(tol, max_reg, n_iterations)
            rounds += 1
            if on_round is not None:
                on_round(rounds, self)
            if max_rounds is not None and rounds >= max_rounds:
                break
            if not (rpl > 1 and self._one_launch is True) and \
                    rounds % rounds_per_sync:
                continue
            # the one host sync: live trajectories, and (plugin graphs) whether
            # any nominal changed
            live, fresh = torch.stack([self.active.sum(),
                                       self.fresh.sum()]).tolist()
            need_derivs = fresh > 0
            if live == 0:
                break
        return rounds
