"""CPU-only tests: the C-ABI library loads and exports every declared symbol,
host-side plugin API (encodings, costs, models vs the oracle / goldens), the
loud failure without a GPU, and the multi-rank best-rollout exchange over
gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle as orc
from golden_util import DT, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_library_exports_every_declared_symbol():
    from pddp_amd import _native
    hdr = open(os.path.join(ROOT, "include", "pddp_hip.h")).read()
    declared = set(re.findall(r"\b(pddp_[a-z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.startswith("pddp_record_layout")
                or d == "pddp_record_layout_of"}
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(_native.exported_symbols()) >= declared
    assert _native.lib().pddp_hip_abi_version() == 1
    lay = _native.record_layout(4, 1)
    assert (lay.stride, lay.gain_stride, lay.o_U) == (48, 5, 46)
    lay = _native.record_layout(27, 1)
    assert lay.stride % 4 == 0 and lay.stride >= 2 * 27 * 27 + 3 * 27 + 3


def test_no_new_spilling_kernel():
    """Register spills / scratch of every kernel in the built library against
    the reviewed list (profiles/kernel_resources_allowed.csv): a kernel that
    starts to spill, or spills more, fails here - on the CPU, from the code
    object's metadata (tools/kernel_resources.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import kernel_resources as kr
    finally:
        sys.path.pop(0)
    rows = kr.kernel_resources()
    assert len(rows) > 100  # (the metadata was found)
    bad = kr.check(rows)
    assert not bad, bad


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing."""
    import pddp_amd
    from pddp_amd import _native
    from pddp_amd.examples import cartpole
    env = cartpole.CartpoleEnv()
    ctrl = pddp_amd.controllers.iLQRController(
        env, cartpole.CartpoleDynamicsModel(0.1), cartpole.CartpoleCost())
    with pytest.raises(_native.NativeError):
        ctrl.fit(torch.zeros(5, 1),
                 encoding=pddp_amd.StateEncoding.IGNORE_UNCERTAINTY)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: nothing under pddp_amd/ may mention it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pddp_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and \
                    "from oracle" not in text and \
                    "pddp_oracle" not in text, f


@pytest.mark.parametrize("problem", ["cartpole", "pendulum",
                                     "double_cartpole", "rendezvous"])
def test_host_models_and_costs_match_oracle(problem):
    """The torch plugin classes (used by envs / user scripts) agree with the
    oracle on dynamics and cost values, and carry the reference's constants."""
    import pddp_amd
    from pddp_amd.utils.encoding import StateEncoding
    enc = StateEncoding.IGNORE_UNCERTAINTY
    mod = getattr(pddp_amd.examples, problem)
    model = [getattr(mod, n) for n in dir(mod) if n.endswith("DynamicsModel")
             and n != "DynamicsModel"][0](DT[problem]).double()
    cost = [getattr(mod, n) for n in dir(mod) if n.endswith("Cost")
            and n != "AugmentedQRCost"][0]().double()
    prob = model.native_problem(enc, cost)
    op = orc.make_problem(problem, DT[problem])
    for f in ("params", "Q", "Q_term", "R", "x_goal", "u_goal"):
        assert list(getattr(prob, f)) == list(getattr(op, f)), f
    o = orc.load(np.float64)
    rng = np.random.RandomState(0)
    n, m = op.encoded_size, op.action_size
    for _ in range(5):
        z, u = rng.randn(n), rng.randn(m)
        zn, _, _ = o.dynamics(op, z, u)
        got = model(torch.from_numpy(z), torch.from_numpy(u), 0, enc)
        assert np.allclose(got.detach().numpy(), zn, rtol=1e-12, atol=1e-12)
        l = o.cost(op, z, u)[0]
        gl = cost(torch.from_numpy(z), torch.from_numpy(u), 0, False, enc)
        assert np.allclose(float(gl), l, rtol=1e-12)
        lt = o.cost(op, z, None, terminal=True)[0]
        glt = cost(torch.from_numpy(z), None, 0, True, enc)
        assert np.allclose(float(glt), lt, rtol=1e-12)
    # batched evaluation == row-wise evaluation (reference test strategy:
    # tests/utils/test_evaluation.py loop-vs-batch)
    Zb, Ub = torch.randn(7, n).double(), torch.randn(7, m).double()
    out = model(Zb, Ub, 0, enc)
    for r in range(7):
        assert torch.allclose(out[r], model(Zb[r], Ub[r], 0, enc))


def test_encoding_sizes_and_round_trip():
    """tests/utils/test_encoding.py of the reference: sizes 30/20/10/10/5 for
    D = 5 and encode . decode round trips."""
    from pddp_amd.utils import encoding as E
    from pddp_amd import GaussianVariable, StateEncoding
    sizes = {StateEncoding.FULL_COVARIANCE_MATRIX: 30,
             StateEncoding.UPPER_TRIANGULAR_CHOLESKY: 20,
             StateEncoding.VARIANCE_ONLY: 10,
             StateEncoding.STANDARD_DEVIATION_ONLY: 10,
             StateEncoding.IGNORE_UNCERTAINTY: 5}
    torch.manual_seed(0)
    for enc, size in sizes.items():
        assert E.infer_encoded_state_size(5, enc) == size
        assert E.infer_state_size(size, enc) == 5
        x = GaussianVariable.random(5, dtype=torch.float64)
        z = x.encode(enc)
        assert z.shape == (size,)
        assert torch.allclose(E.decode_mean(z, enc), x.mean())
        if enc in (StateEncoding.FULL_COVARIANCE_MATRIX,
                   StateEncoding.UPPER_TRIANGULAR_CHOLESKY):
            assert torch.allclose(E.decode_covar(z, enc), x.covar(),
                                  atol=1e-9)
            L = E.decode_covar_sqrt(z, enc)
            assert torch.allclose(L.t() @ L, x.covar(), atol=1e-9)
        if enc != StateEncoding.IGNORE_UNCERTAINTY:
            assert torch.allclose(E.decode_var(z, enc), x.var(), atol=1e-9)
            assert torch.allclose(E.decode_std(z, enc), x.std(), atol=1e-9)
        # batched
        zb = torch.stack([z, z])
        assert E.decode_mean(zb, enc).shape == (2, 5)
        assert E.decode_covar(zb, enc).shape == (2, 5, 5)


def test_default_encoding_matches_reference_golden():
    """z0 under the DEFAULT (upper-triangular Cholesky) encoding equals the
    reference's (golden)."""
    from pddp_amd import GaussianVariable, StateEncoding
    g = load("cartpole", encoding="default")
    mean = torch.tensor([0.01, -0.02, 0.015, 0.0], dtype=torch.float64)
    z0 = GaussianVariable(mean, var=1e-2 * torch.ones_like(mean)).encode(
        StateEncoding.DEFAULT)
    assert np.allclose(z0.numpy(), g["z0"], rtol=1e-12, atol=1e-15)


def test_augment_reduce_round_trip():
    from pddp_amd.utils.angular import augment_state, reduce_state
    from pddp_amd.examples.double_cartpole import DoubleCartpoleDynamicsModel \
        as M
    x = torch.randn(9, 6).double()
    xa = augment_state(x, M.angular_indices, M.non_angular_indices)
    assert xa.shape == (9, 8)
    xr = reduce_state(xa, M.angular_indices, M.non_angular_indices)
    d = (xr - x)
    d[:, M.angular_indices] = torch.remainder(
        d[:, M.angular_indices] + np.pi, 2 * np.pi) - np.pi
    assert d.abs().max() < 1e-12


def test_shard_bounds_cover_batch():
    from pddp_amd.parallel import shard_bounds
    for total in (1, 7, 4096, 8192, 8191):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from pddp_amd.parallel import gather_best_rollout, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
B, N, n, m = 10, 4, 3, 2
g = torch.Generator().manual_seed(5)
J = torch.rand(B, generator=g, dtype=torch.float64) + 1.0
J[3] = float("nan")          # a diverged trajectory never wins
J[7] = 0.25                  # global best lives on the last rank
Z = torch.arange(B * (N + 1) * n, dtype=torch.float64).view(B, N + 1, n)
U = -torch.arange(B * N * m, dtype=torch.float64).view(B, N, m)
lo, hi = shard_bounds(B, rank, world)
Jb, idx, Zb, Ub = gather_best_rollout(J[lo:hi], Z[lo:hi], U[lo:hi], offset=lo)
assert idx == 7 and float(Jb) == 0.25, (idx, float(Jb))
assert torch.equal(Zb, Z[7]) and torch.equal(Ub, U[7])
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_best_rollout_all_gather_two_ranks(tmp_path):
    """world_size-2 gloo run of the one collective on the path."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
         "29561", str(script)], capture_output=True, text=True, env=env,
        timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


_WORKER8 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from pddp_amd.parallel import gather_best_rollout, shard_bounds
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
# a ragged batch: 8 ranks over 29 trajectories (shards of 4 and 3), the rounds
# of a fit one after another with the winner moving between ranks
B, N, n, m = 29, 5, 4, 1
lo, hi = shard_bounds(B, rank, world)
assert 3 <= hi - lo <= 4
g = torch.Generator().manual_seed(11)
for it in range(6):
    J = torch.rand(B, generator=g, dtype=torch.float64) + 1.0
    J[(5 * it + 2) %% B] = float("nan")      # diverged: never wins
    best = (7 * it + 3) %% B
    J[best] = 0.125 + 0.0625 * it
    Z = torch.rand(B, N + 1, n, generator=g, dtype=torch.float64)
    U = torch.rand(B, N, m, generator=g, dtype=torch.float64)
    Jb, idx, Zb, Ub = gather_best_rollout(J[lo:hi], Z[lo:hi], U[lo:hi],
                                          offset=lo)
    assert idx == best and float(Jb) == 0.125 + 0.0625 * it, (it, idx)
    assert torch.equal(Zb, Z[best]) and torch.equal(Ub, U[best])
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_best_rollout_all_gather_eight_ranks_ragged(tmp_path):
    """BASELINE configs[3]'s world size on the CPU: eight gloo ranks, a batch
    that does not divide by eight, six exchanges with the winner on a
    different rank each time."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8 % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node=8", "--master-addr", "127.0.0.1", "--master-port",
         "29571", str(script)], capture_output=True, text=True, env=env,
        timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 8


def test_oracle_under_address_and_ub_sanitizers():
    """SURVEY 5 (sanitizers run on the CPU side): the oracle's golden suite
    re-run in a child interpreter against the -fsanitize=address,undefined
    build of the same C restatement; any report aborts the child."""
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"],
                          capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"),
                           "asan"])
    lib = os.path.join(ROOT, "oracle", "_build", "libpddp_oracle_asan.so")
    env = dict(os.environ, LD_PRELOAD=asan, PDDP_ORACLE_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run(
        [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
         os.path.join(ROOT, "tests", "test_oracle_golden.py")],
        env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    assert "passed" in out.stdout


def test_bench_gpus_flag_is_honoured():
    """`bench.py --gpus N` never silently runs another world size: a torchrun
    environment of a different size, or fewer visible devices than asked,
    exits non-zero before anything touches the GPU."""
    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr
    if torch.cuda.device_count() < 2:
        env = {k: v for k, v in os.environ.items()
               if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        out = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env,
                             capture_output=True, text=True, timeout=300)
        assert out.returncode == 2 and "device(s) visible" in out.stderr


def test_best_rollout_single_process():
    from pddp_amd.parallel import gather_best_rollout
    J = torch.tensor([3.0, float("inf"), 1.5])
    Z = torch.randn(3, 5, 2)
    U = torch.randn(3, 4, 1)
    Jb, idx, Zb, Ub = gather_best_rollout(J, Z, U, offset=100)
    assert idx == 102 and float(Jb) == 1.5
    assert torch.allclose(Zb, Z[2]) and torch.allclose(Ub, U[2])
    # the sync-free form (one exchange per iteration): the index stays a tensor
    Jb2, idx2, Zb2, Ub2 = gather_best_rollout(J, Z, U, offset=100, sync=False)
    assert isinstance(idx2, torch.Tensor) and int(idx2) == 102
    assert torch.equal(Zb2, Zb) and float(Jb2) == 1.5
    # a copy, not a view of anything that a later exchange overwrites
    assert not Zb2._is_view() or Zb2._base is not Z
    assert Zb2.data_ptr() != Z[2].data_ptr()
    # the index travels in the run's dtype: refused when it would not be exact
    with pytest.raises(ValueError):
        gather_best_rollout(J, Z, U, offset=1 << 24)


@pytest.mark.parametrize("problem", ["cartpole", "pendulum"])
def test_plugin_derivatives_match_reference_default_encoding(problem):
    """DEFAULT (upper-triangular Cholesky) encoding, n = 14 / 5: the plugin
    path's autograd derivatives (replicated-input trick) of OUR models / costs
    (moment-matched angle augmentation, tr(Q Sigma) term) against the
    reference's own forward() outputs (golden, fp64).  Device-agnostic code,
    checked here on CPU tensors."""
    import pddp_amd
    from pddp_amd import StateEncoding
    from pddp_amd.controllers.plugin import TorchProblem
    mod = getattr(pddp_amd.examples, problem)
    model = [getattr(mod, n) for n in dir(mod) if n.endswith("DynamicsModel")
             and n != "DynamicsModel"][0](DT[problem]).double()
    cost = [getattr(mod, n) for n in dir(mod) if n.endswith("Cost")
            and n != "AugmentedQRCost"][0]().double()
    g = load(problem, encoding="default")
    tp = TorchProblem(model, cost, StateEncoding.DEFAULT)
    for tag in ("N5_cos", "N25_cos"):
        U = torch.from_numpy(g[tag + "/U"])
        Z = torch.from_numpy(g[tag + "/fwd/Z"])
        N = U.shape[0]
        # batched over time: every (z_t, u_t) pair is one row
        l, lz, lu, lzz, luz, luu = tp._cost_derivs(Z[:N], U, 0, False)
        Fz, Fu = tp._dyn_derivs(Z[:N], U, 0)
        zn = model(Z[:N], U, 0, StateEncoding.DEFAULT)
        for got, nm in ((l, "L"), (lz, "L_z"), (lu, "L_u"), (lzz, "L_zz"),
                        (luz, "L_uz"), (luu, "L_uu"), (Fz, "F_z"),
                        (Fu, "F_u")):
            ref = g["%s/fwd/%s" % (tag, nm)][:N]
            assert np.allclose(got.detach().numpy(), ref, rtol=1e-9,
                               atol=1e-11), (tag, nm)
        assert np.allclose(zn.detach().numpy(), g[tag + "/fwd/Z"][1:],
                           rtol=1e-10, atol=1e-12)
        lt, lzt, _, lzzt, _, _ = tp._cost_derivs(Z[N:], None, N - 1, True)
        assert np.allclose(lzzt[0].numpy(), g[tag + "/fwd/L_zz"][N],
                           rtol=1e-9, atol=1e-11)
        assert np.allclose(lzt[0].numpy(), g[tag + "/fwd/L_z"][N], rtol=1e-9,
                           atol=1e-11)


def test_pddp_dataset_helpers():
    """pddp.py:209-267: trial data plumbing (device-agnostic host code)."""
    import pddp_amd
    from pddp_amd.controllers.pddp import _apply_controller, _concat_datasets
    from pddp_amd.examples import pendulum
    a = (torch.arange(6.).view(3, 2), torch.zeros(3, 1), torch.ones(3, 2))
    b = (torch.arange(6., 10.).view(2, 2), torch.ones(2, 1), torch.ones(2, 2))
    assert _concat_datasets(None, a) is a and _concat_datasets(a, None) is a
    X, U, dX = _concat_datasets(a, b, max_dataset_size=4)
    assert X.shape == (4, 2) and float(X[0, 0]) == 2.0  # keeps the LAST rows
    np.random.seed(0)
    env = pendulum.PendulumEnv(dt=0.1)
    cost = pendulum.PendulumCost()
    enc = pddp_amd.StateEncoding.IGNORE_UNCERTAINTY
    Uo = 0.5 * torch.ones(7, 1)
    (X, U, dX), J = _apply_controller(env, cost, Uo, 7, enc)
    assert X.shape == (7, 2) and U.shape == (7, 1) and dX.shape == (7, 2)
    assert torch.equal(U, Uo) and J.dim() == 0
    # the env stepped its own ground-truth model: dX is consistent with X
    model = pendulum.PendulumDynamicsModel(0.1)
    for t in range(6):
        xn = model(X[t], U[t], 0, enc)
        assert torch.allclose(xn, X[t + 1], atol=1e-6)
        assert torch.allclose(dX[t], X[t + 1] - X[t])


def test_hot_kernels_keep_their_working_set_in_registers():
    """No scratch memory and no register spills in the kernels the benchmarks
    run (read from the built library's code-object metadata, tools/
    kernel_resources.py): a small matrix indexed through a runtime dimension
    silently moves to scratch - that cost four BNN kernels 2 .. 8x before it
    was seen.  Kernels known to use scratch are listed."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools"))
    from kernel_resources import kernel_resources
    from pddp_amd import _native
    _native.build()  # (incremental; a no-op when the library is up to date)
    rows = kernel_resources()
    assert len(rows) > 200
    allowed = (
        "accept_kernel<",                       # 36 B, one launch per round of the un-fused path
        # runtime-D fall-backs (D not 2 / 4 / 6), float and double
        "bnn_jvp_features_kernel<float, 6, false>",
        "bnn_jvp_features_kernel<double, 4, false>",
        "bnn_jvp_features_kernel<double, 6, false>",
        "bnn_jvp_moments_kernel<float, 16, 4, false>",
        "bnn_jvp_moments_kernel<float, 32, 6, false>",
        "bnn_jvp_moments_kernel<double, 16, 4, false>",
        "bnn_jvp_moments_kernel<double, 32, 6, false>",
        # the double cartpole's cost in hyper-dual DOUBLES (27 + 1 inputs, two
        # angles): 96 registers spill at the 512-register file; one launch per
        # round of a float64 BNN run
        "qr_cost_derivs_kernel<double, 6, 2>",
        # the float64 network kernel in JVP mode at the 512-register file
        # (300 of them W2): ten values spill in the prologue, none in the
        # tile loop
        "bnn_mlp_f64_kernel<200, 8,", "bnn_mlp_f64_kernel<200, 16,",
        "derivs_default_kernel<double, 2, 1>",  # double cartpole, hyper-dual on 27 inputs
        "derivs_default_kernel<float, 2, 1>",
        # rendezvous, Cholesky encoding, fp64: the 36-entry re-factorisation in
        # dual numbers at the 512-register file, two values parked in
        # accumulation registers (no scratch); not on a benchmarked path
        "derivs_default_kernel<double, 4, 1>",
        "line_search_default_kernel<float, 2, 2>",
        "line_search_kernel<double, 2>",
        "line_search_lds_kernel<double, 2, true, 4, 2,",
        # (round 5: the search body became a device function shared with
        # round_n4.hip; the double cartpole's f32 instantiation then keeps 68
        # bytes - no spilled register - in scratch; its round is 180.6 us
        # against 181.4 before, tools/nominal_round_time.py)
        "line_search_lds_kernel<float, 2, true, 4, 2,",
        "line_search_lds_kernel<double, 4, true, 4, 2,",
        # one-wavefront workgroups: horizons whose nominal data does not fit
        # four times into 64 KB of LDS (N > 400 for these two problems)
        "line_search_lds_kernel<double, 4, true, 1, 1,",
        "line_search_lds_kernel<float, 2, true, 1, 1,",
        # the dense form (8193 .. 49152 trajectories): eight rows per lane in
        # the tail's short form at the 128-register cap of four workgroups per
        # CU - 7 .. 9 registers spill there, none in the rollouts' loop
        "line_search_lds_kernel<float, 1, true, 4, 1,",
        # the GP line search's kernel, held to 168 registers for three
        # workgroups per CU (0.81 against 1.02 ms per launch): 9 registers
        # spilled in the per-row front end
        "gp_step_fwd_f32_kernel<6, 9>",
        "gp_roll_f32_kernel<6, 9>",             # the same body behind the rollout's front end
        # (fp64 Jacobian kernel at the 512-register file: ten values parked
        # in accumulation registers, no scratch)
        "gp_step_kernel<double, 6, 9, true>",
        # the bf16-split twin of the network kernel (opt-in): 156 registers of
        # W2 parts at the 256-register limit of an eight-wave workgroup; 5 ..
        # 25 registers spill around (not inside) the matrix-instruction loop
        # (with the requests made a tile ahead: 2.35 -> 2.26 ms all the same)
        ", 0, 3>",
    )
    bad = [(r["kernel"], r.get("private_segment_fixed_size", 0),
            r.get("vgpr_spill_count", 0)) for r in rows
           if (r.get("private_segment_fixed_size", 0) > 0
               or r.get("vgpr_spill_count", 0) > 0)
           and not any(a in r["kernel"] for a in allowed)]
    assert not bad, bad
    hot = ("round_n4_kernel<25u, true>", "round_n4_kernel<25u, false>",
           "riccati_n4_quad_kernel<float",
           "line_search_lds_kernel<float, 1, true, 4, 2, 25u, false>",
           "riccati_n4_elem_kernel<25u, true>", "riccati_n4_elem_kernel<25u, false>",
           "riccati_n4_elem_f64_kernel<25u>",
           "bnn_mlp_kernel<200", "riccati_mfma16_kernel<", "riccati_mfma32_kernel<",
           "bnn_mlp_f64_kernel<200", "bnn_moment_step_kernel<float, 4>",
           "bnn_moment_step_kernel<float, 6>", "bnn_moment_step_kernel<double, 4>",
           "bnn_jvp_moments_kernel<float, 16, 4, true>",
           "bnn_jvp_moments_kernel<double, 16, 4, true>",
           "qr_cost_derivs_kernel<float", "qr_cost_derivs_kernel<double, 4, 1>")
    for h in hot:
        assert any(h in r["kernel"] for r in rows), h
