#!/usr/bin/env python3
"""Headline benchmark: batched iLQR iterations on MI355X.

Workload = BASELINE.json configs[1]: cartpole (n=4, m=1), known dynamics,
horizon N=100, B=4096 trajectories per GPU, fp32, control bounds +-10
(controller default branch: eig-clamp + BoxQP), alphas of ilqr.py:282.
Synthetic inputs as SURVEY.md 8(d): z0 = 1e-2 N(0,1), U = 0.1 N(0,1),
seed = rank.

A "step" is one pass of the hot path over the whole batch: derivative records
for every trajectory whose nominal changed, the backward Riccati sweep, the
10-candidate line search with costs, and the accept / regularisation update.
Inputs are resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W

prints ONE JSON line (rank 0).  With --gpus N > 1 and no torchrun environment
the script starts N ranks itself (a fresh `python -m torch.distributed.run`
child, before this process has touched the GPU) and relays its line; under the
driver's own torchrun launch it simply is one of the ranks.  Ranks own disjoint
shards (--scaling weak: --batch trajectories per GPU; strong: --batch in total;
no data-path collective) and exchange their best rollout through RCCL after
every launch of a round (--exchange-every E: every E rounds; 0: once per timed
region; the cartpole's rounds come ten to a launch: every ten) - one
`all_gather_into_tensor` of a fixed-size record, no host synchronisation.

The timed region (W warm-up rounds, then exactly K rounds between barrier +
synchronize pairs) is repeated --repeats times from the same nominal; the line
reports the MEDIAN repetition (and the minimum and every repetition beside it).
Two further repetitions of the same K rounds carry HIP events on the sweep's and
the line search's dispatches: the per-kernel durations of `roofline` come from
those (`ms_per_step_with_kernel_events` shows what the events cost).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cpu_baseline_worker(problem_name, dt, N, bound, seconds, seed):
    """One core's share of the CPU baseline: oracle fits until the time is up.
    Returns (attempts, trajectories, elapsed)."""
    import oracle as orc
    o = orc.load(np.float32)
    op = orc.make_problem(problem_name, dt)
    rng = np.random.RandomState(seed)
    alphas = (1.025 ** (-np.arange(10.0) ** 2)).astype(np.float32)
    u_min = np.full(op.action_size, -bound, np.float32)
    u_max = np.full(op.action_size, bound, np.float32)
    attempts, trajs = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        z0 = (1e-2 * rng.randn(op.encoded_size)).astype(np.float32)
        U = (0.1 * rng.randn(N, op.action_size)).astype(np.float32)
        _, _, _, _, trace = o.fit(op, z0, U, alphas, n_iterations=8,
                                  u_min=u_min, u_max=u_max)
        attempts += trace.shape[0]
        trajs += 1
    return attempts, trajs, time.perf_counter() - t0



if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-worker":
    # a CPU-baseline worker: no torch, no GPU (see cpu_baseline below)
    _name, _dt, _N, _bound, _seconds, _seed = sys.argv[2:8]
    print("%d %d %.6f" % _cpu_baseline_worker(
        _name, float(_dt), int(_N), float(_bound), float(_seconds), int(_seed)))
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-torch-worker":
    # the second CPU leg: the op-for-op PyTorch-CPU restatement,
    # one trajectory at a time like the reference, one thread; CPU tensors only
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    import torch as _t
    _t.set_num_threads(1)
    from oracle import torch_port as _tp
    import pddp_amd as _pa
    from pddp_amd.examples import cartpole as _cp
    _dt, _N, _bound, _seconds = (float(sys.argv[2]), int(sys.argv[3]),
                                 float(sys.argv[4]), float(sys.argv[5]))
    _model, _cost = _cp.CartpoleDynamicsModel(_dt), _cp.CartpoleCost()
    _enc = _pa.StateEncoding.IGNORE_UNCERTAINTY
    _g = _t.Generator().manual_seed(0)
    _alphas = 1.025 ** (-_t.arange(10.0) ** 2)
    _umin, _umax = _t.tensor([-_bound]), _t.tensor([_bound])
    _n, _t0 = 0, time.perf_counter()
    while time.perf_counter() - _t0 < _seconds:
        _z0 = 1e-2 * _t.randn(4, generator=_g)
        _U = 0.1 * _t.randn(_N, 1, generator=_g)
        for _ in range(2):
            try:
                _U = _tp.iteration(_z0, _U, _model, _cost, _enc, _umin, _umax,
                                   _alphas, reg=1e-3)[0]
            except RuntimeError:
                pass  # (a failed sweep is an attempt too)
            _n += 1
    print("%d %.6f" % (_n, time.perf_counter() - _t0))
    sys.exit(0)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_trajectory(N, n, m, itemsize, bounded):
    """BASELINE.md 3 / SURVEY.md 8(d): bytes the backward sweep must move."""
    per_step = 2 * n * n + 3 * n * m + n + 2 * m + m * m + (m if bounded else 0)
    return itemsize * (N * per_step + n + n * n)


def available_cores():
    """Cores this process may actually use: the scheduler affinity capped by
    the cgroup CPU quota (a 1-GPU box shows 256 cores but is granted 16:
    cpu.max = 1600000 100000).  More worker processes than that only
    time-slice."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()
        if q != "max":
            quota = float(q) / float(period)
            cores = max(1, min(cores, int(quota + 0.5)))
    except (OSError, ValueError):
        pass
    return cores, quota


def launch_ranks_if_asked(args):
    """`--gpus N` without a torchrun environment: start the N ranks as a child
    `python -m torch.distributed.run` (this process has not touched the GPU
    and never will), relay the child's output and exit with its code."""
    import socket
    import subprocess
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n"
                             % (args.gpus, env_world))
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    have = torch.cuda.device_count()  # (does not initialise the GPU)
    if have < args.gpus and not os.environ.get("PDDP_BENCH_ONE_DEVICE"):
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n"
                         % (args.gpus, have))
        sys.exit(2)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def init_ranks():
    """(world, rank, device) from the torchrun environment; RCCL process group
    when world > 1.  Rehearsal on a box with one GPU (tests only):
    PDDP_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and PDDP_BENCH_BACKEND=gloo
    carries the exchange (RCCL refuses two ranks on one device)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PDDP_BENCH_ONE_DEVICE"):
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("PDDP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(
                "cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return world, rank, torch.device("cuda", local_rank)


def ranks_seen(world, device):
    """World size as the collective library reports it after a real exchange
    (an all_gather of every rank's id over RCCL)."""
    import torch.distributed as dist
    if world == 1:
        return 1
    mine = torch.tensor([dist.get_rank()], dtype=torch.int64, device=device)
    out = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return int(out.unique().numel())


def cpu_baseline(problem_name, dt, N, bound, seconds=12.0):
    """Times the oracle (plain C port of the reference's algorithm) on the
    host's cores on a bounded sample of the same workload; unit = the same
    trajectory-iterations/s.  One worker process per core (trajectories are
    independent, as on the GPU) this process is granted (available_cores();
    the workers are fresh interpreters that never touch the GPU."""
    import subprocess
    workers, quota = available_cores()
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker",
           problem_name, repr(dt), str(N), repr(bound), repr(seconds)]
    procs = [subprocess.Popen(cmd + [str(12345 + w)], stdout=subprocess.PIPE,
                              text=True) for w in range(workers)]
    attempts, trajs, el = 0, 0, 0.0
    for p in procs:
        out = p.communicate()[0].strip().splitlines()[-1].split()
        attempts += int(out[0])
        trajs += int(out[1])
        el = max(el, float(out[2]))
    # second leg: the torch restatement, one thread, a few iterations
    torch_leg = None
    try:
        tout = subprocess.run(
            [sys.executable, os.path.abspath(__file__),
             "--cpu-baseline-torch-worker", repr(dt), str(N), repr(bound),
             repr(min(seconds, 8.0))], stdout=subprocess.PIPE, text=True,
            timeout=120).stdout.strip().splitlines()[-1].split()
        torch_leg = {
            "value": int(tout[0]) / float(tout[1]),
            "unit": "trajectory-iterations/s", "cores": 1, "kind": "port",
            "sample": "%s iterations of one cartpole trajectory in %s s, "
                      "oracle/torch_port.py (op-for-op PyTorch-CPU restatement "
                      "of the reference's iteration, autograd derivatives, "
                      "torch.set_num_threads(1), fp32)" % (tout[0], tout[1])}
    except Exception as e:  # (reported, not hidden)
        torch_leg = {"error": repr(e)[:200]}
    return {"value": attempts / el, "unit": "trajectory-iterations/s",
            "cores": workers, "kind": "port", "torch_restatement": torch_leg,
            "host_cores": os.cpu_count(), "cgroup_cpu_quota": quota,
            "per_core_value": attempts / el / workers,
            "sample": "%d cartpole trajectories x up to 8 iLQR iterations "
                      "(%d attempts) in %.1f s on %d worker processes (= the "
                      "cores granted: affinity capped by the cgroup quota), "
                      "oracle C port, fp32, host has %d cores"
                      % (trajs, attempts, el, workers, os.cpu_count())}


MFMA_F64_PEAK_TFLOPS = 78.6  # dense f64 matrix rate (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.0  # MI355X_MICROARCH.md: dense f32 matrix peak
_last_gp_solver = None



def bnn_cpu_baseline(model, enc, N, n, m, A, opts):
    """The torch model of the BNN workloads on the host, ONE thread, same
    weights: the network work of one trajectory-iteration - the derivative
    rollout's 1 + n + m rows (the input and its n + m directions,
    evaluation.py:203-226 replicates the input like this) and the line
    search's A candidate rows, N moment-matched steps each - timed once.  The
    sweep, the cost derivatives and autograd's backward pass are NOT included
    (the figure flatters the CPU); kind "port": pddp_amd's own torch
    restatement of modules.py:287-386 (the reference does not travel)."""
    import copy
    import pddp_amd
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        cpu = copy.deepcopy(model).cpu().eval()
        cpu.resample()
        D = cpu.state_size
        g = torch.Generator().manual_seed(0)
        z0 = pddp_amd.GaussianVariable(
            torch.zeros(D), var=1e-2 * torch.ones(D)).encode(enc)
        t0 = time.perf_counter()
        with torch.no_grad():
            for rows in (1 + n + m, A):
                z = z0.unsqueeze(0).repeat(rows, 1)
                for t in range(N):
                    u = 0.1 * torch.randn(rows, m, generator=g)
                    z = cpu(z, u, t, enc, **opts)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(threads)
    return {"value": 1.0 / dt, "unit": "trajectory-iterations/s", "cores": 1,
            "kind": "port",
            "sample": "network part of ONE trajectory-iteration (N = %d steps "
                      "of %d derivative rows + %d candidate rows, 100 "
                      "particles) in %.2f s on one host thread; sweep, cost "
                      "derivatives and the backward pass of autograd excluded"
                      % (N, 1 + n + m, A, dt)}


def sweep_roofline_of(s, lib, reps=6, traffic_tag=None):
    """HBM roofline entry of the backward sweep of a live solver: events on
    the dispatch, algorithmic bytes of SURVEY 8(d)."""
    pool = EventPool(lib)
    # every trajectory swept (active=None): the bytes below are those of all B;
    # the gains of converged trajectories are rewritten with the same values
    for _ in range(2):
        s.backward(active=None, variant=s.kernel_variant)
    for _ in range(reps):
        s.backward(active=None, variant=s.kernel_variant,
                   events=pool.pair())
    torch.cuda.synchronize(s.device)
    d = np.array(pool.durations())
    nbytes = s.B * algorithmic_bytes_per_trajectory(
        s.N, s.n, s.m, s.rec.element_size(), True)
    ach = nbytes / float(d.mean()) / 1e9
    # the same kernel INSIDE the workload's rounds, from the committed
    # rocprofv3 summary of this workload (profiles/<round>_<tag>_kernel_stats
    # .csv): behind the derivative rollout's launches the sweep runs at 0.6-0.8
    # of its back-to-back speed for a few launches - the
    # figure above is the back-to-back one
    in_round = None
    if traffic_tag is not None:
        import csv
        import glob
        for path in sorted(glob.glob(os.path.join(
                ROOT, "profiles", "r0*_%s_kernel_stats.csv" % traffic_tag)),
                reverse=True):
            try:
                for row in csv.DictReader(open(path)):
                    if "riccati" in row["Name"]:
                        avg = float(row["AverageNs"]) * 1e-3
                        in_round = {
                            "avg_launch_us": avg,
                            "min_launch_us": float(row["MinNs"]) * 1e-3,
                            "max_launch_us": float(row["MaxNs"]) * 1e-3,
                            "calls": int(row["Calls"]),
                            "frac": nbytes / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            "source": "profiles/" + os.path.basename(path)}
                        break
            except (OSError, KeyError, ValueError):
                in_round = None
            if in_round:
                break
    return {"bound": "hbm", "kernel": "backward Riccati sweep (n = %d)" % s.n,
            "measured": "launched back to back, events on the dispatch",
            "in_the_workloads_rounds": in_round,
            "avg_launch_us": float(d.mean()) * 1e6,
            "min_launch_us": float(d.min()) * 1e6,
            "algorithmic_bytes_per_launch": nbytes, "achieved": ach,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            # counters exist only for the profiled shapes (traffic_tag names
            # the committed summary of THIS workload); None elsewhere
            "traffic": None if traffic_tag is None else profile_traffic(
                {4: "riccati_n4", 14: "riccati_mfma16",
                 27: "riccati_mfma32"}.get(s.n, "riccati_generic"),
                traffic_tag)}


def profile_traffic(kernel_substr, workload_tag=None):
    """HBM bytes per launch of a kernel from the committed PMC summaries
    (profiles/<round>[_<workload>]_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE
    / WRITE_SIZE in separate passes, 2*FETCH+WRITE), newest round first; None
    when no summary has a kernel whose name contains `kernel_substr`.  Counters
    cannot be read from inside this process: the summaries are of profiled
    runs of the same bench command (tools/collect_profiles.sh)."""
    import glob
    for rnd in ("r05", "r04", "r03", "r02", "r01"):
        pat = os.path.join(ROOT, "profiles", "%s_%spmc_traffic.json" % (
            rnd, (workload_tag + "_") if workload_tag else "*"))
        for path in sorted(glob.glob(pat)):
            try:
                with open(path) as fh:
                    kernels = json.load(fh)["kernels"]
            except (OSError, KeyError, ValueError):
                continue
            for name, v in kernels.items():
                if kernel_substr in name:
                    return {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                            "kernel": name[:100],
                            "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE"
                                      " / WRITE_SIZE, 2*FETCH+WRITE)"
                                      % os.path.basename(path)}
    return None


def bench_bnn(args, emit=True):
    """Secondary workloads.  --workload cartpole_bnn = BASELINE.json configs[2]:
    cartpole with the BNN dynamics model ([200, 200] hidden, 100 particles,
    moment-matched rollouts, DEFAULT encoding n = 14), horizon 100, B = 4096
    trajectories per GPU.  --workload double_cartpole_bnn = configs[3]'s problem
    with the reference's own double-cartpole model (a BNN,
    examples/double_cartpole.py:133-139; the reference has no GP): n = 27,
    horizon 150, B = 1024 per GPU (8192 over 8), ranks own disjoint shards and
    exchange their best rollout once through RCCL.  A step is one round of the
    fit loop: forward-mode derivative rollout of the trajectories whose nominal
    changed, backward sweep, 10-candidate line search, accept.  The dominant
    kernel is the fused network (f32 matrix cores): roofline bound "mfma"."""
    import pddp_amd
    import torch.distributed as dist
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.models.bnn import bnn_dynamics_model_factory
    from pddp_amd.parallel import (gather_best_rollout, post_best_rollout,
                                   shard_bounds)
    world, rank, dev = init_ranks()
    seen = ranks_seen(world, dev)
    torch.manual_seed(0)  # the same network on every rank
    # --dtype f64: the same rounds on the float64 kernels (csrc/bnn_mlp_f64.hip
    # on v_mfma_f64_16x16x4; the reference runs in the dtype of its inputs)
    f64 = args.dtype == "f64"
    dt = torch.float64 if f64 else torch.float32
    peak = MFMA_F64_PEAK_TFLOPS if f64 else MFMA_F32_PEAK_TFLOPS
    if args.workload == "cartpole_bnn":
        from pddp_amd.examples import cartpole as ex
        CM, cost_cls = ex.CartpoleDynamicsModel, ex.CartpoleCost
        mean0, bound, tag = [0.0, 0.0, 3.14159, 0.0], 10.0, "configs[2]: cartpole"
        B = args.batch or 4096
        N = args.horizon or 100
    else:
        from pddp_amd.examples import double_cartpole as ex
        CM, cost_cls = ex.DoubleCartpoleDynamicsModel, ex.DoubleCartpoleCost
        mean0, bound = [0.0, 0.0, 3.14159, 0.0, 3.14159, 0.0], 20.0
        tag = "configs[3]'s problem: double cartpole"
        B = args.batch or 1024
        N = args.horizon or 150
    lo = rank * B
    if args.scaling == "strong":  # --batch in total, sharded
        lo, hi = shard_bounds(B, rank, world)
        B = hi - lo
    D, m, P, A, H = CM.state_size, 1, 100, 10, 200
    n = D + D * (D + 1) // 2
    in_dim = len(CM.non_angular_indices) + 2 * len(CM.angular_indices) + m
    K = args.steps if args.steps != 30 else 3
    W = args.warmup if args.warmup != 5 else 1
    model = bnn_dynamics_model_factory(D, m, [H, H], CM.angular_indices,
                                       CM.non_angular_indices)(
        n_particles=P).to(dev).to(dt).eval()
    with torch.no_grad():  # untrained weights: keep the dynamics gentle
        model.model.out.weight.mul_(0.05)
        model.model.out.bias.mul_(0.05)
    cost = cost_cls().to(dev).to(dt)
    enc = pddp_amd.StateEncoding.DEFAULT
    plugin = TorchProblem(model, cost, enc, {"use_predicted_std": False,
                                             "infer_noise_variables": True}, {})
    s = ILQRSolver(None, B, N, dt, dev, torch.tensor([-bound], dtype=dt),
                   torch.tensor([bound], dtype=dt), fit_alphas(dt, dev),
                   plugin=plugin, n=n, m=m)
    g = torch.Generator().manual_seed(rank)
    mean = torch.tensor(mean0)
    z0 = torch.stack([pddp_amd.GaussianVariable(
        mean + 1e-2 * torch.randn(D, generator=g),
        var=1e-2 * torch.ones(D)).encode(enc) for _ in range(B)]).to(dev).to(dt)
    U0 = (0.1 * torch.randn(B, N, m, generator=g)).to(dev).to(dt)
    s.set_nominal(z0, U0)
    n_iter = 1 << 30
    for _ in range(W):
        s.round(5e-6, 1e10, n_iter)
    s.n_live.zero_()
    live0 = int(s.active.sum().item())
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(K):
        s.round(5e-6, 1e10, n_iter)
        if world > 1 and args.exchange_every > 0 and \
                (i + 1) % args.exchange_every == 0:
            last_exchange = post_best_rollout(s.J_opt, s.Z, s.U, offset=lo)
    if world > 1 and args.exchange_every <= 0:
        last_exchange = post_best_rollout(s.J_opt, s.Z, s.U, offset=lo)
    if world > 1:
        last_exchange.result()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    liveK = int(s.active.sum().item())
    attempted = live0 + int(s.n_live.sum().item()) - liveK
    total_attempted = attempted
    if world > 1:
        t = torch.tensor([elapsed, attempted], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_attempted = float(tmax[0].item()), int(t[1].item())
    # the dominant kernel, timed alone on torch's current stream (the stream it
    # is launched on): one forward-mode network pass of a time step
    grp = 8  # rows per (state, particle) in memory
    live = 1 + D + m  # of which in use: the input and the D + m directions
    F = torch.randn(B * P * grp, in_dim, device=dev, dtype=dt)
    model.model._jvp_native(F, P, D, grp, live=live)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        model.model._jvp_native(F, P, D, grp, live=live)
    e1.record()
    torch.cuda.synchronize(dev)
    dur = e0.elapsed_time(e1) * 1e-3 / reps
    # algorithmic: the rows that exist (padding rows of the 32-row tiles - 2 of
    # 32 with 6 live rows per group - are the kernel's overhead, not work)
    flop = 2.0 * B * P * live * (in_dim * H + H * H + H * D)
    out = {
        "metric": "pddp_iterations_per_sec", "value": total_attempted / elapsed,
        "unit": "trajectory-iterations/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic", "rccl_ranks_seen": seen,
        "config": {
            "workload": "BASELINE.json %s with BNN dynamics ([200,200] hidden, "
                        "%d particles, moment-matched rollouts, DEFAULT "
                        "encoding n=%d m=1), horizon=%d, batch=%d trajectories "
                        "per GPU, bounds +-%g, 10 line-search alphas, "
                        "random-init network weights" % (tag, P, n, N, B, bound),
            "batch_per_gpu": B, "horizon": N, "alphas": A,
            "unit_definition": "one iLQR attempt of one trajectory (derivative "
                               "rollout when its nominal changed + backward "
                               "sweep + line search + accept)",
            "live_trajectories_start_end": [live0, liveK],
            "derivative_path": getattr(plugin, "last_derivs_path", None),
        },
        "roofline": {
            "bound": "mfma",
            "kernel": ("fused BNN network, forward-mode (bnn_mlp_f64_kernel<%d, "
                       "%d>: %d live rows per (state, particle); "
                       "v_mfma_f64_16x16x4_f64)" % (H, grp, live)) if f64 else
                      "fused BNN network, forward-mode (bnn_mlp_kernel<%d, %d, %d, "
                      "%d>: %d live rows per (state, particle))"
                      % (H, 8 if in_dim < 8 else 16, grp,
                         4 if live <= 4 else (6 if live <= 6 else 8), live),
            "achieved": flop / dur * 1e-12, "peak": peak,
            "unit": "TFLOP/s", "frac": flop / dur * 1e-12 / peak,
            "avg_launch_us": dur * 1e6,
            "algorithmic_flop_per_launch": flop,
            # (the network kernel is matrix-bound; its HBM traffic is reported
            # for completeness where a profiled run of this workload exists)
            "traffic": profile_traffic("bnn_mlp_f64_kernel<%d, %d>" % (H, grp),
                                       "cpbnn_f64") if f64 else profile_traffic(
                "bnn_mlp_kernel<%d, %d, %d, %d" % (
                    H, 8 if in_dim < 8 else 16, grp,
                    4 if live <= 4 else (6 if live <= 6 else 8)),
                "dcbnn" if args.workload == "double_cartpole_bnn" else "cpbnn"),
        },
        "cpu_baseline": None,
    }
    if world == 1 and f64:
        if rank == 0 and emit:
            print(json.dumps(out))
        return out
    if world == 1:
        from pddp_amd import _native
        out["roofline"]["other_kernels"] = [
            sweep_roofline_of(s, _native.lib(), traffic_tag=(
                "dcbnn" if args.workload == "double_cartpole_bnn"
                else "cpbnn"))]
        # the same rounds with layer 2 of the network kernel on its bf16-split
        # twin (opt-in: pddp_bnn_mlp_precision(3) - f32 to rounding, not
        # bit-exact; the line above is the exact-f32 kernel)
        lib = _native.lib()
        prev = lib.pddp_bnn_mlp_precision(3)
        try:
            s.set_nominal(z0, U0)
            for _ in range(W):
                s.round(5e-6, 1e10, n_iter)
            s.n_live.zero_()
            l0 = int(s.active.sum().item())
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for i in range(K):
                s.round(5e-6, 1e10, n_iter)
            torch.cuda.synchronize(dev)
            el3 = time.perf_counter() - t1
            att3 = l0 + int(s.n_live.sum().item()) - int(s.active.sum().item())
            model.model._jvp_native(F, P, D, grp, live=live)
            e0.record()
            for _ in range(reps):
                model.model._jvp_native(F, P, D, grp, live=live)
            e1.record()
            torch.cuda.synchronize(dev)
            dur3 = e0.elapsed_time(e1) * 1e-3 / reps
        finally:
            lib.pddp_bnn_mlp_precision(prev)
        out["bf16_split_twin"] = {
            "what": "layer 2 of the network kernel as three bf16 parts per "
                    "operand, six v_mfma_f32_32x32x16_bf16 per product "
                    "(pddp_bnn_mlp_precision(3)); f32 to rounding, opt-in",
            "ms_per_step": el3 / K * 1e3, "value": att3 / el3,
            "network_kernel_avg_launch_us": dur3 * 1e6,
            "network_kernel_tflops_equivalent": flop / dur3 * 1e-12,
            "frac_of_exact_f32_matrix_peak": flop / dur3 * 1e-12 /
                                             MFMA_F32_PEAK_TFLOPS,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = bnn_cpu_baseline(
                model, enc, N, n, m, A, {"use_predicted_std": False,
                                         "infer_noise_variables": True})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and emit:
        print(json.dumps(out))
    return out


def bench_gp(args, emit=True):
    """--workload double_cartpole_gp: BASELINE.json configs[3] AS STATED -
    double cartpole (state 6) with GP dynamics, horizon 150, 1024 trajectories
    per GPU - on the build's own GP plugin (pddp_amd/models/gp.py:
    squared-exponential GPs, exact moment matching; the reference has no GP:
    PARITY UNPINNED, checkers: the torch module + autograd and
    oracle/gp_port.py).  The moment-matched step and its Jacobian are ONE HIP
    kernel (csrc/gp_step.hip, pddp_gp_step_f32): the derivative rollout is a
    single launch over all B N rows, the line search N launches of B A rows.
    A step is one round of the fit loop."""
    import pddp_amd
    from pddp_amd.controllers.ilqr import fit_alphas
    from pddp_amd.controllers.plugin import TorchProblem
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.examples import double_cartpole as ex
    from pddp_amd.models.gp import gp_dynamics_model_factory
    import torch.distributed as dist
    from pddp_amd.parallel import post_best_rollout
    world, rank, dev = init_ranks()
    seen = ranks_seen(world, dev)
    CM, cost_cls = ex.DoubleCartpoleDynamicsModel, ex.DoubleCartpoleCost
    B = args.batch or 1024
    N = args.horizon or 150
    K = args.steps if args.steps != 30 else 2
    W = args.warmup if args.warmup != 5 else 1
    # training points: 60 (rounds 3-4), or what the outer loop of
    # PDDPController.fit hands a model after its first two trials -
    # n_initial_sample_trajectories = 2 x N = 150 rows (pddp.py:67-71,121-150)
    M = getattr(args, "gp_points", None) or 60
    D, m, A = 6, 1, 10
    n = D + D * (D + 1) // 2
    g = torch.Generator().manual_seed(0)
    true = CM(0.05)
    mean0 = torch.tensor([0.0, 0.0, 3.14159, 0.0, 3.14159, 0.0])
    X = mean0 + torch.cat([0.5 * torch.randn(M, 2, generator=g),
                           0.8 * torch.randn(M, 1, generator=g),
                           torch.randn(M, 1, generator=g),
                           0.8 * torch.randn(M, 1, generator=g),
                           torch.randn(M, 1, generator=g)], -1)
    U = 6.0 * torch.randn(M, 1, generator=g)
    with torch.no_grad():
        dX = true(X, U, 0, pddp_amd.StateEncoding.IGNORE_UNCERTAINTY) - X
    model = gp_dynamics_model_factory(D, m, CM.angular_indices,
                                      CM.non_angular_indices)().to(dev)
    model.fit(X.to(dev), U.to(dev), dX.to(dev))
    model.eval()
    cost = cost_cls().to(dev)
    enc = pddp_amd.StateEncoding.DEFAULT
    plugin = TorchProblem(model, cost, enc, {}, {})
    s = ILQRSolver(None, B, N, torch.float32, dev, torch.tensor([-20.0]),
                   torch.tensor([20.0]), fit_alphas(torch.float32, dev),
                   plugin=plugin, n=n, m=m)
    g = torch.Generator().manual_seed(rank)
    z0 = torch.stack([pddp_amd.GaussianVariable(
        mean0 + 1e-2 * torch.randn(D, generator=g),
        var=1e-2 * torch.ones(D)).encode(enc) for _ in range(B)]).to(dev)
    s.set_nominal(z0, (0.1 * torch.randn(B, N, m, generator=g)).to(dev))
    for _ in range(W):
        s.round(5e-6, 1e10, 1 << 30)
    s.n_live.zero_()
    live0 = int(s.active.sum().item())
    # shards of configs[3]'s 8192 trajectories: rank r owns B of them, no
    # data-path collective; the one exchange is the best rollout, per round
    lo = rank * B
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    last_exchange = None
    for _ in range(K):
        s.round(5e-6, 1e10, 1 << 30)
        if world > 1:
            last_exchange = post_best_rollout(s.J_opt, s.Z, s.U, offset=lo)
    if last_exchange is not None:
        last_exchange.result()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    liveK = int(s.active.sum().item())
    attempted = live0 + int(s.n_live.sum().item()) - liveK
    total_attempted = attempted
    if world > 1:
        t = torch.tensor([elapsed, attempted], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_attempted = float(tmax[0].item()), int(t[1].item())
    global _last_gp_solver
    _last_gp_solver = s  # (tests look at the state the rounds left)
    # the same rounds replayed as hipGraphs (ILQRSolver.fit(graph=True)):
    # informational, `value` is the eager loop
    graph_ms = None
    if world == 1 and s.graph_ok() and not getattr(args, "no_graph_replay",
                                                    False):
        try:
            s.capture_round(5e-6, 1e10, 1 << 30)
            s.replay_round(True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(K):
                s.replay_round(True)
            torch.cuda.synchronize(dev)
            graph_ms = (time.perf_counter() - t1) / K * 1e3
        except Exception as e:  # noqa: BLE001 (reported in the line)
            graph_ms = repr(e)[:120]
    # the dominant kernel: one moment-matched step of the line search's
    # candidate rows (B A rows per launch, N launches per round), timed with
    # events on the stream it is launched on (torch's current stream)
    rows = B * A
    zr = z0.repeat_interleave(A, 0)
    ur = torch.zeros(rows, m, device=dev)
    native = bool(model.native_ok(zr, enc))
    with torch.no_grad():
        model(zr, ur, 0, enc)
        reps = 5
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(reps)]
        for i in range(reps):
            e0[i].record()
            model(zr, ur, 0, enc)
            e1[i].record()
        torch.cuda.synchronize(dev)
        dur = sum(a_.elapsed_time(b_) for a_, b_ in zip(e0, e1)) / reps * 1e-3
    d_in = 9
    NPAIR = D * (D + 1) // 2
    # algorithmic flops of a row: the M^2 exponents of every output pair -
    # a d-term product (2 d), two additions, the exponential counted as one,
    # weight and accumulation (3): 2 d + 6 per (pair, i, j)
    flop = rows * NPAIR * M * M * (2 * d_in + 6)
    # the derivative rollout's launch (B N rows with Jacobians), once
    jac_ms = None
    if native:
        zj = s.Z[:, :N].reshape(B * N, n).contiguous()
        uj = s.U.reshape(B * N, m).contiguous()
        Fz = torch.empty(B * N, n, n, device=dev)
        Fu = torch.empty(B * N, n, m, device=dev)
        model.native_step(zj, uj, enc, jacobian=True, Fz=Fz, Fu=Fu)
        ej = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ej[0].record()
        model.native_step(zj, uj, enc, jacobian=True, Fz=Fz, Fu=Fu)
        ej[1].record()
        torch.cuda.synchronize(dev)
        jac_ms = ej[0].elapsed_time(ej[1])
        del Fz, Fu
    out = {
        "metric": "pddp_iterations_per_sec", "value": total_attempted / elapsed,
        "unit": "trajectory-iterations/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "rccl_ranks_seen": seen,
        "config": {
            "workload": "BASELINE.json configs[3] as stated: double cartpole "
                        "(state 6, DEFAULT encoding n=27 m=1) with GP dynamics "
                        "(the build's own plugin - the reference has no GP: "
                        "PARITY UNPINNED, checkers: the torch module + "
                        "autograd, oracle/gp_port.py), %d training points, "
                        "exact moment matching and its Jacobian in one HIP "
                        "kernel (pddp_gp_step_f32), horizon=%d, batch=%d "
                        "trajectories per GPU, bounds +-20, 10 line-search "
                        "alphas; cost and control law of the line search "
                        "still torch ops per time step" % (M, N, B),
            "gp_step_on": "hip" if native else "torch",
            "derivative_rollout_launch_ms": jac_ms,
            "hipgraph_replay_ms_per_step": graph_ms,
            "batch_per_gpu": B, "horizon": N, "alphas": A,
            "training_points": M, "parity": "unpinned",
            "live_trajectories_start_end": [live0, liveK],
            "derivative_path": getattr(plugin, "last_derivs_path", None)},
        "roofline": {
            # (vector arithmetic, no matrix cores: priced against the f32
            # rate of the part, which is the same number for both)
            "bound": "mfma", "unit_of_work": "valu",
            "kernel": "gp_roll_f32_kernel<6, 9> (gp_step_body: control law, "
                      "GP moment-matched step and stage cost of %d candidate "
                      "rows - one line-search time step)"
                      % rows,
            "achieved": flop / dur * 1e-12, "peak": MFMA_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": flop / dur * 1e-12 / MFMA_F32_PEAK_TFLOPS,
            "avg_launch_us": dur * 1e6, "algorithmic_flop_per_launch": flop,
            "flop_per_pair_point_point": 2 * d_in + 6,
            "traffic": profile_traffic("gp_roll_f32_kernel<6, 9>", "dcgp") or
                       profile_traffic("gp_step_fwd_f32_kernel<6, 9>", "dcgp")},
        "cpu_baseline": None,
    }
    if world == 1:
        from pddp_amd import _native
        out["roofline"]["other_kernels"] = [sweep_roofline_of(
            s, _native.lib(), traffic_tag="dcgp")]
        if not args.no_cpu_baseline:
            import copy
            cpu = copy.deepcopy(model).cpu()
            threads = torch.get_num_threads()
            torch.set_num_threads(1)
            try:
                zc, uc = z0[:1].cpu().repeat(1 + n + m, 1), torch.zeros(1 + n + m, m)
                t2 = time.perf_counter()
                with torch.no_grad():
                    for t in range(8):
                        zc = cpu(zc, uc, t, enc)
                per_step = (time.perf_counter() - t2) / 8
            finally:
                torch.set_num_threads(threads)
            # one trajectory-iteration: N steps of (1 + n + m) derivative rows
            # and of A candidate rows
            sec = per_step * N * (1 + n + m + A) / (1 + n + m)
            out["cpu_baseline"] = {
                "value": 1.0 / sec, "unit": "trajectory-iterations/s",
                "cores": 1, "kind": "port",
                "sample": "8 moment-matched steps of %d rows of the same torch "
                          "GP on one host thread (%.3f s per step), scaled to "
                          "N = %d steps of derivative + candidate rows; sweep "
                          "and autograd's backward pass excluded"
                          % (1 + n + m, per_step, N)}
    if rank == 0 and emit:
        print(json.dumps(out))
    return out


def bench_mpc_bnn(args, emit=True):
    """--workload mpc_bnn = BASELINE.json configs[4]: the receding-horizon loop
    of examples/mpc_animation.py:29-39 on cartpole with the BNN dynamics model,
    horizon 50, 256 restarts x 200 control steps, 11-alpha schedule
    (ilqr.py:116), one GPU.  A step = one control step of all restarts:
    `iLQRController.forward(mpc=True)` (ilqr.py:318-362: reset the
    regularisation, one fit iteration from the measured state, emit U[0], shift
    the plan); the plant is the true cartpole model (the reference steps a gym
    env; no env on the GPU box).  Reported eager and, where the controller
    supports it, with every round replayed as a captured hipGraph."""
    import pddp_amd
    from pddp_amd.examples import cartpole
    from pddp_amd.models.bnn import bnn_dynamics_model_factory
    world, rank, dev = init_ranks()
    B = args.batch or 256
    N = args.horizon or 50
    K = args.steps if args.steps != 30 else 200
    W = args.warmup if args.warmup != 5 else 3
    P = 100
    CM = cartpole.CartpoleDynamicsModel
    enc = pddp_amd.StateEncoding.DEFAULT
    ienc = pddp_amd.StateEncoding.IGNORE_UNCERTAINTY
    iu = torch.triu_indices(4, 4)
    tri = (0.1 * torch.eye(4))[iu[0], iu[1]].to(dev)  # var 1e-2: chol = 0.1 I

    keep = {}

    def run(graph):
        torch.manual_seed(0)
        model = bnn_dynamics_model_factory(
            4, 1, [200, 200], CM.angular_indices, CM.non_angular_indices)(
                n_particles=P).to(dev).eval()
        with torch.no_grad():  # untrained network: keep its dynamics gentle
            model.model.out.weight.mul_(0.05)
            model.model.out.bias.mul_(0.05)
        cost = cartpole.CartpoleCost().to(dev)
        plant = CM(0.1).to(dev)
        ctrl = pddp_amd.controllers.iLQRController(
            None, model, cost, graph=graph,
            model_opts={"use_predicted_std": False,
                        "infer_noise_variables": True})
        u_min, u_max = torch.tensor([-10.0]), torch.tensor([10.0])
        g = torch.Generator().manual_seed(rank)
        ctrl._U_nominal = (0.1 * torch.randn(B, N, 1, generator=g)).to(dev)
        x = (torch.tensor([0.0, 0.0, 3.14159, 0.0]) +
             1e-2 * torch.randn(B, 4, generator=g)).to(dev)
        x_start = x.clone()
        rounds = [0]
        resets = torch.zeros((), dtype=torch.int64, device=dev)

        def control_step(x):
            z = torch.cat([x, tri.expand(B, -1)], -1)
            u = ctrl(z, 0, enc, mpc=True, u_min=u_min, u_max=u_max)
            rounds[0] += ctrl._last_rounds
            with torch.no_grad():
                xn = plant(x, u.clamp(-10.0, 10.0), 0, ienc)
                # an episode ends where a gym environment would end it: the
                # random-weight network sometimes holds the action at its
                # bound until the explicit-Euler plant runs away
                gone = ~(xn.abs().amax(-1) < 1e3)
                resets.add_(gone.sum())
                return torch.where(gone[:, None], x_start, xn)

        for _ in range(W):
            x = control_step(x)
        torch.cuda.synchronize(dev)
        rounds[0] = 0
        t0 = time.perf_counter()
        for _ in range(K):
            x = control_step(x)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        keep["model"], keep["solver"] = model, ctrl._solver
        return dt, rounds[0] / K, getattr(ctrl._solver.plugin,
                                          "last_derivs_path", None), \
            bool(torch.isfinite(x).all()), int(resets)

    dt, rps, path, finite, nres = run(False)
    res = {"eager": {"ms_per_control_step": dt / K * 1e3,
                     "rounds_per_control_step": rps,
                     "episodes_restarted": nres}}
    try:
        dtg, rpsg, _, fg, nresg = run(True)
        res["graph"] = {"ms_per_control_step": dtg / K * 1e3,
                        "rounds_per_control_step": rpsg,
                        "episodes_restarted": nresg}
        if dtg < dt:
            dt = dtg
    except Exception as e:  # (reported, not hidden)
        res["graph"] = {"error": repr(e)[:200]}
    out = {
        "metric": "mpc_restart_control_steps_per_sec", "value": B * K / dt,
        "unit": "restart-control-steps/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[4]: MPC receding-horizon loop, "
                        "cartpole BNN [200,200] P=%d DEFAULT encoding n=14, "
                        "horizon %d, %d restarts x %d control steps, 11 "
                        "alphas, bounds +-10, random-init network weights"
                        % (P, N, B, K),
            "restarts": B, "horizon": N, "control_steps": K,
            "modes": res, "derivative_path": path, "state_finite": finite},
        "roofline": None, "cpu_baseline": None,
    }
    if world == 1 and "model" in keep:
        # the dominant kernel of a control step: the fused network in forward
        # mode (derivative rollout of every restart), timed alone
        from pddp_amd import _native
        model, sv = keep["model"], keep["solver"]
        D, m, H, grp = 4, 1, 200, 8
        live = 1 + D + m
        in_dim = len(CM.non_angular_indices) + 2 * len(CM.angular_indices) + m
        F = torch.randn(B * P * grp, in_dim, device=dev)
        model.model._jvp_native(F, P, D, grp, live=live)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            model.model._jvp_native(F, P, D, grp, live=live)
        e1.record()
        torch.cuda.synchronize(dev)
        dur = e0.elapsed_time(e1) * 1e-3 / 5
        flop = 2.0 * B * P * live * (in_dim * H + H * H + H * D)
        out["roofline"] = {
            "bound": "mfma",
            "kernel": "fused BNN network, forward mode, %d restarts "
                      "(launch-bound at this batch: a control step is %.1f "
                      "rounds of ~%d launches)" % (B, rps, 2 * (N + 1)),
            "achieved": flop / dur * 1e-12, "peak": MFMA_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": flop / dur * 1e-12 / MFMA_F32_PEAK_TFLOPS,
            "avg_launch_us": dur * 1e6, "algorithmic_flop_per_launch": flop,
            # (this workload's own counters, profiles/<round>_mpc_pmc_traffic
            # .json, or null: never another batch's launch)
            "traffic": profile_traffic("bnn_mlp_kernel<200, 8, 8", "mpc"),
            "other_kernels": [sweep_roofline_of(sv, _native.lib(),
                                                traffic_tag="mpc")]}
        if not args.no_cpu_baseline:
            cb = bnn_cpu_baseline(model, enc, N, 14, 1, 11,
                                  {"use_predicted_std": False,
                                   "infer_noise_variables": True})
            # a control step is `rps` trajectory-iterations of every restart
            cb["value"] = cb["value"] / max(rps, 1e-9)
            cb["unit"] = "restart-control-steps/s"
            cb["sample"] += "; x %.2f rounds per control step" % rps
            out["cpu_baseline"] = cb
    if rank == 0 and emit:
        print(json.dumps(out))
    return out


class EventPool(object):
    """HIP event pairs attached to single dispatches (pddp_attach_events /
    pddp_riccati_backward_timed): elapsed(start, stop) is the kernel's own
    duration on the stream it was launched on."""

    def __init__(self, lib):
        self.lib, self.pairs = lib, []

    def pair(self):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        self.lib.pddp_event_create(ctypes.byref(a))
        self.lib.pddp_event_create(ctypes.byref(b))
        self.pairs.append((a, b))
        return a, b

    def durations(self):
        """Seconds per pair, in creation order; destroys the events."""
        out = []
        for a, b in self.pairs:
            ms = ctypes.c_float()
            self.lib.pddp_event_elapsed_ms(a, b, ctypes.byref(ms))
            out.append(ms.value * 1e-3)
            self.lib.pddp_event_destroy(a)
            self.lib.pddp_event_destroy(b)
        self.pairs = []
        return out


def make_cartpole_solver(B, N, dtype, device, seed, variant=0):
    import pddp_amd
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.examples import cartpole
    enc = pddp_amd.StateEncoding.IGNORE_UNCERTAINTY
    model, cost = cartpole.CartpoleDynamicsModel(0.1), cartpole.CartpoleCost()
    prob = model.native_problem(enc, cost)
    n, m = prob.encoded_size, prob.action_size
    bound = 10.0
    u_min = torch.full((m,), -bound, dtype=dtype)
    u_max = torch.full((m,), bound, dtype=dtype)
    s = ILQRSolver(prob, B, N, dtype, device, u_min, u_max,
                   kernel_variant=variant)
    g = torch.Generator().manual_seed(seed)
    z0 = (1e-2 * torch.randn(B, n, generator=g, dtype=torch.float64)).to(dtype)
    U = (0.1 * torch.randn(B, N, m, generator=g, dtype=torch.float64)).to(dtype)
    s._keep = prob
    return s, z0.to(device), U.to(device), bound


def sweep_point(lib, B, N, dtype, device, variant, rounds=12, cold=False,
                solver=None):
    """One extra roofline point of the backward sweep: `rounds` rounds of the
    fit loop from a fresh nominal with events on the sweep's dispatch (warm: the
    records were just written by the previous launch - the fit loop's own
    condition), or, `cold`, the sweep alone with > 512 MB written through the
    caches before every launch."""
    if solver is None:
        s, z0, U, _ = make_cartpole_solver(B, N, dtype, device, 0, variant)
    else:
        s, z0, U = solver
    s._one_launch = False  # (the sweep as a launch of its own)
    s.set_nominal(z0, U)
    for _ in range(3):
        s.round(5e-6, 1e10, 1 << 30)
    pool = EventPool(lib)
    if cold:
        flush = torch.empty(768 << 20, dtype=torch.uint8, device=device)
        nominal = getattr(s, "_nominal_sweep", False) is True
        for i in range(rounds):
            flush.fill_(i)
            if nominal:
                # the sweep the rounds run: from the nominal (its inputs - Z,
                # U - and everything else evicted)
                s.fresh.fill_(1)
                s.sweep_nominal(events=pool.pair())
            else:
                s.backward(active=s.active, variant=s.kernel_variant,
                           events=pool.pair())
        del flush
    else:
        for _ in range(rounds):
            s.round(5e-6, 1e10, 1 << 30, backward_events=pool.pair())
    torch.cuda.synchronize(device)
    s._one_launch = None
    live = int(s.active.sum().item())
    d = np.array(pool.durations())
    itemsize = 4 if dtype == torch.float32 else 8
    nbytes = live * algorithmic_bytes_per_trajectory(N, s.n, s.m, itemsize, True)
    return {"batch": B, "horizon": N,
            "dtype": "f32" if dtype == torch.float32 else "f64",
            "cache": "cold (768 MB written between launches)" if cold else
                     "as in the fit loop",
            "kernel": "pddp_sweep_nominal" if getattr(
                s, "_nominal_sweep", False) is True else
                "pddp_riccati_backward (records)",
            "avg_launch_us": float(d.mean()) * 1e6,
            "min_launch_us": float(d.min()) * 1e6,
            "algorithmic_bytes_per_launch": nbytes,
            "achieved": nbytes / float(d.mean()) / 1e9,
            "frac": nbytes / float(d.mean()) / 1e9 / HBM_PEAK_GBS}


def search_accept_bytes(B, N, n, m, A, S, itemsize, accepted_share=1.0,
                        records=True):
    """Algorithmic bytes of the fused line search + accept (+ records) launch:
    reads the nominal (Z, U) and the gains; writes the A candidate rollouts and
    their costs; for an accepted trajectory reads the winning rollout back,
    writes it as the new nominal, copies the gains (self._K) and - `records`:
    when the next sweep reads them from HBM - writes the derivative records +
    stage costs of the new nominal."""
    zu = (N + 1) * n + N * m
    gains = N * (m + m * n)
    per = zu + gains + A * zu + A
    per_acc = 2 * zu + 2 * gains + ((N + 1) * (S + 1) if records else 0)
    return itemsize * B * (per + accepted_share * per_acc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=None,
                    help="repetitions of the timed region (default 5; 1 for "
                         "the BNN workloads)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch trajectories per GPU; strong: --batch "
                         "trajectories in total, sharded over the GPUs")
    ap.add_argument("--workload", default="cartpole",
                    choices=["cartpole", "cartpole_bnn", "double_cartpole_bnn",
                             "double_cartpole_gp", "mpc_bnn"],
                    help="cartpole = BASELINE configs[1] (the headline line); "
                         "cartpole_bnn = configs[2]; double_cartpole_bnn = "
                         "configs[3]'s problem with the reference's BNN model; "
                         "mpc_bnn = configs[4]")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--horizon", type=int, default=None)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-points", action="store_true",
                    help="skip the extra roofline points (B = 16384, fp64, "
                         "cold cache)")
    ap.add_argument("--exchange-every", type=int, default=None,
                    help="multi-GPU: all-gather the best rollout every E "
                         "rounds inside the timed region; 0: once, after the "
                         "rounds.  Default: after every LAUNCH - every round "
                         "for the workloads whose round is several launches "
                         "(SURVEY 8(e): one exchange per iteration), every "
                         "--rounds-per-launch rounds for the cartpole's "
                         "one-launch rounds (the exchange reports the best "
                         "rollout, nothing on a rank's path reads it; "
                         "--exchange-every 1 = per iteration, one round per "
                         "launch)")
    ap.add_argument("--no-graph-replay", action="store_true",
                    help="double_cartpole_gp: skip the informational hipGraph "
                         "replay (rocprofv3 --pmc and stream capture do not "
                         "get on)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the short runs of BASELINE configs[2] / [3] / "
                         "[4] appended to the default line")
    ap.add_argument("--kernel-variant", type=int, default=0,
                    help="backward kernel (include/pddp_hip.h): 0 auto")
    ap.add_argument("--gp-points", type=int, default=None,
                    help="double_cartpole_gp: training points of the GP "
                         "(default 60; 300 = two trials of 150 steps)")
    ap.add_argument("--rounds-per-launch", type=int, default=10,
                    help="cartpole f32: rounds per pddp_round_nominal_f32 "
                         "launch (csrc/round_n4.hip: a workgroup owns its "
                         "trajectories, their attempts follow one another "
                         "without a launch boundary); 1: one launch per round; "
                         "0: the two launches per round of rounds 3-4.  With "
                         "more than one rank the exchange needs the round "
                         "boundary: at most --exchange-every rounds per launch")
    args = ap.parse_args()
    exchange_default = args.exchange_every is None
    if exchange_default:
        args.exchange_every = 1
    launch_ranks_if_asked(args)
    if args.workload == "mpc_bnn":
        return bench_mpc_bnn(args)
    if args.workload == "double_cartpole_gp":
        return bench_gp(args)
    if args.workload != "cartpole":
        return bench_bnn(args)

    import torch.distributed as dist
    world, rank, device = init_ranks()
    from pddp_amd import _native
    from pddp_amd.parallel import (gather_best_rollout, post_best_rollout,
                                   shard_bounds)

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    N = args.horizon or 100
    total = args.batch or 4096
    if args.scaling == "strong":
        lo, hi = shard_bounds(total, rank, world)
        B = hi - lo
    else:
        B, lo = total, rank * total
    s, z0, U, bound = make_cartpole_solver(B, N, dtype, device, rank,
                                           args.kernel_variant)
    n, m = s.n, s.m
    lib = _native.lib()
    K, W = args.steps, args.warmup
    R = args.repeats or 5
    n_iter = 1 << 30  # the fit loop never runs out inside the benchmark
    seen = ranks_seen(world, device)

    # R repetitions give the headline; R_EV more repetitions of the SAME K rounds
    # carry HIP events on the two dispatches for the per-kernel durations of
    # the roofline (the events cost ~4.5 us per kernel - 0.1205 against 0.1116
    # ms per round at B = 4096 - and are not part of the product's step)
    R_EV = 2
    pool_sweep, pool_search = EventPool(lib), EventPool(lib)
    pool_round = EventPool(lib)
    reps, reps_ev = [], []
    accepted_acc = torch.zeros((), dtype=torch.int64, device=device)
    # rounds per launch of the timed region (the exchange of the multi-GPU
    # path sits between launches)
    rpl = max(args.rounds_per_launch, 0)
    if world > 1 and exchange_default and rpl > 1:
        args.exchange_every = rpl  # (one exchange per launch)
    if world > 1 and args.exchange_every > 0:
        rpl = min(rpl, args.exchange_every)
    chunks_timed = []
    phase_us = None
    # (torch loads an operator's code object at its first use - 0.2 s once)
    accepted_acc += ((s.state == 1) | (s.state == 5)).sum()
    accepted_acc.zero_()
    # (does the one-launch round apply here?  cartpole f32, B <= 4096)
    s.set_nominal(z0, U)
    s.round(5e-6, 1e10, n_iter)
    round_kernel_ok = s._one_launch is True
    for rep in range(R + R_EV):
        with_events = rep >= R
        # the repetitions with events: first the timed region's own launches
        # with events on them, then the same rounds as TWO launches (sweep,
        # search + accept - the same device code) with events on each: the
        # sweep's own duration for the roofline of SURVEY 8(d)
        two_launch = rpl == 0 or not round_kernel_ok or rep == R + R_EV - 1
        s._one_launch = False if two_launch else None
        # every repetition times the same K rounds from the same nominal
        s.set_nominal(z0, U)
        for _ in range(W):
            s.round(5e-6, 1e10, n_iter)
        s.n_live.zero_()
        live0 = int(s.active.sum().item())
        one_launch = not two_launch
        chunk = max(rpl, 1) if one_launch else 1
        ev = [(pool_sweep.pair(), pool_search.pair())
              if with_events and two_launch else (None, None)
              for _ in range(K)]
        ev_round = [pool_round.pair() if with_events and one_launch else None
                    for _ in range((K + chunk - 1) // chunk)]
        if with_events and one_launch:
            # the workgroups' own clocks around their two phases (not in the
            # timed repetitions)
            s.phase_ticks = torch.zeros((B + 15) // 16, 2, dtype=torch.int64,
                                        device=device)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        i = 0
        while i < K:
            c = min(chunk, K - i)
            if one_launch:
                # the product's round(s) in one launch (ILQRSolver.rounds):
                # sweep from the nominal, line search, accept, c times
                if with_events:
                    chunks_timed.append(c)
                s.rounds(c, 5e-6, 1e10, n_iter, events=ev_round[i // chunk]) \
                    if c > 1 else s.round(5e-6, 1e10, n_iter,
                                          backward_events=ev_round[i // chunk])
            else:
                # the product's round as separate launches (ILQRSolver.round):
                # records of fresh nominals, sweep, fused line search + accept
                # + records of the accepted nominals
                s.round(5e-6, 1e10, n_iter, backward_events=ev[i][0],
                        search_events=ev[i][1])
            i += c
            if world > 1 and args.exchange_every > 0 and \
                    i % args.exchange_every == 0:
                # the exchange of the path: best rollout over RCCL - one pack
                # launch on this stream, the all-gather on a side stream
                # behind an event, no host synchronisation
                last_exchange = post_best_rollout(s.J_opt, s.Z, s.U, offset=lo)
            if with_events and two_launch:  # (no host sync: a device-side sum)
                # attempts of this round that were accepted (state 1 ACCEPTED,
                # 5 CONVERGED; every trajectory is live throughout the region)
                accepted_acc += ((s.state == 1) | (s.state == 5)).sum()
        if world > 1 and (args.exchange_every <= 0 or
                          K % args.exchange_every != 0):
            # one exchange per region / behind the last, shorter launch
            last_exchange = post_best_rollout(s.J_opt, s.Z, s.U, offset=lo)
        if world > 1:
            last_exchange.result()  # (this stream waits for the last gather)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        # units processed: attempts actually made (live trajectories per round)
        liveK = int(s.active.sum().item())
        cum_live = int(s.n_live.sum().item())  # sum over rounds of live-after
        attempted = live0 + cum_live - liveK   # sum over rounds of live-before
        total_attempted = attempted
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            t = torch.tensor([attempted], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            total_attempted = int(t.item())
        if s.phase_ticks is not None:
            phase_us = (s.phase_ticks.double().cpu().numpy() / 100.0) / K
            s.phase_ticks = None
        (reps_ev if with_events else reps).append(
            {"elapsed": elapsed, "attempted": attempted,
             "total_attempted": total_attempted, "live": [live0, liveK],
             "form": "two launches" if two_launch else
                     "%d rounds per launch" % chunk})
    s._one_launch = None

    order = sorted(range(R), key=lambda i: reps[i]["elapsed"])
    med = reps[order[R // 2]]
    elapsed, total_attempted = med["elapsed"], med["total_attempted"]
    # (the repetition with events on the separate sweep / search launches)
    attempted_all = reps_ev[-1]["attempted"]
    timed_form = reps[0]["form"]

    itemsize = 4 if dtype == torch.float32 else 8
    per_traj = algorithmic_bytes_per_trajectory(N, n, m, itemsize, True)
    d_sweep = np.array(pool_sweep.durations())
    d_search = np.array(pool_search.durations())
    launches = K
    # every timed launch swept `attempted_all / launches` trajectories on average
    sweep_bytes = attempted_all / launches * per_traj
    achieved = sweep_bytes / float(d_sweep.mean()) / 1e9
    # share of the attempts that were accepted (only those read the winner
    # back, write a new nominal and its records)
    accepted_share = float(accepted_acc.item()) / max(attempted_all, 1)
    # the sweep from the nominal evaluates its records itself: none written
    from_nominal = getattr(s, "_nominal_sweep", False) is True
    search_bytes = search_accept_bytes(attempted_all / launches, N, n, m,
                                       int(s.A), s.lay.stride, itemsize,
                                       accepted_share,
                                       records=not from_nominal)
    search_timed = getattr(s, "last_search_timed", None)
    # the timed region's own launches (pddp_round_nominal_f32: `c` rounds each)
    d_round = np.array(pool_round.durations())
    round_obj = None
    if len(d_round):
        per_round = d_round / np.array(chunks_timed[:len(d_round)], np.float64)
        round_bytes = sweep_bytes + search_bytes
        round_obj = {
            "kernel": "round_n4_kernel (csrc/round_n4.hip): sweep from the "
                      "nominal, then line search + accept, in the same "
                      "workgroups; %s" % timed_form,
            "rounds_per_launch": chunks_timed[:len(d_round)],
            "avg_launch_us": float(d_round.mean()) * 1e6,
            "avg_round_us": float(per_round.mean()) * 1e6,
            "min_round_us": float(per_round.min()) * 1e6,
            "algorithmic_bytes_per_round": round_bytes,
            "what_the_bytes_are": "the backward sweep's SURVEY 8(d) figure "
                                  "(%.1f MB: the records it would read) + the "
                                  "search's (%.1f MB: nominal in, candidates, "
                                  "costs and the accepted nominals out)" % (
                                      sweep_bytes / 1e6, search_bytes / 1e6),
            "achieved": round_bytes / float(per_round.mean()) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round_bytes / float(per_round.mean()) / 1e9 / HBM_PEAK_GBS,
        }
        if phase_us is not None:
            # rocprofv3 sees one kernel: the split is the workgroups' own
            # 100 MHz clock around their phases (pddp_round_nominal_f32
            # `phase_ticks`), per round, over the 256 workgroups.  Every
            # workgroup sweeps its 16 trajectories in `sweep_us` while all the
            # others do the same: the batch's sweep bandwidth is the agreed
            # bytes over the MEAN workgroup's sweep time
            sw_, se_ = phase_us[:, 0], phase_us[:, 1]
            round_obj["phases"] = {
                "source": "device clock (s_memrealtime) around the phases, "
                          "wavefront 0 of every workgroup, mean per round",
                "sweep_us": {"mean": float(sw_.mean()), "min": float(sw_.min()),
                             "max": float(sw_.max())},
                "search_accept_us": {"mean": float(se_.mean()),
                                     "min": float(se_.min()),
                                     "max": float(se_.max())},
                "sweep_achieved_GBs": sweep_bytes / (float(sw_.mean()) * 1e-6)
                                      / 1e9,
                "sweep_frac": sweep_bytes / (float(sw_.mean()) * 1e-6) / 1e9 /
                              HBM_PEAK_GBS,
            }

    # HBM traffic of the same kernels from rocprofv3 PMC passes (FETCH_SIZE and
    # WRITE_SIZE cannot share a pass, and counters cannot be read from inside
    # this process): taken from the committed summary of the profiled run of
    # this very command, see profiles/.
    traffic, traffic_search, traffic_round = None, None, None
    for tname in ("r05_pmc_traffic.json", "r04_pmc_traffic.json",
                  "r03_pmc_traffic.json",
                  "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", tname)
        if not (os.path.exists(tpath) and B == 4096 and N == 100
                and args.dtype == "f32"):
            continue
        try:
            with open(tpath) as fh:
                for kname, v in json.load(fh)["kernels"].items():
                    t = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"],
                         "source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / "
                                   "WRITE_SIZE, 2*FETCH+WRITE)" % tname}
                    if "riccati" in kname:
                        traffic = t
                    elif "line_search" in kname:
                        traffic_search = t
                    elif "round_n4_kernel<25u, true>" in kname:
                        # (the profiled command: --steps 10, one launch of ten
                        # rounds per repetition)
                        traffic_round = dict(t, rounds_per_launch=10,
                                             hbm_bytes_per_round=v[
                                                 "hbm_bytes_per_launch"] / 10)
            break
        except (OSError, KeyError, ValueError):
            traffic = traffic_search = traffic_round = None
    if round_obj is not None:
        round_obj["traffic"] = traffic_round

    out = None
    if rank == 0:
        out = {
            "metric": "pddp_iterations_per_sec",
            "value": total_attempted / elapsed,
            "unit": "trajectory-iterations/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "rccl_ranks_seen": seen,
            "repeats": R,
            "ms_per_step_min": reps[order[0]]["elapsed"] / K * 1e3,
            "ms_per_step_all": [r["elapsed"] / K * 1e3 for r in reps],
            "ms_per_step_with_kernel_events": [
                {"form": r["form"], "ms": r["elapsed"] / K * 1e3}
                for r in reps_ev],
            "config": {
                "workload": "BASELINE.json configs[1]: cartpole n=4 m=1, "
                            "known-dynamics iLQR, horizon=%d, batch=%d "
                            "trajectories per GPU, bounds +-10 (eig-clamp + "
                            "BoxQP branch), 10 line-search alphas" % (N, B),
                "batch_per_gpu": B, "horizon": N, "alphas": int(s.A),
                "unit_definition": "one iLQR attempt of one trajectory: "
                                   "derivative records when its nominal "
                                   "changed + backward sweep + line search + "
                                   "accept",
                "value_definition": "median of %d repetitions of the timed "
                                    "region (each: set_nominal, %d warm-up "
                                    "rounds, %d timed rounds)" % (R, W, K),
                # how the K timed rounds were issued
                "launch_form": timed_form,
                "rounds_per_launch": args.rounds_per_launch,
                # (ranks > 1) the best rollout is all-gathered every so many
                # rounds, between launches
                "exchange_every_rounds": args.exchange_every if world > 1
                else None,
                "batched_iterations_per_s": K / elapsed,
                "trajectory_timesteps_per_s": total_attempted * N / elapsed,
                "live_trajectories_start_end": med["live"],
                "backward_kernel_variant": args.kernel_variant,
                # the timed kernels' f32 arithmetic: reciprocal by v_rcp_f32
                # (1 ulp) where the reference divides, the BoxQP in its lean
                # closed form (sign-bit predicates; QpClosed and the
                # reference's loop behind it), own sincos; the IEEE-division
                # twins are kernel variants 6 / 8 / 16 / 20 / 24
                "arithmetic": "f32, v_rcp, lean BoxQP (fall-back: closed form "
                              "+ the reference's loop), branch-free sincos",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("backward Riccati sweep from the nominal "
                           "(riccati_n4_elem_kernel: one wavefront per four "
                           "trajectories, the derivative records evaluated "
                           "by its partner wavefront into LDS - "
                           "`algorithmic_bytes_per_launch` is SURVEY 8(d)'s "
                           "figure for the sweep that reads them, `traffic` "
                           "what this launch moves)") if from_nominal
                          else "backward Riccati sweep",
                # what bounds this launch: N dependent steps on wavefronts
                # that have a SIMD to themselves - the step's instruction
                # count at the lone-wavefront issue rate
                "latency_model": {
                    "steps": N, "instructions_per_step": 88,
                    "cycles_per_instruction_lone_wavefront": 5.0,
                    "what": "time = N x instructions x issue interval; HBM "
                            "and the matrix cores are idle",
                } if from_nominal else None,
                # the step's dependent work alone - products, transposes, lean
                # BoxQP, value update; operands in registers, no LDS reads, no
                # generator, no stores: tools/probe/riccati_floor_probe.hip,
                # profiles/r05_riccati_floor.txt (409 cycles per step: 22.35
                # us per 100 steps, launch included)
                # (a MODEL: round 5's probe of round 5's step on one box, not
                # re-measured by this run)
                "floor_model_us": 22.35 * N / 100.0 if from_nominal else None,
                "floor_model_source": "profiles/r05_riccati_floor.txt",
                "frac_of_floor_model": (22.35 * N / 100.0) /
                                       (float(d_sweep.mean()) * 1e6)
                                       if from_nominal else None,
                # the sweep is timed as a launch of its own (the last
                # repetition runs the rounds as two launches, events on each);
                # the timed region itself runs `timed_region_kernel`
                "measured_in": "two-launch repetition (same device code as "
                               "the round kernel's first phase)",
                "timed_region_kernel": round_obj,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "avg_launch_us": float(d_sweep.mean()) * 1e6,
                "median_launch_us": float(np.median(d_sweep)) * 1e6,
                "min_launch_us": float(d_sweep.min()) * 1e6,
                "launches_timed": int(launches),
                "algorithmic_bytes_per_launch": sweep_bytes,
                "traffic": traffic,
                "other_kernels": [{
                    "bound": "hbm",
                    "kernel": ("fused line search + accept "
                               "(line_search_lds_kernel<.., FUSED>; no "
                               "records: the sweep evaluates them)")
                              if from_nominal else
                              "fused line search + accept + records "
                              "(line_search_lds_kernel<.., FUSED>)",
                    "launch_timed": search_timed,
                    "avg_launch_us": float(d_search.mean()) * 1e6,
                    "min_launch_us": float(d_search.min()) * 1e6,
                    "accepted_share": accepted_share,
                    "algorithmic_bytes_per_launch": search_bytes,
                    "achieved": search_bytes / float(d_search.mean()) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": search_bytes / float(d_search.mean()) / 1e9 /
                            HBM_PEAK_GBS,
                    "traffic": traffic_search,
                    "traffic_frac": (
                        traffic_search["hbm_bytes_per_launch"] /
                        float(d_search.mean()) / 1e9 / HBM_PEAK_GBS
                        if traffic_search else None),
                }],
            },
        }
        if round_obj is not None:
            # The timed region's dominant kernel is the round kernel: IT is
            # the `roofline` object (bytes = the two phases' agreed figures,
            # duration = events on its launches; rocprofv3: round_n4_kernel).
            # The backward sweep - BASELINE's 0.60 target - is its first
            # phase: `backward_sweep.inside_this_launch` (the workgroups' own
            # clocks) and `.as_a_launch_of_its_own` (riccati_n4_elem_kernel in
            # the two-launch repetition: events / rocprofv3)
            alone = out["roofline"]
            alone.pop("timed_region_kernel", None)
            others = alone.pop("other_kernels", [])
            ph = round_obj.get("phases") or {}
            n_launch = max(len(round_obj["rounds_per_launch"]), 1)
            mean_chunk = sum(round_obj["rounds_per_launch"]) / n_launch
            top = {
                "bound": "hbm",
                "kernel": round_obj["kernel"],
                "achieved": round_obj["achieved"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round_obj["frac"],
                "avg_launch_us": round_obj["avg_launch_us"],
                "rounds_per_launch": round_obj["rounds_per_launch"],
                "avg_round_us": round_obj["avg_round_us"],
                "min_round_us": round_obj["min_round_us"],
                "launches_timed": n_launch,
                "algorithmic_bytes_per_launch":
                    round_obj["algorithmic_bytes_per_round"] * mean_chunk,
                "algorithmic_bytes_per_round":
                    round_obj["algorithmic_bytes_per_round"],
                "what_the_bytes_are": round_obj["what_the_bytes_are"],
                "traffic": round_obj.get("traffic"),
                "latency_model": {
                    "what": "both phases are chains of N dependent steps on "
                            "wavefronts that have a SIMD to themselves: time "
                            "= N x instructions per step x the lone "
                            "wavefront's issue interval (~5-6 cycles); HBM "
                            "and the matrix cores are idle (DESIGN.md 3.5b)",
                    "steps": N, "sweep_instructions_per_step": 88,
                    "rollout_instructions_per_step": 100},
                "backward_sweep": {
                    "inside_this_launch": None if not ph else {
                        "us_per_round": ph["sweep_us"],
                        "algorithmic_bytes": sweep_bytes,
                        "achieved": ph["sweep_achieved_GBs"],
                        "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ph["sweep_frac"], "source": ph["source"],
                        "search_accept_us_per_round": ph["search_accept_us"]},
                    "as_a_launch_of_its_own": alone},
                "other_kernels": others,
            }
            out["roofline"] = top
        if world == 1 and not args.no_points and B == 4096 and N == 100 and \
                args.dtype == "f32":
            # the points that defeat the caches, in the line
            out["roofline"]["points"] = [
                sweep_point(lib, B, N, dtype, device, args.kernel_variant,
                            cold=True, solver=(s, z0, U)),
                sweep_point(lib, 16384, N, torch.float32, device,
                            args.kernel_variant),
                sweep_point(lib, 4096, N, torch.float64, device,
                            args.kernel_variant),
            ]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline("cartpole", 0.1, N, bound)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_secondary and B == 4096 and N == 100 \
                and args.dtype == "f32":
            # BASELINE configs[2] / [3] (one GPU's shard, BNN and GP) / [4] in
            # the driver's line: five timed rounds each, configs[4] in full
            # (200 control steps of all 256 restarts)
            import copy
            del s
            torch.cuda.empty_cache()
            sec = []
            # (configs[2] a second time on the float64 kernels: three rounds)
            # (the GP a second time at the data-set size the reference's
            # outer loop produces - 300 rows: one round, 25 x the pair loop)
            for wl, k, w, dt_, gp_m in (("cartpole_bnn", 5, 1, "f32", None),
                                        ("cartpole_bnn", 3, 1, "f64", None),
                                        ("double_cartpole_bnn", 5, 1, "f32",
                                         None),
                                        ("double_cartpole_gp", 5, 1, "f32",
                                         None),
                                        ("double_cartpole_gp", 1, 1, "f32",
                                         300),
                                        ("mpc_bnn", 200, 2, "f32", None)):
                a2 = copy.copy(args)
                a2.workload, a2.steps, a2.warmup, a2.dtype = wl, k, w, dt_
                a2.gp_points = gp_m
                if gp_m:
                    a2.no_graph_replay = True
                a2.batch = a2.horizon = None
                try:
                    fn = {"mpc_bnn": bench_mpc_bnn,
                          "double_cartpole_gp": bench_gp}.get(wl, bench_bnn)
                    sec.append(fn(a2, emit=False))
                except Exception as e:  # (reported, not hidden)
                    sec.append({"workload": wl, "error": repr(e)[:300]})
                torch.cuda.empty_cache()
            out["secondary"] = sec
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
