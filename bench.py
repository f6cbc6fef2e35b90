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

prints ONE JSON line (rank 0).  For N > 1 launch through torch.distributed.run;
ranks own disjoint shards (weak scaling, no data-path collective) and exchange
their best rollout once through RCCL.
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

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_trajectory(N, n, m, itemsize, bounded):
    """BASELINE.md 3 / SURVEY.md 8(d): bytes the backward sweep must move."""
    per_step = 2 * n * n + 3 * n * m + n + 2 * m + m * m + (m if bounded else 0)
    return itemsize * (N * per_step + n + n * n)


def cpu_baseline(problem_name, dt, N, bound, seconds=12.0):
    """Times the oracle (plain C port of the reference's algorithm) on the
    host's cores on a bounded sample of the same workload; unit = the same
    trajectory-iterations/s.  One worker process per core (trajectories are
    independent, as on the GPU), at most 16 - a GPU box's CPU share; the
    workers are fresh interpreters that never touch the GPU."""
    import subprocess
    workers = max(1, min(16, os.cpu_count() or 1))
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
    return {"value": attempts / el, "unit": "trajectory-iterations/s",
            "cores": workers, "kind": "port",
            "per_core_value": attempts / el / workers,
            "sample": "%d cartpole trajectories x up to 8 iLQR iterations "
                      "(%d attempts) in %.1f s on %d worker processes, oracle "
                      "C port, fp32, host has %d cores"
                      % (trajs, attempts, el, workers, os.cpu_count())}


MFMA_F32_PEAK_TFLOPS = 157.0  # MI355X_MICROARCH.md: dense f32 matrix peak


def bench_bnn(args):
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
    from pddp_amd.parallel import gather_best_rollout
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(
            "cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.manual_seed(0)  # the same network on every rank
    if args.workload == "cartpole_bnn":
        from pddp_amd.examples import cartpole as ex
        CM, cost_cls = ex.CartpoleDynamicsModel, ex.CartpoleCost
        mean0, bound, tag = [0.0, 0.0, 3.14159, 0.0], 10.0, "configs[2]: cartpole"
        B = args.batch
        N = args.horizon
    else:
        from pddp_amd.examples import double_cartpole as ex
        CM, cost_cls = ex.DoubleCartpoleDynamicsModel, ex.DoubleCartpoleCost
        mean0, bound = [0.0, 0.0, 3.14159, 0.0, 3.14159, 0.0], 20.0
        tag = "configs[3]'s problem: double cartpole"
        B = args.batch if args.batch != 4096 else 1024
        N = args.horizon if args.horizon != 100 else 150
    D, m, P, A, H = CM.state_size, 1, 100, 10, 200
    n = D + D * (D + 1) // 2
    in_dim = len(CM.non_angular_indices) + 2 * len(CM.angular_indices) + m
    K = args.steps if args.steps != 30 else 3
    W = args.warmup if args.warmup != 5 else 1
    model = bnn_dynamics_model_factory(D, m, [H, H], CM.angular_indices,
                                       CM.non_angular_indices)(
        n_particles=P).to(dev).eval()
    with torch.no_grad():  # untrained weights: keep the dynamics gentle
        model.model.out.weight.mul_(0.05)
        model.model.out.bias.mul_(0.05)
    cost = cost_cls().to(dev)
    enc = pddp_amd.StateEncoding.DEFAULT
    plugin = TorchProblem(model, cost, enc, {"use_predicted_std": False,
                                             "infer_noise_variables": True}, {})
    s = ILQRSolver(None, B, N, torch.float32, dev, torch.tensor([-bound]),
                   torch.tensor([bound]), fit_alphas(torch.float32, dev),
                   plugin=plugin, n=n, m=m)
    g = torch.Generator().manual_seed(rank)
    mean = torch.tensor(mean0)
    z0 = torch.stack([pddp_amd.GaussianVariable(
        mean + 1e-2 * torch.randn(D, generator=g),
        var=1e-2 * torch.ones(D)).encode(enc) for _ in range(B)]).to(dev)
    s.set_nominal(z0, (0.1 * torch.randn(B, N, m, generator=g)).to(dev))
    n_iter = 1 << 30
    for _ in range(W):
        s.round(5e-6, 1e10, n_iter)
    s.n_live.zero_()
    live0 = int(s.active.sum().item())
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        s.round(5e-6, 1e10, n_iter)
    if world > 1:  # the one exchange of the path: best rollout over RCCL
        gather_best_rollout(s.J_opt, s.Z, s.U, offset=rank * B)
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
    F = torch.randn(B * P * 8, in_dim, device=dev)
    model.model._jvp_native(F, P, D, 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps):
        model.model._jvp_native(F, P, D, 8)
    e1.record()
    torch.cuda.synchronize(dev)
    dur = e0.elapsed_time(e1) * 1e-3 / reps
    flop = 2.0 * B * P * 8 * (in_dim * H + H * H + H * D)
    out = {
        "metric": "pddp_iterations_per_sec", "value": total_attempted / elapsed,
        "unit": "trajectory-iterations/s", "n_gpus": world, "steps": K,
        "warmup": W, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
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
            "kernel": "fused BNN network, forward-mode (bnn_mlp_kernel<200, 8, 8>)",
            "achieved": flop / dur * 1e-12, "peak": MFMA_F32_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": flop / dur * 1e-12 / MFMA_F32_PEAK_TFLOPS,
            "avg_launch_us": dur * 1e6,
            "algorithmic_flop_per_launch": flop, "traffic": None,
        },
        "cpu_baseline": None,
    }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="cartpole",
                    choices=["cartpole", "cartpole_bnn", "double_cartpole_bnn"],
                    help="cartpole = BASELINE configs[1] (the headline line); "
                         "cartpole_bnn = configs[2]; double_cartpole_bnn = "
                         "configs[3]'s problem with the reference's BNN model")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-variant", type=int, default=0,
                    help="backward kernel: 0 auto, 1 generic, 2 n4, 3 n4 fast")
    args = ap.parse_args()
    if args.workload != "cartpole":
        return bench_bnn(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(
            "cuda", local_rank))
    device = torch.device("cuda", local_rank)

    import pddp_amd
    from pddp_amd import _native
    from pddp_amd.controllers.solver import ILQRSolver
    from pddp_amd.examples import cartpole
    from pddp_amd.parallel import gather_best_rollout

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    B, N = args.batch, args.horizon
    enc = pddp_amd.StateEncoding.IGNORE_UNCERTAINTY
    model, cost = cartpole.CartpoleDynamicsModel(0.1), cartpole.CartpoleCost()
    prob = model.native_problem(enc, cost)
    n, m = prob.encoded_size, prob.action_size
    bound = 10.0
    u_min = torch.full((m,), -bound, dtype=dtype)
    u_max = torch.full((m,), bound, dtype=dtype)
    s = ILQRSolver(prob, B, N, dtype, device, u_min, u_max)
    g = torch.Generator().manual_seed(rank)
    z0 = (1e-2 * torch.randn(B, n, generator=g, dtype=torch.float64)).to(dtype)
    U = (0.1 * torch.randn(B, N, m, generator=g, dtype=torch.float64)).to(dtype)
    s.set_nominal(z0.to(device), U.to(device))

    lib = _native.lib()
    K, W = args.steps, args.warmup
    n_iter = 1 << 30  # the fit loop never runs out inside the benchmark

    def one_round(ev=None):
        # the product's round (ILQRSolver.round): records of fresh nominals,
        # sweep (events attached to its own dispatch: the kernel's duration),
        # fused line search + accept + records of the accepted nominals
        s.round(5e-6, 1e10, n_iter, variant=args.kernel_variant,
                backward_events=ev)

    for _ in range(W):
        one_round()
    torch.cuda.synchronize(device)
    events = []
    for _ in range(K):
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        lib.pddp_event_create(ctypes.byref(a))
        lib.pddp_event_create(ctypes.byref(b))
        events.append((a, b))
    s.n_live.zero_()
    live0 = int(s.active.sum().item())
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(K):
        one_round(events[i])
    if world > 1:  # the one exchange of the path: best rollout over RCCL
        lo = rank * B
        Jb, idx, Zb, Ub = gather_best_rollout(s.J_opt, s.Z, s.U, offset=lo)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # units processed: attempts actually made (live trajectories per round)
    liveK = int(s.active.sum().item())
    cum_live = int(s.n_live.sum().item())    # sum over rounds of live-after
    attempted = live0 + cum_live - liveK     # sum over rounds of live-before
    total_attempted = attempted
    if world > 1:
        t = torch.tensor([attempted], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_attempted = int(t.item())

    durs = []
    for a, b in events:
        ms = ctypes.c_float()
        lib.pddp_event_elapsed_ms(a, b, ctypes.byref(ms))
        durs.append(ms.value * 1e-3)
        lib.pddp_event_destroy(a)
        lib.pddp_event_destroy(b)
    itemsize = 4 if dtype == torch.float32 else 8
    per_traj = algorithmic_bytes_per_trajectory(N, n, m, itemsize, True)
    avg_dur = float(np.mean(durs))
    # every timed launch swept `attempted / K` trajectories on average
    achieved = (attempted / K) * per_traj / avg_dur / 1e9

    # HBM traffic of the same kernel from rocprofv3 PMC passes (FETCH_SIZE and
    # WRITE_SIZE cannot share a pass, and counters cannot be read from inside
    # this process): taken from the committed summary of the profiled run of
    # this very command, see profiles/.
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if os.path.exists(tpath) and B == 4096 and N == 100 and args.dtype == "f32":
        try:
            with open(tpath) as fh:
                for kname, v in json.load(fh)["kernels"].items():
                    if "riccati" in kname:
                        traffic = {"hbm_bytes_per_launch":
                                   v["hbm_bytes_per_launch"],
                                   "source": "profiles/r01_pmc_traffic.json "
                                             "(rocprofv3 --pmc FETCH_SIZE / "
                                             "WRITE_SIZE, 2*FETCH+WRITE)"}
        except (OSError, KeyError, ValueError):
            traffic = None

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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
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
                "batched_iterations_per_s": K / elapsed,
                "trajectory_timesteps_per_s": total_attempted * N / elapsed,
                "live_trajectories_start_end": [live0, liveK],
                "backward_kernel_variant": args.kernel_variant,
            },
            "roofline": {
                "bound": "hbm", "kernel": "backward Riccati sweep",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "avg_launch_us": avg_dur * 1e6,
                "min_launch_us": float(np.min(durs)) * 1e6,
                "algorithmic_bytes_per_launch": (attempted / K) * per_traj,
                "traffic": traffic,
                # what the memory system allows for this transfer with no
                # arithmetic at all (tools/probe/record_stream_probe.hip on a
                # cold cache, B = 4096, N = 100, fp32): informational
                "transfer_only_us": ({"plain_coalesced_read": 27.9,
                                      "this_kernels_streaming_pattern": 33.0,
                                      "source": "profiles/r01_record_stream_probe.txt"}
                                     if (B == 4096 and N == 100 and
                                         args.dtype == "f32") else None),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline("cartpole", 0.1, N, bound)
        else:
            out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
