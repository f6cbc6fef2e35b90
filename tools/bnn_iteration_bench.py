"""BASELINE.json configs[2]: cartpole with a BNN dynamics model ([200, 200]
hidden, P = 100 particles), DEFAULT encoding (n = 14, m = 1), moment-matched
rollouts, horizon 100, B trajectories on one MI355X - one iLQR iteration
phase by phase:

  derivative rollout  F_z, F_u in forward mode (csrc/bnn_jvp.hip + the fused
                      network in JVP mode), cost derivatives (autograd, torch);
  backward sweep      matrix-core HIP kernel for n <= 14 (riccati_mfma16.hpp),
                      generic kernel beside it (eig-clamp + BoxQP branch);
  line search         A = 10 candidates, native moment-step + network kernels;
  accept              HIP state machine.

    python tools/bnn_iteration_bench.py [--batch 4096] [--horizon 100]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pddp_amd  # noqa: E402
from pddp_amd.controllers.ilqr import fit_alphas  # noqa: E402
from pddp_amd.controllers.plugin import TorchProblem  # noqa: E402
from pddp_amd.controllers.solver import ILQRSolver  # noqa: E402
from pddp_amd.examples import cartpole  # noqa: E402
from pddp_amd.models.bnn import bnn_dynamics_model_factory  # noqa: E402


def timed(fn, reps=1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--problem", default="cartpole",
                    choices=["cartpole", "double_cartpole"],
                    help="double_cartpole: BASELINE.json configs[3]'s problem "
                         "with the reference's own BNN model "
                         "(examples/double_cartpole.py:133-139; no GP exists "
                         "in the reference), e.g. --batch 1024 --horizon 150 "
                         "= one GPU's shard of 8192")
    ap.add_argument("--autograd-too", action="store_true",
                    help="also time the autograd Jacobians (small batches)")
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    if a.problem == "cartpole":
        CM, cost_cls = cartpole.CartpoleDynamicsModel, cartpole.CartpoleCost
        mean0, bound = [0.0, 0.0, 3.14159, 0.0], 10.0
    else:
        from pddp_amd.examples import double_cartpole as dc
        CM, cost_cls = dc.DoubleCartpoleDynamicsModel, dc.DoubleCartpoleCost
        mean0, bound = [0.0, 0.0, 3.14159, 0.0, 3.14159, 0.0], 20.0
    D = CM.state_size
    cls = bnn_dynamics_model_factory(D, 1, [200, 200], CM.angular_indices,
                                     CM.non_angular_indices)
    model = cls(n_particles=a.particles).to(dev).eval()
    with torch.no_grad():  # an untrained network: keep its dynamics gentle
        model.model.out.weight.mul_(0.05)
        model.model.out.bias.mul_(0.05)
    cost = cost_cls().to(dev)
    enc = pddp_amd.StateEncoding.DEFAULT
    B, N, A, P = a.batch, a.horizon, 10, a.particles
    n, m = D + D * (D + 1) // 2, 1
    in_dim = len(CM.non_angular_indices) + 2 * len(CM.angular_indices) + m
    G = 8  # network rows per (state, particle) in forward mode
    plugin = TorchProblem(model, cost, enc,
                          {"use_predicted_std": False,
                           "infer_noise_variables": True}, {})
    s = ILQRSolver(None, B, N, torch.float32, dev, torch.tensor([-bound]),
                   torch.tensor([bound]), fit_alphas(torch.float32, dev),
                   plugin=plugin, n=n, m=m)
    g = torch.Generator().manual_seed(0)
    z0 = torch.stack([pddp_amd.GaussianVariable(
        torch.tensor(mean0) + 1e-2 * torch.randn(D, generator=g),
        var=1e-2 * torch.ones(D)).encode(enc) for _ in range(B)]).to(dev)
    t_roll = timed(lambda: s.set_nominal(
        z0, (0.1 * torch.randn(B, N, m, generator=g)).to(dev)))
    out = {"workload": "%s BNN [200,200] P=%d, DEFAULT encoding n=%d m=1, "
                       "B=%d N=%d A=%d fp32" % (a.problem, P, n, B, N, A)}
    # --- derivative rollout
    opts = dict(dtype=torch.float32, device=dev)
    Fz = torch.zeros(B, N, n, n, **opts)
    Fu = torch.zeros(B, N, n, m, **opts)
    plugin._dyn_derivs_bnn(s, Fz, Fu)  # warm-up
    t_jvp = timed(lambda: plugin._dyn_derivs_bnn(s, Fz, Fu))
    s.derivs()
    t_derivs = timed(lambda: s.derivs())
    net_flop = 2.0 * B * P * G * (in_dim * 200 + 200 * 200 + 200 * D) * N
    out["derivative_rollout"] = {
        "total_s": t_derivs, "dynamics_jacobians_s": t_jvp,
        "cost_derivatives_and_packing_s": t_derivs - t_jvp,
        "network_TFLOPs": net_flop / t_jvp * 1e-12}
    if a.autograd_too:
        plugin.use_native_bnn_jvp = False
        s.derivs()
        out["derivative_rollout"]["autograd_total_s"] = timed(lambda: s.derivs())
        plugin.use_native_bnn_jvp = True
        s.derivs()
    # --- backward sweep: matrix-core kernel (n <= 14) vs the generic one
    reg = torch.full((B,), 1.0, dtype=torch.float64, device=dev)
    words = N * (2 * n * n + 3 * n * m + n + 2 * m + m * m + m) + n + n * n

    def sweep_us(variant, reps=20):
        s.backward(reg=reg, variant=variant)
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            s.backward(reg=reg, variant=variant)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    t_bwd = sweep_us(0) * 1e-6
    out["backward_sweep"] = {
        "s": t_bwd, "algorithmic_MB": B * 4 * words / 1e6,
        "GBps": B * 4 * words / t_bwd / 1e9,
        "frac_of_8TBps": B * 4 * words / t_bwd / 8e12,
        "failed": int((s.bwd_status != 0).sum()),
        "kernel": "riccati_mfma16 (auto)" if n <= 14 else "riccati_mfma32 (auto)",
        "generic_kernel_us": sweep_us(1, 3)}
    if n <= 14:
        out["backward_sweep"]["mfma16_ieee_division_us"] = sweep_us(14)
    # --- line search + accept
    s.line_search()
    t_ls = timed(lambda: s.line_search())
    ls_flop = 2.0 * B * A * P * (in_dim * 200 + 200 * 200 + 200 * D) * N
    out["line_search"] = {"s": t_ls, "network_TFLOPs": ls_flop / t_ls * 1e-12}
    t_acc = timed(lambda: s.accept(5e-6, 1e10, 1 << 30))
    out["accept_s"] = t_acc
    total = t_derivs + t_bwd + t_ls + t_acc
    out["iteration_s"] = total
    out["trajectory_iterations_per_s"] = B / total
    out["trajectory_timesteps_per_s"] = B * N / total
    out["nominal_rollout_torch_ops_s"] = t_roll
    print(json.dumps(out))


if __name__ == "__main__":
    main()
