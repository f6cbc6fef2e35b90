"""BASELINE.json configs[2] in the small: cartpole with a BNN dynamics model
([200, 200] hidden, P particles), DEFAULT encoding (n = 14), moment-matched
line-search rollouts of B trajectories x A = 10 step sizes over N steps through
the plugin path - as N + 1 moment-step launches with the fused network kernel
in between (pddp_bnn_moment_step_f32 + pddp_bnn_mlp_f32), as torch ops around
the fused network kernel, and as torch ops on library GEMMs.

    python tools/bnn_rollout_bench.py [--batch 256] [--horizon 100]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--particles", type=int, default=100)
    ap.add_argument("--native-only", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    CM = cartpole.CartpoleDynamicsModel
    cls = bnn_dynamics_model_factory(4, 1, [200, 200], CM.angular_indices,
                                     CM.non_angular_indices)
    model = cls(n_particles=a.particles).to(dev).eval()
    cost = cartpole.CartpoleCost().to(dev)
    enc = pddp_amd.StateEncoding.DEFAULT
    B, N, A = a.batch, a.horizon, 10
    n = 14
    plugin = TorchProblem(model, cost, enc,
                          {"use_predicted_std": False,
                           "infer_noise_variables": True}, {})
    s = ILQRSolver(None, B, N, torch.float32, dev, torch.tensor([-10.0]),
                   torch.tensor([10.0]), fit_alphas(torch.float32, dev),
                   plugin=plugin, n=n, m=1)
    z0 = pddp_amd.GaussianVariable(
        torch.tensor([0.0, 0.0, 3.14159, 0.0]),
        var=1e-2 * torch.ones(4)).encode(enc).to(dev)
    s.set_nominal(z0.unsqueeze(0).expand(B, -1).contiguous(),
                  0.1 * torch.randn(B, N, 1, device=dev))
    s.gains.normal_(0, 1e-2)
    out = {"workload": "cartpole BNN [200,200] P=%d DEFAULT encoding, B=%d "
                       "N=%d A=%d fp32" % (a.particles, B, N, A),
           "rows_per_step": B * A * a.particles}
    flop = 2.0 * B * A * a.particles * (6 * 200 + 200 * 200 + 200 * 8) * N
    modes = [("native_rollout", True, True)]
    if not a.native_only:
        modes += [("torch_ops_fused_network", False, True),
                  ("torch_ops_library_gemms", False, False)]
    for key, rollout_native, net_native in modes:
        plugin.use_native_bnn = rollout_native
        model.model.use_native = net_native
        s.line_search()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.line_search()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[key] = {"line_search_s": dt,
                    "candidate_steps_per_s": B * A * N / dt,
                    "network_TFLOPs": flop / dt * 1e-12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
