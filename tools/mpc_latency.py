"""Receding-horizon loop latency (BASELINE.json configs[4]: cartpole, B=64
controllers, N=30, one iLQR iteration per control step, 11 MPC step sizes):
eager launches vs one hipGraph replay per round.

    python tools/mpc_latency.py [--batch 64] [--horizon 30] [--steps 200]
"""
import argparse
import json
import time

import torch

import pddp_amd
from pddp_amd.controllers.ilqr import mpc_alphas
from pddp_amd.controllers.solver import ILQRSolver
from pddp_amd.examples import cartpole as cp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--horizon", type=int, default=30)
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    B, N = a.batch, a.horizon
    model, cost = cp.CartpoleDynamicsModel(0.05), cp.CartpoleCost()
    prob = model.native_problem(pddp_amd.StateEncoding.IGNORE_UNCERTAINTY, cost)
    dev = "cuda"
    out = {}
    for graph, rps in ((False, 1), (True, 1), (True, 4)):
        torch.manual_seed(0)
        s = ILQRSolver(prob, B, N, torch.float32, dev, torch.tensor([-10.0]),
                       torch.tensor([10.0]), mpc_alphas(torch.float32, dev))
        z = torch.zeros(B, 4, device=dev)
        z[:, 2] = 3.14159
        z += 0.01 * torch.randn(B, 4, device=dev)
        U = 0.1 * torch.randn(B, N, 1, device=dev)
        rounds = 0

        def control_step(z, U):
            # ilqr.py:356-362: reset regularisation, one step() from z, apply
            # U[0], shift the plan; the "plant" is the model itself
            s.set_nominal(z, U)
            r = s.fit(n_iterations=1, graph=graph, rounds_per_sync=rps)
            z1 = s.Z[:, 1].clone()
            U1 = torch.cat([s.U[:, 1:], s.U[:, -1:]], 1)
            return z1, U1, r

        for _ in range(10):
            z, U, _ = control_step(z, U)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            z, U, r = control_step(z, U)
            rounds += r
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[("graph_sync%d" % rps) if graph else "eager"] = {
            "ms_per_control_step": 1e3 * dt / a.steps,
            "rounds_per_control_step": rounds / a.steps,
            "controller_steps_per_s": B * a.steps / dt,
        }
    print(json.dumps({"workload": "cartpole MPC B=%d N=%d A=11 fp32" % (B, N),
                      **out}))


if __name__ == "__main__":
    main()
