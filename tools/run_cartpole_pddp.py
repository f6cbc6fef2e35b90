#!/usr/bin/env python3
"""Headless counterpart of the reference's examples/cartpole.py on the MI355X:
the same objects and the same calls through `pddp_amd` - CartpoleEnv /
CartpoleCost / bnn_dynamics_model_factory([200, 200], 100 particles),
PDDPController(model_opts = use_predicted_std False, infer_noise_variables
True).fit(U, encoding DEFAULT, on_iteration, on_trial, max_trials, u_min, u_max),
then the feedback controller driving the environment
(reference examples/cartpole.py:126-178; plotting and rendering left out).

The iLQR inside every trial runs on the HIP path: forward-mode derivative
rollout and line search through the fused network kernel, matrix-core backward
sweep (n = 14), device-resident accept / regularisation state machine.

    python tools/run_cartpole_pddp.py [--trials 3] [--iterations 10] [--train-iters 300]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pddp_amd as pddp  # noqa: E402
import pddp_amd.examples  # noqa: E402
from pddp_amd.models.bnn import bnn_dynamics_model_factory  # noqa: E402

DT = 0.1  # time step (reference examples/cartpole.py:20-25)
N = 25    # horizon
ENCODING = pddp.StateEncoding.DEFAULT
UMIN, UMAX = torch.tensor([-10.0]), torch.tensor([10.0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--iterations", type=int, default=10)
    ap.add_argument("--train-iters", type=int, default=300)
    ap.add_argument("--particles", type=int, default=100)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    J_hist = []

    def on_trial(trial, X, U):
        print("trial %d: %d environment steps" % (trial + 1, X.shape[0]))

    def on_iteration(iteration, state, Z, U, J_opt):
        J_hist.append(float(J_opt))
        final = pddp.utils.encoding.decode_mean(Z[-1], ENCODING)
        print("  iteration %2d %-12s J = %10.4f  final mean %s"
              % (iteration + 1, getattr(state, "name", state), float(J_opt),
                 [round(float(v), 3) for v in final.cpu()]))

    cost = pddp.examples.cartpole.CartpoleCost().to(dev)
    env = pddp.examples.cartpole.CartpoleEnv(dt=DT)
    model_class = pddp.examples.cartpole.CartpoleDynamicsModel
    model = bnn_dynamics_model_factory(
        env.state_size, env.action_size, [200, 200],
        model_class.angular_indices, model_class.non_angular_indices,
    )(n_particles=a.particles).to(dev)

    U = ((UMAX - UMIN) * torch.rand(N, model.action_size) + UMIN).to(dev)
    controller = pddp.controllers.PDDPController(
        env, model, cost,
        model_opts={"use_predicted_std": False, "infer_noise_variables": True},
        training_opts={"n_iter": a.train_iters, "learning_rate": 1e-3})
    controller.train()
    t0 = time.perf_counter()
    Z, U, state = controller.fit(
        U, encoding=ENCODING, n_iterations=a.iterations,
        on_iteration=on_iteration, on_trial=on_trial, max_trials=a.trials,
        u_min=UMIN, u_max=UMAX, quiet=True)
    torch.cuda.synchronize()
    print("fit: %.1f s, final state %s, planned cost %.4f"
          % (time.perf_counter() - t0, state, J_hist[-1] if J_hist else float("nan")))
    plugin = controller._solver.plugin if hasattr(controller, "_solver") else None
    if plugin is not None:
        print("derivative rollout path:", getattr(plugin, "last_derivs_path", None))
    env.reset()
    for i in range(N):  # the feedback controller on the plant
        z = env.get_state().encode(ENCODING).to(dev)
        u = controller(z, i, ENCODING)
        env.apply(u.detach().cpu())
    print("state after the controlled episode:",
          [round(float(v), 3) for v in env.get_state().mean()])
    env.close()


if __name__ == "__main__":
    main()
