#!/usr/bin/env python3
"""Headless counterparts of the reference's example scripts on the MI355X: the
same objects, the same calls, through `pddp_amd`; plotting and rendering are
left out.

  pendulum         examples/pendulum.py:125-175         PDDP, BNN [200, 200], N = 25
  cartpole         examples/cartpole.py:126-178         PDDP, BNN [200, 200], N = 25
  double_cartpole  examples/double_cartpole.py:128-175  PDDP, BNN [200, 200], N = 50
  experiment       examples/experiment.py:160-213       the same flow through
                   SampleProblems.<PROBLEM>.setup(DT), N = 25, DT = 0.1
  mpc_animation    examples/mpc_animation.py:24-39      iLQR, known dynamics, MPC

A PDDP flow = `PDDPController(env, model, cost, model_opts, training_opts)`,
`.train()`, `.fit(U, encoding DEFAULT, n_iterations, on_iteration, on_trial,
max_trials, u_min, u_max)`, then the feedback law `controller(z, i, encoding)`
driving the environment for N steps; `on_iteration` also rolls the plan out on
the true model (examples/utils.py:25-30 `rollout`).  The iLQR inside runs on
the HIP path: forward-mode derivative rollout and line search through the
fused network kernel, matrix-core backward sweep, device-resident accept /
regularisation state machine.

    python tools/run_example.py pendulum [--trials 3] [--iterations 10] [--train-iters 300]
    python tools/run_example.py mpc_animation [--steps 50]
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

# (module, prefix, N, dt, |u| bound, U0, model_opts, fit kwargs, train iters):
# the constants at the top of each reference script
PDDP_FLOWS = {
    "pendulum": dict(mod="pendulum", cls="Pendulum", N=25, dt=0.1, bound=2.5,
                     U0="randn", model_opts={}, fit={"n_iterations": 50},
                     n_iter=1000),
    "cartpole": dict(mod="cartpole", cls="Cartpole", N=25, dt=0.1, bound=10.0,
                     U0="uniform",
                     model_opts={"use_predicted_std": False,
                                 "infer_noise_variables": True},
                     fit={"n_iterations": 50}, n_iter=2000),
    "double_cartpole": dict(mod="double_cartpole", cls="DoubleCartpole", N=50,
                            dt=0.05, bound=20.0, U0="randn",
                            model_opts={"use_predicted_std": False,
                                        "infer_noise_variables": True},
                            fit={"n_iterations": 200, "tol": 0}, n_iter=1000),
}


def rollout(model, z0, U, encoding, **kwargs):
    """examples/utils.py:25-30: the plan's actions on a model, step by step."""
    Z = torch.empty(U.shape[0] + 1, z0.shape[-1], dtype=z0.dtype,
                    device=z0.device)
    Z[0] = z0.detach()
    for i in range(U.shape[0]):
        Z[i + 1] = model(Z[i], U[i], i, encoding=encoding, **kwargs).detach()
    return Z.detach()


# examples/experiment.py: one script for every sample problem, through the
# SampleProblems registry (UMAX per problem, :24-30; rendezvous is unbounded
# there, which its `U = (UMAX - UMIN) * rand + UMIN` cannot take - not offered)
EXPERIMENT_BOUND = {"CARTPOLE": 10.0, "DOUBLE_CARTPOLE": 20.0, "PENDULUM": 2.5}


def run_pddp(name, trials=3, iterations=None, train_iters=None, particles=100,
             device="cuda", quiet=True, seed=0, problem="CARTPOLE"):
    """One of the PDDP example scripts (`name` = "experiment": the flow of
    examples/experiment.py on `problem`).  Returns a dict with the cost
    history, the final iLQR state, the plan and the controlled episode."""
    torch.manual_seed(seed)
    enc = pddp.StateEncoding.DEFAULT
    if name == "experiment":
        f = dict(N=25, dt=0.1, bound=EXPERIMENT_BOUND[problem], U0="uniform",
                 model_opts={"use_predicted_std": False,
                             "infer_noise_variables": True},
                 fit={"n_iterations": 50}, n_iter=2000)
        env, cost, real_model = pddp.examples.SampleProblems[problem].setup(
            f["dt"])
        cost, real_model = cost.to(device), real_model.to(device)
        model_class = type(real_model)
    else:
        f = PDDP_FLOWS[name]
        mod = getattr(pddp.examples, f["mod"])
        cost = getattr(mod, f["cls"] + "Cost")().to(device)
        env = getattr(mod, f["cls"] + "Env")(dt=f["dt"])
        model_class = getattr(mod, f["cls"] + "DynamicsModel")
        real_model = model_class(f["dt"]).to(device)
    model = bnn_dynamics_model_factory(
        env.state_size, env.action_size, [200, 200],
        model_class.angular_indices, model_class.non_angular_indices,
    )(n_particles=particles).to(device)
    N = f["N"]
    umin = torch.tensor([-f["bound"]])
    umax = torch.tensor([f["bound"]])
    if f["U0"] == "uniform":
        U = (umax - umin) * torch.rand(N, model.action_size) + umin
    else:
        U = torch.randn(N, model.action_size)
    J_hist, reality = [], []

    def on_trial(trial, X, U_):
        if not quiet:
            print("trial %d: %d environment steps" % (trial + 1, X.shape[0]))

    def on_iteration(iteration, state, Z, U_, J_opt):
        J_hist.append(float(J_opt))
        if iteration % 10 == 9 or iteration == 0:
            ienc = pddp.StateEncoding.IGNORE_UNCERTAINTY
            x0 = pddp.utils.encoding.decode_mean(Z[0], enc)
            reality.append(rollout(real_model, x0, U_, ienc)[-1].cpu())
        if not quiet:
            print("  iteration %2d %-12s J = %10.4f"
                  % (iteration + 1, getattr(state, "name", state),
                     float(J_opt)))

    fit = dict(f["fit"])
    if iterations is not None:
        fit["n_iterations"] = iterations
    controller = pddp.controllers.PDDPController(
        env, model, cost, model_opts=f["model_opts"],
        training_opts={"n_iter": train_iters or f["n_iter"],
                       "learning_rate": 1e-3})
    controller.train()
    t0 = time.perf_counter()
    Z, U, state = controller.fit(
        U.to(device), encoding=enc, on_iteration=on_iteration,
        on_trial=on_trial, max_trials=trials, u_min=umin, u_max=umax,
        quiet=True, **fit)
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    env.reset()
    for i in range(N):  # the feedback controller on the plant
        z = env.get_state().encode(enc).to(device)
        u = controller(z, i, enc)
        env.apply(u.detach().cpu())
    final = env.get_state().mean()
    env.close()
    solver = getattr(controller, "_solver", None)
    path = getattr(getattr(solver, "plugin", None), "last_derivs_path", None)
    return {"J_hist": J_hist, "state": state, "Z": Z, "U": U, "fit_s": fit_s,
            "final_state": final, "reality": reality, "derivs_path": path}


def run_mpc_animation(steps=50, device="cuda", seed=0, graph=False):
    """examples/mpc_animation.py: cartpole, known dynamics, IGNORE_UNCERTAINTY,
    N = 25; one `fit(n_iterations=1, tol=0)`, then `steps` receding-horizon
    control steps `controller(z0, i, encoding, mpc=True, ...)` on the plant."""
    torch.manual_seed(seed)
    DT, N = 0.1, 25
    umax = torch.tensor([10.0])
    umin = -umax
    enc = pddp.StateEncoding.IGNORE_UNCERTAINTY
    cost = pddp.examples.cartpole.CartpoleCost()
    model = pddp.examples.cartpole.CartpoleDynamicsModel(DT)
    env = pddp.examples.cartpole.CartpoleEnv(dt=DT)
    controller = pddp.controllers.iLQRController(env, model, cost, graph=graph)
    U = (1e-1 * torch.randn(N, model.action_size)).to(device)
    controller.fit(U, encoding=enc, n_iterations=1, tol=0, u_min=umin,
                   u_max=umax)
    actions, plans = [], []
    for iteration in range(steps):
        if iteration == 0:
            env.reset()
        z0 = env.get_state().encode(enc).to(device)
        u = controller(z0, iteration, enc, mpc=True, u_min=umin, u_max=umax)
        env.apply(u.detach().cpu())
        actions.append(u.detach().cpu())
        plans.append(pddp.utils.encoding.decode_mean(
            controller._Z_nominal.detach(), enc).cpu())
    final = env.get_state().mean()
    env.close()
    return {"actions": torch.stack(actions), "plans": plans,
            "final_state": final}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("example", choices=list(PDDP_FLOWS) + ["experiment",
                                                           "mpc_animation"])
    ap.add_argument("--problem", default="CARTPOLE",
                    choices=list(EXPERIMENT_BOUND))
    ap.add_argument("--trials", type=int, default=3)
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--train-iters", type=int, default=None)
    ap.add_argument("--steps", type=int, default=50)
    a = ap.parse_args()
    if a.example == "mpc_animation":
        out = run_mpc_animation(a.steps)
        print("state after %d MPC steps:" % a.steps,
              [round(float(v), 3) for v in out["final_state"]])
        return
    out = run_pddp(a.example, a.trials, a.iterations, a.train_iters,
                   quiet=False, problem=a.problem)
    print("fit: %.1f s, final state %s, planned cost %.4f, derivative path %s"
          % (out["fit_s"], out["state"],
             out["J_hist"][-1] if out["J_hist"] else float("nan"),
             out["derivs_path"]))
    print("state after the controlled episode:",
          [round(float(v), 3) for v in out["final_state"]])


if __name__ == "__main__":
    main()
