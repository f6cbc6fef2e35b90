"""BASELINE.json configs[4]: receding-horizon loop (examples/mpc_animation.py)
on cartpole with a BNN dynamics model, horizon 50, 256 restarts x 200 control
steps, one MI355X.  Every control step is `iLQRController.forward(mpc=True)`
(ilqr.py:318-362): reset the regularisation, one fit iteration from the
measured state (derivative rollout, backward sweep, 11-alpha line search,
accept), emit U[0], shift the plan; the "plant" here is the model's own mean
prediction plus noise (the reference steps a gym env; no env on the GPU box).

    python tools/bnn_mpc_bench.py [--restarts 256] [--horizon 50] [--steps 200]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pddp_amd  # noqa: E402
from pddp_amd.examples import cartpole  # noqa: E402
from pddp_amd.models.bnn import bnn_dynamics_model_factory  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--restarts", type=int, default=256)
    ap.add_argument("--horizon", type=int, default=50)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--particles", type=int, default=100)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    CM = cartpole.CartpoleDynamicsModel
    cls = bnn_dynamics_model_factory(4, 1, [200, 200], CM.angular_indices,
                                     CM.non_angular_indices)
    model = cls(n_particles=a.particles).to(dev).eval()
    with torch.no_grad():  # untrained network: keep its dynamics gentle
        model.model.out.weight.mul_(0.05)
        model.model.out.bias.mul_(0.05)
    cost = cartpole.CartpoleCost().to(dev)
    plant = CM(0.1).to(dev)  # the true cartpole steps the state
    enc = pddp_amd.StateEncoding.DEFAULT
    B, N = a.restarts, a.horizon
    ctrl = pddp_amd.controllers.iLQRController(
        None, model, cost,
        model_opts={"use_predicted_std": False, "infer_noise_variables": True})
    u_min, u_max = torch.tensor([-10.0]), torch.tensor([10.0])
    ctrl._U_nominal = 0.1 * torch.randn(B, N, 1, device=dev)  # mpc_animation.py:27
    x = (torch.tensor([0.0, 0.0, 3.14159, 0.0]) +
         1e-2 * torch.randn(B, 4)).to(dev)
    ienc = pddp_amd.StateEncoding.IGNORE_UNCERTAINTY

    rounds = [0]

    def on_iteration(*args):
        rounds[0] += 1

    def control_step():
        nonlocal x
        z = _encode(x)
        u = ctrl(z, 0, enc, mpc=True, u_min=u_min, u_max=u_max,
                 on_iteration=on_iteration)
        with torch.no_grad():
            x = plant(x, u.clamp(-10.0, 10.0), 0, ienc)

    eye_tri = None

    def _encode(xb):
        # z = mean | triu(chol) with var 1e-2: chol = 0.1 I (gym_env.py:75-85)
        nonlocal eye_tri
        if eye_tri is None:
            U = 0.1 * torch.eye(4)
            iu = torch.triu_indices(4, 4)
            eye_tri = U[iu[0], iu[1]].to(dev)
        return torch.cat([xb, eye_tri.expand(xb.shape[0], -1)], -1)

    for _ in range(3):
        control_step()
    torch.cuda.synchronize()
    rounds[0] = 0
    t0 = time.perf_counter()
    for _ in range(a.steps):
        control_step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    path = getattr(ctrl._solver.plugin, "last_derivs_path", None)
    print(json.dumps({
        "workload": "BASELINE.json configs[4]: MPC, cartpole BNN [200,200] "
                    "P=%d DEFAULT encoding, horizon %d, %d restarts x %d "
                    "control steps, fp32" % (a.particles, N, B, a.steps),
        "s_total": dt, "ms_per_control_step": dt / a.steps * 1e3,
        "restart_control_steps_per_s": B * a.steps / dt,
        "rounds_per_control_step": rounds[0] / a.steps,
        "derivative_path": path}))


if __name__ == "__main__":
    main()
