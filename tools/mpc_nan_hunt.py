"""bench.py's mpc_bnn loop (eager) with a finiteness check after every control
step: reports the first restart whose action or state stops being finite."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pddp_amd  # noqa: E402
from pddp_amd.examples import cartpole  # noqa: E402
from pddp_amd.models.bnn import bnn_dynamics_model_factory  # noqa: E402

dev = "cuda"
B, N, K, P = 256, 50, int(sys.argv[1]) if len(sys.argv) > 1 else 203, 100
CM = cartpole.CartpoleDynamicsModel
enc = pddp_amd.StateEncoding.DEFAULT
ienc = pddp_amd.StateEncoding.IGNORE_UNCERTAINTY
iu = torch.triu_indices(4, 4)
tri = (0.1 * torch.eye(4))[iu[0], iu[1]].to(dev)
torch.manual_seed(0)
model = bnn_dynamics_model_factory(
    4, 1, [200, 200], CM.angular_indices, CM.non_angular_indices)(
        n_particles=P).to(dev).eval()
with torch.no_grad():
    model.model.out.weight.mul_(0.05)
    model.model.out.bias.mul_(0.05)
cost = cartpole.CartpoleCost().to(dev)
plant = CM(0.1).to(dev)
ctrl = pddp_amd.controllers.iLQRController(
    None, model, cost, graph=False,
    model_opts={"use_predicted_std": False, "infer_noise_variables": True})
u_min, u_max = torch.tensor([-10.0]), torch.tensor([10.0])
g = torch.Generator().manual_seed(0)
ctrl._U_nominal = (0.1 * torch.randn(B, N, 1, generator=g)).to(dev)
x = (torch.tensor([0.0, 0.0, 3.14159, 0.0]) +
     1e-2 * torch.randn(B, 4, generator=g)).to(dev)
for step in range(K):
    z = torch.cat([x, tri.expand(B, -1)], -1)
    u = ctrl(z, 0, enc, mpc=True, u_min=u_min, u_max=u_max)
    s = ctrl._solver
    bad_u = ~torch.isfinite(u).all(-1)
    with torch.no_grad():
        xn = plant(x, u.clamp(-10.0, 10.0), 0, ienc)
    bad_x = ~torch.isfinite(xn).all(-1)
    if bool(bad_u.any()) or bool(bad_x.any()):
        b = int((bad_u | bad_x).nonzero()[0])
        print("step", step, "restart", b, "bad_u", int(bad_u.sum()), "bad_x",
              int(bad_x.sum()))
        print("x", x[b].tolist(), "u", u[b].tolist())
        print("J_opt", float(s.J_opt[b]), "state", int(s.state[b]), "iter",
              int(s.iter[b]), "mu", float(s.mu[b]), "bwd",
              int(s.bwd_status[b]) if hasattr(s, "bwd_status") else None)
        print("Z finite", bool(torch.isfinite(s.Z[b]).all()), "U finite",
              bool(torch.isfinite(s.U[b]).all()), "gains finite",
              bool(torch.isfinite(s.gains[b]).all()))
        print("U nominal head", s.U[b, :5, 0].tolist())
        print("Z[0]", s.Z[b, 0].tolist())
        nf = (~torch.isfinite(s.Z[b]).all(-1)).nonzero()
        print("first non-finite Z row", nf[:3].tolist())
        break
    x = xn
else:
    print("all finite after", K, "steps; rounds", ctrl._last_rounds)
