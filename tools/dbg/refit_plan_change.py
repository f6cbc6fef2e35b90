"""How much the MPC plan changes across resample / refit in
test_bnn_graphs_follow_model_resample_and_refit (eager run)."""
import sys
import torch
sys.path[:0] = ["tests", "."]
import pddp_amd
from test_gpu_parity import _bnn_mpc_controller
B, N = 32, 20
enc = pddp_amd.StateEncoding.DEFAULT
iu = torch.triu_indices(4, 4)
tri = (0.1 * torch.eye(4))[iu[0], iu[1]].cuda()
u_min, u_max = torch.tensor([-10.0]), torch.tensor([10.0])
ctrl, plant, x = _bnn_mpc_controller(B, N, False, P=32, H=64)
model = ctrl.model
z = torch.cat([x, tri.expand(B, -1)], -1)
us = []
mpc = lambda: us.append(ctrl(z, 0, enc, mpc=True, u_min=u_min, u_max=u_max).clone())
mpc()
torch.manual_seed(11)
model.resample()
mpc()
g = torch.Generator().manual_seed(5)
Xd = torch.randn(256, 4, generator=g).cuda()
Ud = torch.randn(256, 1, generator=g).cuda()
dXd = 0.05 * torch.randn(256, 4, generator=g).cuda()
torch.manual_seed(12)
model.fit(Xd, Ud, dXd, n_iter=8, batch_size=64, quiet=True, graph=False)
model.eval()
mpc()
print("max |u|", [float(u.abs().max()) for u in us])
print("resample changes the plan by", float((us[1] - us[0]).abs().max()))
print("refit changes the plan by", float((us[2] - us[1]).abs().max()))
