"""Isolation run of the GP rollout kernel on the double cartpole (E = 6)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import pddp_amd.examples as ex
from pddp_amd import GaussianVariable, StateEncoding
from pddp_amd.controllers.ilqr import fit_alphas
from pddp_amd.controllers.plugin import TorchProblem
from pddp_amd.controllers.solver import ILQRSolver
from pddp_amd.models.gp import gp_dynamics_model_factory
mod = ex.double_cartpole
MC = mod.DoubleCartpoleDynamicsModel
enc = StateEncoding.DEFAULT
E, m = 6, 1
for dtype in (torch.float64, torch.float32):
    g = torch.Generator().manual_seed(2)
    Md = 24
    Xd = torch.randn(Md, E, generator=g, dtype=torch.float64)
    Ud = torch.randn(Md, m, generator=g, dtype=torch.float64)
    dXd = 0.1 * torch.randn(Md, E, generator=g, dtype=torch.float64)
    model = gp_dynamics_model_factory(E, m, MC.angular_indices, MC.non_angular_indices)().double().cuda()
    model.fit(Xd.cuda(), Ud.cuda(), dXd.cuda())
    model = model.to(dtype).eval()
    n = E + E * (E + 1) // 2
    B, N = 7, 6
    z0 = torch.stack([GaussianVariable(0.3 * torch.randn(E, generator=g, dtype=torch.float64),
                                       var=1e-2 * torch.ones(E, dtype=torch.float64)).encode(enc)
                      for _ in range(B)]).to(dtype).cuda()
    U0 = (0.3 * torch.randn(B, N, m, generator=g)).to(dtype).cuda()
    bound = torch.tensor([2.0], dtype=dtype)
    plugin = TorchProblem(model, mod.DoubleCartpoleCost().to(dtype).cuda(), enc, {}, {})
    s = ILQRSolver(None, B, N, dtype, "cuda", -bound, bound, fit_alphas(dtype, "cuda"), plugin=plugin, n=n, m=m)
    s.set_nominal(z0, U0)
    s.derivs()
    s.mu.fill_(1.0)
    s.backward(active=s.active)
    torch.cuda.synchronize()
    print(dtype, "setup ok, status", s.bwd_status.tolist(), flush=True)
    for rep in range(20):
        s.line_search(active=s.active)
        torch.cuda.synchronize()
        print(" rep", rep, "Jc finite", bool(torch.isfinite(s.Jc).all()), float(s.Jc.abs().max()), flush=True)
print("done")
