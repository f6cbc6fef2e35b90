"""Is pddp_gp_step deterministic run to run?  (It has to be: fixed-order
reductions, no atomics.)  Same rows, many launches, outputs compared bit for
bit, step and Jacobian."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import pddp_amd.examples as ex
from pddp_amd import GaussianVariable, StateEncoding
from pddp_amd.models.gp import gp_dynamics_model_factory
enc = StateEncoding.DEFAULT
for system in ("double_cartpole", "cartpole", "pendulum"):
    mod = getattr(ex, system)
    MC = [getattr(mod, k) for k in dir(mod) if k.endswith("DynamicsModel") and k != "DynamicsModel"][0]
    E, m = MC.state_size, 1
    for Md in (24, 20, 33, 64):
        for dtype in (torch.float64, torch.float32):
            g = torch.Generator().manual_seed(2)
            Xd = torch.randn(Md, E, generator=g, dtype=torch.float64)
            Ud = torch.randn(Md, m, generator=g, dtype=torch.float64)
            dXd = 0.1 * torch.randn(Md, E, generator=g, dtype=torch.float64)
            model = gp_dynamics_model_factory(E, m, MC.angular_indices, MC.non_angular_indices)().double().cuda()
            model.fit(Xd.cuda(), Ud.cuda(), dXd.cuda())
            model = model.to(dtype).eval()
            R = 70
            z = torch.stack([GaussianVariable(0.3 * torch.randn(E, generator=g, dtype=torch.float64),
                                              var=1e-2 * torch.ones(E, dtype=torch.float64)).encode(enc)
                             for _ in range(R)]).to(dtype).cuda()
            u = (0.3 * torch.randn(R, m, generator=g)).to(dtype).cuda()
            ref = model.native_step(z, u, enc).clone()
            refJ = [t.clone() for t in model.native_step(z, u, enc, jacobian=True)]
            bad = badJ = 0
            for rep in range(150):
                out = model.native_step(z, u, enc)
                bad += int(not torch.equal(out.view(torch.int64 if dtype == torch.float64 else torch.int32),
                                           ref.view(torch.int64 if dtype == torch.float64 else torch.int32)))
                if rep % 10 == 0:
                    o = model.native_step(z, u, enc, jacobian=True)
                    badJ += int(not all(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(o, refJ)))
            torch.cuda.synchronize()
            print("%-16s M %3d %-14s step differs in %3d of 150 launches, jacobian in %2d of 15; finite: %s" % (
                system, Md, str(dtype), bad, badJ, bool(torch.isfinite(ref).all())), flush=True)
