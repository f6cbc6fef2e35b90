"""pddp_gp_step_* (csrc/gp_step.hip) against the torch module it replaces
(pddp_amd/models/gp.py): the moment-matched step and its Jacobian by autograd,
per system, encoding and dtype - the largest deviations (time per row:
tools/gp_step_time.py)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from pddp_amd import StateEncoding  # noqa: E402
from pddp_amd.models.gp import gp_dynamics_model_factory  # noqa: E402
from pddp_amd.utils.encoding import encode  # noqa: E402

SYSTEMS = {"pendulum": (2, 1, [0]), "cartpole": (4, 1, [2]),
           "double_cartpole": (6, 1, [1, 2])}


def make(system, M, dtype, seed=0):
    D, m, ang = SYSTEMS[system]
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(M, D, generator=g, dtype=torch.float64)
    U = torch.randn(M, m, generator=g, dtype=torch.float64)
    dX = 0.3 * torch.sin(X @ torch.randn(D, D, generator=g,
                                          dtype=torch.float64)) + 0.2 * U
    model = gp_dynamics_model_factory(D, m, ang)().double()
    model.fit(X, U, dX)
    return model.to(dtype).cuda()


def rows(system, R, encoding, dtype, seed=1):
    D, m, _ = SYSTEMS[system]
    g = torch.Generator().manual_seed(seed)
    mean = 0.5 * torch.randn(R, D, generator=g, dtype=torch.float64)
    A = 0.2 * torch.randn(R, D, D, generator=g, dtype=torch.float64)
    C = A @ A.transpose(-1, -2) + 1e-3 * torch.eye(D, dtype=torch.float64)
    z = encode(mean, C=C, encoding=encoding)
    u = torch.randn(R, m, generator=g, dtype=torch.float64)
    return z.to(dtype).cuda(), u.to(dtype).cuda()


def torch_step(model, z, u, encoding, jac):
    model.use_native = False
    try:
        if not jac:
            with torch.no_grad():
                return model(z, u, 0, encoding)
        R, n = z.shape
        m = u.shape[1]
        zu = torch.cat([z, u], -1)
        rep = zu.unsqueeze(1).expand(R, n, n + m).reshape(R * n, n + m)
        rep = rep.detach().clone().requires_grad_()
        zn = model(rep[:, :n], rep[:, n:], 0, encoding)
        eye = torch.eye(n, dtype=z.dtype, device=z.device).repeat(R, 1)
        J, = torch.autograd.grad(zn, rep, eye)
        J = J.reshape(R, n, n + m)
        return zn.reshape(R, n, n)[:, 0].detach(), J[:, :, :n], J[:, :, n:]
    finally:
        model.use_native = True


if __name__ == "__main__":
    for system in SYSTEMS:
        for dtype in (torch.float64, torch.float32):
            model = make(system, 40, dtype)
            for encoding in (1, 2, 3, 4):
                enc = StateEncoding(encoding)
                z, u = rows(system, 24, enc, dtype)
                assert model.native_ok(z, enc, True)
                ref, Fz_r, Fu_r = torch_step(model, z, u, enc, True)
                out, Fz, Fu = model.native_step(z, u, enc, jacobian=True)
                out0 = model.native_step(z, u, enc)
                torch.cuda.synchronize()
                sc = lambda a, b: float((a - b).abs().max() /
                                        (b.abs().max() + 1e-30))
                print("%-16s %-8s enc %d  step %.2e (no jac %.2e)  Fz %.2e  "
                      "Fu %.2e" % (system, str(dtype)[6:], encoding,
                                   sc(out, ref), sc(out0, ref), sc(Fz, Fz_r),
                                   sc(Fu, Fu_r)), flush=True)
