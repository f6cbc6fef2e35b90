"""Runs the bench workload's fit rounds with two sweep-kernel variants and
compares the per-trajectory controller state after every round.
python tools/variant_ab.py --a 7 --b 9 --rounds 35"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pddp_amd  # noqa: E402
from pddp_amd.controllers.solver import ILQRSolver  # noqa: E402
from pddp_amd.examples import cartpole  # noqa: E402


def make(B, N, dtype):
    model, cost = cartpole.CartpoleDynamicsModel(0.1), cartpole.CartpoleCost()
    prob = model.native_problem(pddp_amd.StateEncoding.IGNORE_UNCERTAINTY, cost)
    s = ILQRSolver(prob, B, N, dtype, "cuda", torch.full((1,), -10.0, dtype=dtype),
                   torch.full((1,), 10.0, dtype=dtype))
    g = torch.Generator().manual_seed(0)
    z0 = (1e-2 * torch.randn(B, 4, generator=g, dtype=torch.float64)).to(dtype)
    U = (0.1 * torch.randn(B, N, 1, generator=g, dtype=torch.float64)).to(dtype)
    s.set_nominal(z0.cuda(), U.cuda())
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", type=int, default=7)
    ap.add_argument("--b", type=int, default=9)
    ap.add_argument("--rounds", type=int, default=35)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=100)
    a = ap.parse_args()
    sa = make(a.batch, a.horizon, torch.float32)
    sb = make(a.batch, a.horizon, torch.float32)
    for r in range(a.rounds):
        sa.round(5e-6, 1e10, 1 << 30, variant=a.a)
        sb.round(5e-6, 1e10, 1 << 30, variant=a.b)
        torch.cuda.synchronize()
        same_state = int((sa.state == sb.state).sum())
        same_mu = int((sa.mu == sb.mu).sum())
        dJ = float(((sa.J_opt - sb.J_opt).abs() / sa.J_opt.abs().clamp_min(1e-30)).max())
        dg = float((sa.gains - sb.gains).abs().max())
        st = (int((sa.bwd_status != 0).sum()), int((sb.bwd_status != 0).sum()))
        acc = (int((sa.state == 1).sum()), int((sb.state == 1).sum()))
        print("round %2d same state %d mu %d  max rel dJ %.2e  max |dgains| %.2e "
              "bwd fail %s accepted %s" % (r, same_state, same_mu, dJ, dg, st, acc))


if __name__ == "__main__":
    main()
