"""Times the backward sweep per (gain branch, bounded, kernel variant) on one
derivative rollout - what the BoxQP / Cholesky twins cost on top of the bare
Riccati recursion.  python tools/sweep_ablation.py [--batch 4096]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pddp_amd  # noqa: E402
from pddp_amd.controllers.solver import ILQRSolver  # noqa: E402
from pddp_amd.examples import cartpole as cp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--variants", default="",
                    help="comma-separated kernel variants (default: all)")
    ap.add_argument("--sweep-only", action="store_true")
    a = ap.parse_args()
    td = torch.float32 if a.dtype == "f32" else torch.float64
    model, cost = cp.CartpoleDynamicsModel(0.05), cp.CartpoleCost()
    prob = model.native_problem(pddp_amd.StateEncoding.IGNORE_UNCERTAINTY, cost)
    s = ILQRSolver(prob, a.batch, a.horizon, td, "cuda",
                   torch.tensor([-10.0], dtype=td), torch.tensor([10.0], dtype=td))
    torch.manual_seed(0)
    z0 = torch.zeros(a.batch, 4, dtype=td)
    z0[:, 2] = 3.14159
    s.set_nominal((z0 + 0.01 * torch.randn(a.batch, 4, dtype=td)).cuda(),
                  (0.1 * torch.randn(a.batch, a.horizon, 1, dtype=td)).cuda())
    s.derivs()
    reg = torch.full((a.batch,), 1.0, dtype=torch.float64, device="cuda")
    out = {}
    variants = (2, 3, 6, 7, 8, 9) if a.dtype == "f32" else (2, 6, 8)
    if a.variants:
        variants = tuple(int(v) for v in a.variants.split(","))
    for variant in variants:
        for branch, bounded in ((0, False), (0, True), (1, False), (1, True)):
            if variant in (18,) and not bounded:
                continue
            if False:
                continue
            for _ in range(3):
                s.backward(reg=reg, branch=branch, bounded=bounded, variant=variant)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                s.backward(reg=reg, branch=branch, bounded=bounded, variant=variant)
            e1.record()
            torch.cuda.synchronize()
            out["v%d_%s%s" % (variant, "chol" if branch else "eig",
                              "_box" if bounded else "")] = round(
                                  e0.elapsed_time(e1) / 20 * 1e3, 1)
    # the other kernels of a round, timed alone on the same state
    s.backward(reg=reg)
    for name, fn in (() if a.sweep_only else
                     (("line_search", lambda: s.line_search()),
                      ("derivs", lambda: s.derivs()),
                      ("rollout", lambda: s.nominal_rollout()))):
        for _ in range(3):
            fn()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    print(json.dumps({"B": a.batch, "N": a.horizon, "dtype": a.dtype, "us": out}))


if __name__ == "__main__":
    main()
