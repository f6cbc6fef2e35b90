"""A round of the fit loop (sweep + fused search / accept) with the sweep from
the nominal (pddp_sweep_nominal_*) against the round on records, for the known-
dynamics sample problems: python tools/nominal_round_time.py [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch
import pddp_amd
from pddp_amd.controllers.solver import ILQRSolver
from pddp_amd.utils.encoding import StateEncoding

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
CASES = {"pendulum": (0.1, 50, 2.5, [0, 0]), "double_cartpole": (0.05, 60, 20.0, [0, 0, np.pi, 0, np.pi, 0]),
         "cartpole": (0.1, 100, 10.0, [0, 0, 0, 0])}
for problem, (dt_, N, bound, mean0) in CASES.items():
    mod = getattr(pddp_amd.examples, problem)
    model = [getattr(mod, k) for k in dir(mod) if k.endswith("DynamicsModel") and k != "DynamicsModel"][0](dt_)
    cost = [getattr(mod, k) for k in dir(mod) if k.endswith("Cost") and k != "AugmentedQRCost"][0]()
    prob = model.native_problem(StateEncoding.IGNORE_UNCERTAINTY, cost)
    n, m = prob.encoded_size, prob.action_size
    for td in (torch.float32, torch.float64):
        res = {}
        for nominal in (True, False):
            rng = np.random.RandomState(0)
            s = ILQRSolver(prob, B, N, td, "cuda", torch.full((m,), -bound, dtype=td), torch.full((m,), bound, dtype=td))
            if not nominal:
                s._nominal_sweep = False
            elif not s._nominal_sweep_possible():
                continue
            else:
                s._nominal_sweep = None  # (also where round() would not take it)
            z0 = torch.from_numpy(np.asarray(mean0) + 1e-2 * rng.randn(B, n)).to(td).cuda()
            U = torch.from_numpy(0.1 * rng.randn(B, N, m)).to(td).cuda()
            s.set_nominal(z0, U)
            for _ in range(5):
                s.round(5e-6, 1e10, 1 << 30)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            K = 30
            for _ in range(K):
                s.round(5e-6, 1e10, 1 << 30)
            e1.record()
            torch.cuda.synchronize()
            res[nominal] = e0.elapsed_time(e1) / K * 1e3
        print("%-16s %-8s B %d N %d: round on records %.1f us%s" % (
            problem, str(td).split(".")[-1], B, N, res[False],
            "" if True not in res else ", from the nominal %.1f us (%.2fx)" % (res[True], res[False] / res[True])), flush=True)
