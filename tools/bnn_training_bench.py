"""BNN training (modules.py:131-198) on the GPU: eager Adam steps against the
captured hipGraph step (pddp_amd/models/bnn.py fit(graph=...)).
    python tools/bnn_training_bench.py [--rows 1000] [--iters 1000]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pddp_amd.examples import cartpole  # noqa: E402
from pddp_amd.models.bnn import bnn_dynamics_model_factory  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000)   # pddp.py:262-265
    ap.add_argument("--iters", type=int, default=1000)  # examples/*.py n_iter
    a = ap.parse_args()
    CM = cartpole.CartpoleDynamicsModel
    cls = bnn_dynamics_model_factory(4, 1, [200, 200], CM.angular_indices,
                                     CM.non_angular_indices)
    torch.manual_seed(0)
    X = torch.randn(a.rows, 4).cuda()
    U = torch.randn(a.rows, 1).cuda()
    dX = 0.1 * X + 0.2 * U
    out = {}
    for graph in (False, True):
        model = cls(n_particles=100).cuda()
        model.fit(X, U, dX, n_iter=20, quiet=True, graph=graph)  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.fit(X, U, dX, n_iter=a.iters, learning_rate=1e-3, quiet=True,
                  graph=graph)
        torch.cuda.synchronize()
        out["graph" if graph else "eager"] = {
            "s": time.perf_counter() - t0,
            "ms_per_step": (time.perf_counter() - t0) / a.iters * 1e3,
            "used_graph": model.last_fit_used_graph}
    print(json.dumps({"workload": "BNN [200,200] training, %d rows, batch 128, "
                                  "%d Adam steps" % (a.rows, a.iters), **out}))


if __name__ == "__main__":
    main()
