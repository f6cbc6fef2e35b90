"""What the per-iteration exchange costs a rank's fit loop at the bench's
batch: rounds with `post_best_rollout` after each, against rounds alone, and
against the torch form of the same selection (parallel.pack_best).  World of
one (the pack launch and the buffer rotation; the all-gather itself runs on a
side stream and is not on this stream's path)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pddp_amd.parallel import pack_best, post_best_rollout  # noqa: E402

s, z0, U, _ = bench.make_cartpole_solver(4096, 100, torch.float32, "cuda", 0, 0)
K = 60
for mode in ("rounds alone", "with pddp_pack_best", "with the torch selection"):
    s.set_nominal(z0, U)
    for _ in range(5):
        s.round(5e-6, 1e10, 1 << 30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        s.round(5e-6, 1e10, 1 << 30)
        if mode == "with pddp_pack_best":
            post_best_rollout(s.J_opt, s.Z, s.U, offset=0)
        elif mode == "with the torch selection":
            pack_best(s.J_opt, s.Z, s.U, 0)
    torch.cuda.synchronize()
    print("%-26s %.4f ms per round" % (mode, (time.perf_counter() - t0) / K * 1e3))
