"""Why the closed-form BoxQP of the quad sweep (variant 17) asks for the loop:
counts per trajectory-step on the bench workload.  Needs the counters:

    make -C pddp_amd/csrc FLAGS_riccati_quad="-fno-slp-vectorize -DPDDP_QP_STATS"

(never ship that build: the atomics cost ~1 ms per sweep)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402

lib = ctypes.CDLL(_native.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
s, z0, U, _ = bench.make_cartpole_solver(4096, 100, torch.float32, "cuda", 0, 17)
s.set_nominal(z0, U)
lib.pddp_debug_quad_stats(out, 1)
names = ("traj-steps", "slow", "done0", "armijo/guard failed", "live on bound",
         "x1 on bound", "guard used", "not descent")
for r in range(30):
    s.round(5e-6, 1e10, 1 << 30)
    if r in (0, 1, 4, 9, 29):
        lib.pddp_debug_quad_stats(out, 1)
        tot = max(out[0], 1)
        print("round", r, " ".join("%s %.4f%%" % (n, 100.0 * out[i] / tot)
                                    if i else "%s %d" % (n, out[i])
                                    for i, n in enumerate(names)))
