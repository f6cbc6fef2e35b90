"""Barrier waits per role of the three-wavefront quad sweep (variant 21).
Needs the counters:
    make -C pddp_amd/csrc FLAGS_riccati_quad="-fno-slp-vectorize -DPDDP_QP_STATS"
(never ship that build)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402

lib = ctypes.CDLL(_native.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
B, N = 4096, 100
s, z0, U, _ = bench.make_cartpole_solver(B, N, torch.float32, "cuda", 0, 21)
s.set_nominal(z0, U)
for r in range(4):
    s.round(5e-6, 1e10, 1 << 30)
lib.pddp_debug_quad_stats(out, 1)
rounds = 10
for r in range(rounds):
    s.round(5e-6, 1e10, 1 << 30)
lib.pddp_debug_quad_stats(out, 1)
wg = (B + 15) // 16
for role, name in enumerate(("M (matrices)", "Q (scalars)", "P (producer)")):
    tot = out[3 + role] / (wg * rounds)
    wait = out[role] / (wg * rounds)
    print("%-13s %7.0f cycles per sweep, %5.0f per step, %4.0f of them at the "
          "barrier (%.0f %%)" % (name, tot, tot / N, wait / N,
                                 100.0 * wait / max(tot, 1)))
