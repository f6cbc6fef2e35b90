"""Barrier waits per role of the deferred four-wavefront sweep (variant 25).
Needs the counters:
    make -C pddp_amd/csrc FLAGS_riccati_defer="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -DPDDP_QP_STATS"
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 25
N = 100
s, z0, U, _ = bench.make_cartpole_solver(B, N, torch.float32, "cuda", 0, variant)
s.set_nominal(z0, U)
for r in range(4):
    s.round(5e-6, 1e10, 1 << 30)
lib.pddp_debug_defer_stats(out, 1)
seg = (ctypes.c_ulonglong * 32)()
lib.pddp_debug_defer_seg(seg, 1)
rounds = 10
for r in range(rounds):
    s.backward(active=s.active, variant=variant)
lib.pddp_debug_defer_stats(out, 1)
wg = (B + 15) // 16
for role, name in enumerate(("M (matrices)", "Q (scalars)", "Y (vectors)",
                             "P (producer)")):
    tot = out[4 + role] / (wg * rounds)
    wait = out[role] / (wg * rounds)
    print("%-13s %7.0f cycles per sweep, %5.0f per phase, %4.0f of them at the "
          "barrier (%.0f %%)" % (name, tot, tot / (N + 2), wait / (N + 2),
                                 100.0 * wait / max(tot, 1)))
lib.pddp_debug_defer_seg(seg, 1)
for role, name in enumerate("MQYP"):
    print(name, "segments (cycles per phase; 7 = last stamp -> barrier):",
          " ".join("%d:%.0f" % (i, seg[role * 8 + i] / (wg * rounds * (N + 2)))
                   for i in range(8)))
