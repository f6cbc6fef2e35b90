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
# variant 0: the sweep from the nominal (generator wavefronts; their two waves
# are summed under "P")
nominal = variant == 0
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
    if nominal:
        assert s.sweep_nominal()
    else:
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
if hasattr(lib, "pddp_debug_defer_marks"):
    mk = (ctypes.c_longlong * 8)()
    lib.pddp_debug_defer_marks(mk)
    for w, name in enumerate(("M", "generator")):
        t = [mk[4 * w + i] for i in range(4)]
        print(name, "workgroup 0, cycles from its begin: first phase %d, "
              "phases done %d, end %d" % (t[1] - t[0], t[2] - t[0],
                                          t[3] - t[0]))
