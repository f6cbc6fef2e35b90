"""Time marks of the sweep from the nominal INSIDE the fit loop (needs the
-DPDDP_QP_MARKS build, see tools/nominal_sweep_time.py): cycles until the first
phase and per phase, of workgroup 0 of the loop's last sweep."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402

s, z0, U, _ = bench.make_cartpole_solver(4096, 100, torch.float32, "cuda", 0, 0)
s.set_nominal(z0, U)
raw = ctypes.CDLL(_native.LIB_PATH)
mk = (ctypes.c_longlong * 8)()
odd = (ctypes.c_ulonglong * 2)()
for r in range(24):
    s.round(5e-6, 1e10, 1 << 30)
    if r >= 18:
        # (the search kernel of round r ran after this round's sweep; read the
        # marks of the sweep of the NEXT round by running only the sweep)
        raw.pddp_debug_defer_odd(odd, 1)
        assert s.sweep_nominal()
        raw.pddp_debug_defer_marks(mk)
        raw.pddp_debug_defer_odd(odd, 1)
        print("  of %d (workgroup, phase) pairs, role Q left the lean BoxQP in "
              "%d and ran the reference's loop in %d" % (256 * 102, odd[0],
                                                         odd[1]))
        t = [mk[i] for i in range(4)]
        g = [mk[4 + i] for i in range(4)]
        print("round %d: M first phase after %d cycles, phases %d (%.0f each); "
              "generator block 0 done after %d" % (
                  r, t[1] - t[0], t[2] - t[1], (t[2] - t[1]) / 102.0,
                  g[1] - g[0]))
