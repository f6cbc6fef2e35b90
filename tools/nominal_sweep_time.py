"""Duration of the sweep from the nominal (pddp_sweep_nominal_f32) and of the
recorded deferred sweep (variant 25) on the same fresh nominal, events on the
dispatches:  python tools/nominal_sweep_time.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s, z0, U, _ = bench.make_cartpole_solver(B, 100, torch.float32, "cuda", 0, 0)
lib = _native.lib()
s.set_nominal(z0, U)
s.mu.fill_(1.0)
s.derivs()
for name in ("nominal", "recorded"):
    pool = bench.EventPool(lib)
    for i in range(24):
        ev = pool.pair() if i >= 4 else None
        if name == "nominal":
            s.fresh.fill_(1)
            assert s.sweep_nominal(events=ev)
        else:
            s._rec_stale = False
            s.backward(active=s.active, variant=25, events=ev)
    torch.cuda.synchronize()
    d = np.array(pool.durations()) * 1e6
    print("%-9s mean %.1f us  min %.1f us  (status != 0: %d)" % (
        name, d.mean(), d.min(), int((s.bwd_status != 0).sum())))
    import ctypes
    raw = ctypes.CDLL(_native.LIB_PATH)
    if hasattr(raw, "pddp_debug_defer_marks"):  # (-DPDDP_QP_MARKS build)
        mk = (ctypes.c_longlong * 8)()
        raw.pddp_debug_defer_marks(mk)
        t = [mk[i] for i in range(4)]
        print("  wave M of workgroup 0: first phase after %d cycles, phases "
              "%d cycles (%.0f each)" % (t[1] - t[0], t[2] - t[1],
                                         (t[2] - t[1]) / 102.0))
