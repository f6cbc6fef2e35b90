"""Duration of the sweep from the nominal (pddp_sweep_nominal_f32,
riccati_n4_elem.hpp: generator inline / on wavefronts of its own / auto) and of
the recorded deferred sweep (variant 25) on the same fresh nominal, events on
the dispatches; gains against each other:
python tools/nominal_sweep_time.py [B ...]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402

lib = _native.lib()
for B in [int(v) for v in sys.argv[1:]] or [4096]:
    s, z0, U, _ = bench.make_cartpole_solver(B, 100, torch.float32, "cuda", 0, 0)
    s.set_nominal(z0, U)
    s.mu.fill_(1.0)
    s.derivs()
    out = {}
    for name in ("elem", "elem_inl", "elem_ovl", "recorded"):
        pool = bench.EventPool(lib)
        lib.pddp_sweep_nominal_kernel({"elem_inl": 3, "elem_ovl": 4}.get(name, 0))
        s.gains.zero_()
        for i in range(24):
            ev = pool.pair() if i >= 4 else None
            if name != "recorded":
                s.fresh.fill_(1)
                assert s.sweep_nominal(events=ev)
            else:
                s._rec_stale = False
                s.backward(active=s.active, variant=25, events=ev)
        torch.cuda.synchronize()
        d = np.array(pool.durations()) * 1e6
        out[name] = (s.gains.clone(), s.bwd_status.clone(), s.L.clone(),
                     s.J_opt.clone())
        print("B %6d %-9s mean %.1f us  min %.1f us  (status != 0: %d)" % (
            B, name, d.mean(), d.min(), int((s.bwd_status != 0).sum())))
        raw = ctypes.CDLL(_native.LIB_PATH)
        if name.startswith("elem") and hasattr(raw, "pddp_debug_elem_marks"):
            mk = (ctypes.c_longlong * 8)()  # (-DPDDP_ELEM_MARKS build)
            raw.pddp_debug_elem_marks(mk)
            t = [mk[i] for i in range(8)]
            print("  wave 0 of workgroup 0 (last launch): first block after %d "
                  "cycles, blocks %d cycles, end after %d; over the launches: "
                  "generator passes %d cycles, steps %d (%.0f per step), "
                  "block barriers %d" % (
                      t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] // 24,
                      t[5] // 24, t[5] / 2400.0, t[6] // 24))
    lib.pddp_sweep_nominal_kernel(0)
    g = {k: v[0].double() for k, v in out.items()}
    ok = (out["elem"][1] == 0) & (out["recorded"][1] == 0)
    sc = float(g["recorded"][ok].abs().max())
    for a, b in (("elem", "recorded"), ("elem_inl", "elem_ovl")):
        e = (g[a][ok] - g[b][ok]).abs().amax(dim=(1, 2)) / sc
        print("  gains %s vs %s: max %.2e  median %.2e   status equal: %s" % (
            a, b, float(e.max()), float(e.median()),
            bool(torch.equal(out[a][1], out[b][1]))))
    print("  L inline vs overlapped: %.2e   J_opt: %.2e" % (
        float((out["elem_inl"][2] - out["elem_ovl"][2]).abs().max()),
        float(((out["elem_inl"][3] - out["elem_ovl"][3]).abs() /
               out["elem_ovl"][3].abs()).max())))
