"""Eager rounds against hipGraph replays of the same round (cartpole, the
benchmark's workload): ms per round, from one solver state each."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = 30
out = {}
for mode in ("eager", "graph", "graph8"):
    s, z0, U, _ = bench.make_cartpole_solver(B, 100, torch.float32, "cuda", 0, 0)
    s.set_nominal(z0, U)
    for _ in range(5):
        s.round(5e-6, 1e10, 1 << 30)
    if mode == "graph":
        s.capture_round(5e-6, 1e10, 1 << 30)
        step = lambda: s.replay_round(False)
    elif mode == "graph8":  # eight rounds per graph launch
        torch.cuda.synchronize()
        g8 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g8):
            for _ in range(8):
                s.round(5e-6, 1e10, 1 << 30, always_derivs=True)
        step = None
    else:
        step = lambda: s.round(5e-6, 1e10, 1 << 30)
    ts = []
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "graph8":
            for _ in range(4):
                g8.replay()
            n = 32
        else:
            for _ in range(K):
                step()
            n = K
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3)
    out[mode] = sorted(ts)[2]
    print(mode, ["%.4f" % t for t in ts])
print(out)
