"""Time of pddp_gp_step_* per row (double cartpole shape) against the number
of training points, with and without the Jacobian."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gp_native_check import make, rows  # noqa: E402
from pddp_amd import StateEncoding  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for dtype in (torch.float32, torch.float64):
    for M in (20, 60, 120):
        model = make("double_cartpole", M, dtype)
        enc = StateEncoding.DEFAULT
        for jac in (False, True):
            z, u = rows("double_cartpole", R, enc, dtype)
            if not model.native_ok(z, enc, jac):
                print("M=%d %s jac=%s: not covered" % (M, dtype, jac))
                continue
            for _ in range(20):  # (the clocks come down while the host
                # prepares inputs: the first launches after that run slow)
                model.native_step(z, u, enc, jacobian=jac)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                model.native_step(z, u, enc, jacobian=jac)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print("M=%3d %s R=%d jac=%-5s: %8.3f ms (%.2f us per row)" % (
                M, str(dtype)[6:], R, jac, dt * 1e3, dt / R * 1e6), flush=True)
