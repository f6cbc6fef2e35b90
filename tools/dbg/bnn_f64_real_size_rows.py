"""Per-quantity errors of the float64 BNN kernels against the reference's
float64 outputs at the real size (tests/golden/bnn_cartpole_real_size.npz), and
the line search's error per time step.  GPU box: python tools/dbg/bnn_f64_real_size_rows.py"""
import sys
import numpy as np
import torch
sys.path[:0] = ["tests", "."]
from test_gpu_parity import _bnn_real_size_run
raw = {}
rows = _bnn_real_size_run(torch.float64, raw)
for r in rows[:-1]:
    print(r["what"], r["r"], "%.3e %.3e %.3e" % (r["hip_vs_f64"], r["hip_vs_f32"], r["ref32_vs_f64"]))
print(rows[-1])
g = raw["g"]
for r in range(raw["Uc"].shape[0]):
    U64 = g["f64/%d/ls/U_new" % r]
    Z64 = g["f64/%d/ls/Z_new" % r]
    print("r", r, "shapes", raw["Uc"][r].shape, U64.shape, raw["Zc"][r].shape, Z64.shape)
    eu = np.abs(raw["Uc"][r] - U64).reshape(U64.shape[0], -1).max(1)
    ez = np.abs(raw["Zc"][r] - Z64).reshape(Z64.shape[0], -1).max(1)
    print(" U err per t:", " ".join("%.1e" % v for v in eu))
    print(" Z err per t:", " ".join("%.1e" % v for v in ez))
    ea = np.abs(raw["Uc"][r] - U64).reshape(U64.shape[0], U64.shape[1], -1).max((0, 2))
    print(" U err per alpha:", " ".join("%.1e" % v for v in ea))
