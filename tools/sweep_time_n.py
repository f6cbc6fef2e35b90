"""Times the backward sweep alone for a given state size / dtype / variant on
random well-conditioned records (controller branch: bounds, eig clamp):
    python tools/sweep_time_n.py --n 14 --dtype f64 --variants 0 1"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pddp_amd import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=14)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--horizon", type=int, default=100)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--variants", type=int, nargs="+", default=[0, 1])
ap.add_argument("--chol", action="store_true")
a = ap.parse_args()
dt = torch.float64 if a.dtype == "f64" else torch.float32
B, N, n, m = a.batch, a.horizon, a.n, 1
lay = _native.record_layout(n, m)
gen = torch.Generator(device="cuda").manual_seed(0)
r = lambda *s: torch.randn(*s, generator=gen, device="cuda", dtype=dt)
F_z = torch.eye(n, device="cuda", dtype=dt) + 0.05 * r(B, N, n, n)
F_u = 0.3 * r(B, N, n, m)
L_z, L_u = r(B, N + 1, n), r(B, N, m)
R = 0.2 * r(B, N + 1, n, n)
L_zz = torch.eye(n, device="cuda", dtype=dt) + R @ R.transpose(-1, -2)
L_uz = 0.05 * r(B, N, m, n)
L_uu = 1.0 + 0.04 * r(B, N, m, m) ** 2
U = 0.5 * r(B, N, m)
rec = torch.empty(B, N + 1, lay.stride, dtype=dt, device="cuda")
p = _native.ptr
st = _native.stream_handle(rec.device)
_native.call("pddp_pack_records", dt, B, N, n, m, p(F_z), p(F_u), p(L_z), p(L_u),
             p(L_zz), p(L_uz), p(L_uu), p(U), p(rec), st)
del F_z, L_zz, R
u_min = -torch.ones(m, dtype=dt, device="cuda")
u_max = torch.ones(m, dtype=dt, device="cuda")
reg = torch.full((B,), 1e-3, dtype=torch.float64, device="cuda")
gains = torch.empty(B, N, lay.gain_stride, dtype=dt, device="cuda")
status = torch.empty(B, dtype=torch.int32, device="cuda")
nbytes = rec.numel() * rec.element_size() + gains.numel() * gains.element_size()
ref = None
for v in a.variants:
    def go():
        _native.call("pddp_riccati_backward_variant", dt, B, N, n, m, p(rec),
                     p(u_min), p(u_max), p(reg), 1 if a.chol else 0, None,
                     p(gains), p(status), st, v)
    try:
        go()
    except RuntimeError as e:
        print("variant", v, "unsupported:", e)
        continue
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for i in range(5):
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    g = gains.clone()
    if ref is None:
        ref = g
    err = ((g - ref).abs().max() / ref.abs().max()).item()
    print("n=%d %s B=%d N=%d variant %d: %.1f us (min of 5; %.2f TB/s of %d MB), "
          "status!=0: %d, max rel diff to first variant %.2e"
          % (n, a.dtype, B, N, v, min(ts), nbytes / min(ts) / 1e6, nbytes / 1e6,
             int((status != 0).sum()), err))
