"""riccati_mfma32s (variants 26 / 27) against the one-wave kernel (14 / 15) and
the generic kernel (1): gains, status, masks; then timings at B = 1024, N = 150."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pddp_amd import _native
dt = torch.float32
p = _native.ptr


def problem(n, B, N, seed):
    m = 1
    lay = _native.record_layout(n, m)
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, device="cuda", dtype=dt)
    eye = torch.eye(n, device="cuda")
    F_z, F_u = eye + 0.05 * r(B, N, n, n), 0.3 * r(B, N, n, m)
    L_z, L_u = r(B, N + 1, n), r(B, N, m)
    R = 0.2 * r(B, N + 1, n, n)
    L_zz = eye + R @ R.transpose(-1, -2)
    L_uz = 0.05 * r(B, N, m, n)
    L_uu = 1.0 + 0.04 * r(B, N, m, m) ** 2
    U = 0.5 * r(B, N, m)
    rec = torch.empty(B, N + 1, lay.stride, dtype=dt, device="cuda")
    st = _native.stream_handle(rec.device)
    _native.call("pddp_pack_records", dt, B, N, n, m, p(F_z), p(F_u), p(L_z), p(L_u), p(L_zz), p(L_uz),
                 p(L_uu), p(U), p(rec), st)
    return lay, rec


def run(lay, rec, B, N, n, variant, bounded=True, active=None, regv=1e-3):
    m = 1
    st = _native.stream_handle(rec.device)
    u_min, u_max = -torch.ones(m, device="cuda"), torch.ones(m, device="cuda")
    reg = torch.full((B,), regv, dtype=torch.float64, device="cuda")
    gains = torch.full((B, N, lay.gain_stride), 7.0, device="cuda")
    status = torch.full((B,), -5, dtype=torch.int32, device="cuda")
    _native.call("pddp_riccati_backward_variant", dt, B, N, n, m, p(rec), p(u_min) if bounded else None,
                 p(u_max) if bounded else None, p(reg), 0, None if active is None else p(active), p(gains),
                 p(status), st, variant)
    torch.cuda.synchronize()
    return gains, status


ok = True
for n, B, N in ((27, 37, 11), (20, 9, 6), (15, 2, 5), (30, 5, 7), (27, 1, 1), (16, 3, 2), (29, 4, 12)):
    lay, rec = problem(n, B, N, n)
    act = (torch.arange(B, device="cuda") % 3 != 1).to(torch.uint8)
    a = act.bool()
    for bounded in (True, False):
        for regv in (1e-3, 1.0):
            ref, sr = run(lay, rec, B, N, n, 1, bounded, None, regv)
            for v_old, v_new in ((15, 27), (14, 26)):
                old, so = run(lay, rec, B, N, n, v_old, bounded, None, regv)
                new, sn = run(lay, rec, B, N, n, v_new, bounded, None, regv)
                part, sp = run(lay, rec, B, N, n, v_new, bounded, act, regv)
                e_old = float((old - ref).abs().max() / ref.abs().max())
                e_new = float((new - ref).abs().max() / ref.abs().max())
                good = (int(sn.abs().max()) == 0 and e_new < 2e-5 and torch.equal(part[a], new[a]) and
                        bool((part[~a] == 7.0).all()) and bool((sp[~a] == -5).all()))
                ok &= good
                print("n %2d B %3d N %2d bounded %d reg %g variant %d: err %.2e (one-wave %.2e) status %d  %s" % (
                    n, B, N, bounded, regv, v_new, e_new, e_old, int(sn.abs().max()), "ok" if good else "BAD"), flush=True)
print("ALL OK" if ok else "FAILURES")
if ok and len(sys.argv) > 1:
    m = 1
    for B, N, n in ((1024, 150, 27), (4096, 150, 27), (256, 150, 27), (2048, 150, 27), (1024, 100, 20)):
        lay, rec = problem(n, B, N, 1)
        st = _native.stream_handle(rec.device)
        u_min, u_max = -torch.ones(m, device="cuda"), torch.ones(m, device="cuda")
        reg = torch.full((B,), 1e-3, dtype=torch.float64, device="cuda")
        gains = torch.full((B, N, lay.gain_stride), 7.0, device="cuda")
        status = torch.full((B,), -5, dtype=torch.int32, device="cuda")
        for variant in (15, 27):
            def launch():
                _native.call("pddp_riccati_backward_variant", dt, B, N, n, m, p(rec), p(u_min), p(u_max), p(reg), 0,
                             None, p(gains), p(status), st, variant)
            for _ in range(3):
                launch()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(20):
                launch()
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 50
            bytes_ = B * (N + 1) * lay.stride * 4 + B * N * lay.gain_stride * 4
            print("B %5d N %3d n %2d variant %d: %.1f us per sweep, %.2f TB/s = %.2f of 8" % (
                B, N, n, variant, us, bytes_ / us / 1e6, bytes_ / us / 8e6), flush=True)
