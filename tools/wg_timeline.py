"""Where the time of the two launches of a configs[1] round goes, workgroup by
workgroup (a -DPDDP_WG_TIMELINE build of the library, see tools/build_variant.sh):
every workgroup of pddp_sweep_nominal_f32 and of pddp_search_accept_f32 records
the chip-wide 100 MHz wall clock at entry, after its prologue, after its chain
and at exit, plus where it ran (XCC, CU) and the shader clock around it.

    PDDP_HIP_LIB=pddp_amd/lib_tl/libpddp_hip.so python tools/wg_timeline.py [B]

Prints, for the LAST round of a short fit: the spread of the entries, the
per-phase durations (min / median / p99 / max over the workgroups), the shader
clock each workgroup saw, the idle gap between the two launches, and the
launch-level envelope (first entry to last exit) against the events'."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402

lib = _native.lib()
raw = ctypes.CDLL(_native.LIB_PATH)
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
N = 100
s, z0, U, _ = bench.make_cartpole_solver(B, N, torch.float32, "cuda", 0, 0)
s.set_nominal(z0, U)


def q(x):
    x = np.asarray(x, np.float64)
    return "min %7.2f  med %7.2f  p99 %7.2f  max %7.2f" % (
        x.min(), np.median(x), np.percentile(x, 99), x.max())


def fetch(name):
    buf = (ctypes.c_longlong * (1024 * 12))()
    getattr(raw, name)(buf)
    return np.array(buf[:], np.int64).reshape(1024, 12)


def rounds_us(solver, one_launch, n=200):
    """Mean duration of a round in a fit from a fresh nominal (wall clock
    around n rounds, no events)."""
    import time
    solver._one_launch = None if one_launch else False
    solver.set_nominal(z0, U)
    for _ in range(5):
        solver.round(5e-6, 1e10, 1 << 30)
    torch.cuda.synchronize()
    t_ = time.perf_counter()
    for _ in range(n):
        solver.round(5e-6, 1e10, 1 << 30)
    torch.cuda.synchronize()
    return (time.perf_counter() - t_) / n * 1e6


def multi_us(solver, R, n=200):
    import time
    solver._one_launch = None
    solver.set_nominal(z0, U)
    for _ in range(5):
        solver.round(5e-6, 1e10, 1 << 30)
    torch.cuda.synchronize()
    t_ = time.perf_counter()
    for _ in range(n // R):
        solver.rounds(R, 5e-6, 1e10, 1 << 30)
    torch.cuda.synchronize()
    return (time.perf_counter() - t_) / (n // R * R) * 1e6


for rep in range(2):
    print("round, mean of 200 (fresh nominal): one launch %.2f us, two "
          "launches %.2f us; 2 / 5 / 10 / 20 rounds per launch: %.2f / %.2f / "
          "%.2f / %.2f us" % (rounds_us(s, True), rounds_us(s, False),
                            multi_us(s, 2), multi_us(s, 5), multi_us(s, 10),
                            multi_us(s, 20)))
s._one_launch = False if "--two" in sys.argv else None
s.set_nominal(z0, U)
rounds = 12
pool_b, pool_s = bench.EventPool(lib), bench.EventPool(lib)
with_events = "--events" in sys.argv
for r in range(rounds):
    if with_events:
        s.round(5e-6, 1e10, 1 << 30, backward_events=pool_b.pair(),
                search_events=pool_s.pair())
    else:  # (an event is a packet of its own between two launches)
        s.round(5e-6, 1e10, 1 << 30)
torch.cuda.synchronize()
if with_events:
    print("events: sweep %s us, search %s us" % (
        np.round(np.array(pool_b.durations()) * 1e6, 1).tolist()[-4:],
        np.round(np.array(pool_s.durations()) * 1e6, 1).tolist()[-4:]))
nwg = (B + 15) // 16
if s._one_launch:
    b1 = (ctypes.c_longlong * (1024 * 12))()
    b2 = (ctypes.c_longlong * (1024 * 12))()
    raw.pddp_debug_round_timeline(b1, b2)
    sw = np.array(b1[:], np.int64).reshape(1024, 12)[:nwg]
    se = np.array(b2[:], np.int64).reshape(1024, 12)[:nwg]
else:
    sw = fetch("pddp_debug_elem_timeline")[:nwg]
    se = fetch("pddp_debug_search_timeline")[:nwg]
us = lambda ticks: np.asarray(ticks, np.float64) / 100.0  # 100 MHz -> us
t0 = sw[:, 0].min()
print("B = %d, %d workgroups; last round of %d; one launch: %s" % (
    B, nwg, rounds, s._one_launch))
print("== sweep (riccati_n4_elem_kernel)")
print("  entry after the first entry   [us] " + q(us(sw[:, 0] - t0)))
print("  generator: entry              [us] " + q(us(sw[:, 4] - t0)))
print("  generator: block 0 written    [us] " + q(us(sw[:, 5] - sw[:, 4])))
print("  entry -> first step           [us] " + q(us(sw[:, 1] - sw[:, 0])))
print("  first step -> last step done  [us] " + q(us(sw[:, 2] - sw[:, 1])))
print("  last step -> exit             [us] " + q(us(sw[:, 3] - sw[:, 2])))
print("  generator exit after sweep's  [us] " + q(us(sw[:, 7] - sw[:, 3])))
print("  exit after the first entry    [us] " + q(us(sw[:, 3] - t0)))
cyc = (sw[:, 11] - sw[:, 10]).astype(np.float64)
dur = us(sw[:, 3] - sw[:, 0])
print("  shader clock seen            [GHz] " + q(cyc / dur / 1e3))
print("  cycles entry -> exit               " + q(cyc))
xcc = sw[:, 9] & 15
for x in sorted(set(xcc.tolist())):
    sel = xcc == x
    print("    XCC %d: %3d workgroups, chain %s" % (
        x, int(sel.sum()), q(us(sw[sel, 2] - sw[sel, 1]))))
cu = (sw[:, 8] >> 8) & 15
se_id = (sw[:, 8] >> 13) & 7
print("  distinct (XCC, SE, CU) triples: %d" % len(set(zip(xcc.tolist(), se_id.tolist(), cu.tolist()))))
env_sweep = us(sw[:, 3].max() - t0)
print("  envelope first entry -> last exit: %.2f us" % env_sweep)
print("== search + accept (line_search_lds_kernel)")
t1 = se[:, 0].min()
print("  idle: last sweep exit -> first search entry: %.2f us" % us(t1 - sw[:, 3].max()))
print("  entry after the first entry   [us] " + q(us(se[:, 0] - t1)))
print("  entry -> staged               [us] " + q(us(se[:, 1] - se[:, 0])))
print("  staged -> rollouts done       [us] " + q(us(se[:, 2] - se[:, 1])))
print("  rollouts done -> exit         [us] " + q(us(se[:, 3] - se[:, 2])))
has = se[:, 10] > se[:, 9]
print("    rollouts done -> decided    [us] " + q(us(se[:, 8] - se[:, 2])))
print("    decided -> fence + barrier  [us] " + q(us(se[:, 9] - se[:, 8])))
if has.any():
    print("    -> rows requested, gains out[us] " + q(us(se[has, 10] - se[has, 9])))
    print("    -> exit                     [us] " + q(us(se[has, 3] - se[has, 10])))
print("  helper: exit after main's     [us] " + q(us(se[:, 7] - se[:, 3])))
print("  exit after the first entry    [us] " + q(us(se[:, 3] - t1)))
print("  envelope first entry -> last exit: %.2f us" % us(max(se[:, 3].max(), se[:, 7].max()) - t1))
print("== round: first sweep entry -> last search exit: %.2f us" % us(max(se[:, 3].max(), se[:, 7].max()) - t0))
# a fused launch would let every workgroup go on as soon as ITS sweep is done:
fused = (sw[:, 3] - sw[:, 0]) + (se[:, 3] - se[:, 1])
print("== per workgroup: own sweep + own (staged -> exit) [us] " + q(us(fused)))
