import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from pddp_amd import _native
lib = _native.lib()
s, z0, U, _ = bench.make_cartpole_solver(4096, 100, torch.float32, "cuda", 0, 0)
def region(one, events, K=30):
    s._one_launch = None if one else False
    s.set_nominal(z0, U)
    for _ in range(5):
        s.round(5e-6, 1e10, 1 << 30)
    pa, pb = bench.EventPool(lib), bench.EventPool(lib)
    ev = [(pa.pair(), pb.pair()) if events else (None, None) for _ in range(K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = []
    for i in range(K):
        s.round(5e-6, 1e10, 1 << 30, backward_events=ev[i][0], search_events=ev[i][1])
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if events:
        pa.durations(); pb.durations()
    return el / K * 1e6, [round(t * 1e6) for t in ts[:6]]
for one, events in ((True, False), (False, False), (False, True), (False, True), (True, False), (False, True)):
    print(one, events, region(one, events))
