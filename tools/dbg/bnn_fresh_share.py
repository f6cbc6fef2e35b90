"""The share of trajectories with a NEW nominal at each derivative rollout of
the BNN workloads (what a masked derivative rollout could skip):
python tools/dbg/bnn_fresh_share.py [mpc_bnn|cartpole_bnn|double_cartpole_bnn]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from pddp_amd.controllers import plugin as P
wl = sys.argv[1] if len(sys.argv) > 1 else "mpc_bnn"
orig = P.TorchProblem.derivs
log = []
def derivs(self, s, mask=None, set_state=True, in_graph=False):
    if not in_graph:
        log.append((s.B if mask is None else int(mask.sum().item()), s.B))
    return orig(self, s, mask, set_state, in_graph)
P.TorchProblem.derivs = derivs
sys.argv = ["bench.py", "--workload", wl, "--no-cpu-baseline"] + (
    ["--steps", "10"] if wl == "mpc_bnn" else ["--steps", "8", "--warmup", "1"])
try:
    bench.main()
except SystemExit:
    pass
import collections
fr = [a / b for a, b in log]
print("derivative rollouts:", len(log), " mean fresh share %.3f" % (sum(fr) / max(len(fr), 1)))
hist = collections.Counter(min(int(f * 10), 9) for f in fr)
print("histogram of the fresh share (tenths):", sorted(hist.items()))
print("first 40:", [a for a, _ in log[:40]])
