"""configs[4] (BNN MPC): how many of the 256 restarts have a NEW nominal at the
start of each round of the captured fit loop - the rows a masked derivative
rollout could skip."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from pddp_amd.controllers import solver as S
log = []
for name in ("round", "replay_round"):
    orig = getattr(S.ILQRSolver, name)
    def make(orig):
        def f(self, *a, **k):
            fr, ac = int(self.fresh.sum().item()), int(self.active.sum().item())
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            r = orig(self, *a, **k)
            torch.cuda.synchronize()
            log.append((fr, ac, round((time.perf_counter() - t0) * 1e3, 2)))
            return r
        return f
    setattr(S.ILQRSolver, name, make(orig))
sys.argv = ["bench.py", "--workload", "mpc_bnn", "--no-cpu-baseline", "--steps", "12"]
bench.main()
fr = [a for a, b, c in log]
print("rounds:", len(log), " mean fresh %.1f of 256, mean active %.1f" % (
    sum(fr) / len(fr), sum(b for a, b, c in log) / len(log)))
hist = collections.Counter(min(a // 32, 8) for a in fr)
print("histogram of fresh / 32:", sorted(hist.items()))
print("a stretch (fresh, active, ms):", log[40:80])
full = [c for a, b, c in log if a > 128]; rest = [c for a, b, c in log if a == 0]
print("full rounds: %d, mean %.2f ms; rounds without a new nominal: %d, mean %.2f ms" % (len(full), sum(full) / max(len(full), 1), len(rest), sum(rest) / max(len(rest), 1)))
