"""How many trajectories of the GP workload (bench.py --workload
double_cartpole_gp) have a NEW nominal at the start of a round - the share of
the derivative launch's rows that a row mask would skip."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from pddp_amd.controllers import solver as S
orig = S.ILQRSolver.round
log = []
def rnd(self, *a, **k):
    log.append((int(self.fresh.sum().item()), int(self.active.sum().item())))
    return orig(self, *a, **k)
S.ILQRSolver.round = rnd
sys.argv = ["bench.py", "--workload", "double_cartpole_gp", "--no-cpu-baseline", "--no-graph-replay", "--steps", "12", "--warmup", "1"]
bench.main()
print("fresh / active per round:", log)
