"""How often is a step of the sweep clamped (K == 0), per round of the bench's
fit loop - per trajectory-step and per (wavefront of 4 trajectories, step)?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
B = 4096
s, z0, U, _ = bench.make_cartpole_solver(B, 100, torch.float32, "cuda", 0, 0)
s.set_nominal(z0, U)
for r in range(36):
    s.round()
    k, K = s.gain_views()
    act = s.active.bool()
    z = (K == 0).all(-1).all(-1)          # [B, N] clamped steps
    z = z & act[:, None]
    w = z.view(B // 4, 4, -1).any(1)      # any of the 4 trajectories of a wave
    w16 = z.view(B // 16, 16, -1).any(1)
    kb = (k.abs().squeeze(-1) >= 9.99) & act[:, None]
    print("round %2d live %4d mu med %.1e  clamped traj-steps %.3f  wave4-steps %.3f  wg16-steps %.3f |k|>=9.99: %.3f  accepted %d" % (
        r, int(act.sum()), float(s.mu[act].median()) if act.any() else 0, float(z.float().mean()), float(w.float().mean()),
        float(w16.float().mean()), float(kb.float().mean()), int(((s.state == 1) | (s.state == 5)).sum())))
