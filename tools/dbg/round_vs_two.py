"""One-launch round against the two launches from the same state, round by
round (rewind): where do they part?  python tools/dbg/round_vs_two.py B N"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import test_gpu_parity as T  # noqa: E402

B, N = int(sys.argv[1]), int(sys.argv[2])
s, op, z0, U, u_min, u_max = T._setup("cartpole", "f32", B, N, seed=5)
s.set_nominal(torch.from_numpy(z0).cuda(), torch.from_numpy(U).cuda())
names = ("Z", "U", "L", "J_opt", "mu", "delta", "state", "iter", "active",
         "fresh", "gains", "gains_acc", "Jc", "bwd_status", "n_live", "Zc", "Uc")
for r in range(10):
    pre = {k: getattr(s, k).clone() for k in names}
    s._one_launch = None
    s.round(n_iterations=10)
    one = {k: getattr(s, k).clone() for k in names}
    for k in names:
        getattr(s, k).copy_(pre[k])
    s._one_launch = False
    s.round(n_iterations=10)
    two = {k: getattr(s, k).clone() for k in names}
    dz = (one["Z"] - two["Z"]).abs().reshape(B, -1).amax(1)
    am1 = torch.nan_to_num(one["Jc"], nan=-1e30).argmin(1)
    am2 = torch.nan_to_num(two["Jc"], nan=-1e30).argmin(1)
    acc = (two["state"] == 1) | (two["state"] == 5)
    bad = (dz > 1e-3).nonzero().flatten().tolist()
    print("round %d: accepted %d, argmin differs %d, |dZ| > 1e-3 in %s" % (
        r, int(acc.sum()), int((am1 != am2).sum()), bad[:8]))
    for b in bad[:3]:
        t_bad = ((one["Z"][b] - two["Z"][b]).abs().amax(1) > 1e-3).nonzero().flatten().tolist()
        print("   b %d state %d amin %d/%d J %s | %s  rows differing: %s" % (
            b, int(two["state"][b]), int(am1[b]), int(am2[b]),
            one["Jc"][b, :4].tolist(), two["Jc"][b, :4].tolist(), t_bad[:12]))
        print("      row 0: pre %s\n             one %s\n             two %s" % (
            pre["Z"][b, 0].tolist(), one["Z"][b, 0].tolist(), two["Z"][b, 0].tolist()))
        print("      row 5: pre %s\n             one %s\n             two %s" % (
            pre["Z"][b, 5].tolist(), one["Z"][b, 5].tolist(), two["Z"][b, 5].tolist()))
        print("      U[:4]: pre %s one %s two %s" % (pre["U"][b, :4, 0].tolist(), one["U"][b, :4, 0].tolist(), two["U"][b, :4, 0].tolist()))
        # is either the candidate row?
        a_ = int(am2[b])
        zc2 = two["Zc"][b, :, a_, :] if a_ else None
        if zc2 is not None:
            print("      two == its Zc[amin]: %s; one == two's Zc[amin]: %s; one == one's Zc: %s" % (
                float((two["Z"][b] - zc2).abs().max()),
                float((one["Z"][b] - zc2).abs().max()),
                float((one["Z"][b] - one["Zc"][b, :, int(am1[b]), :]).abs().max())))
    for k in names:
        getattr(s, k).copy_(one[k])
