"""Per-launch duration of the fused line search + accept + records against
the share of trajectories it accepted, in the bench's own loop (configs[1]):
    python tools/ls_tail_profile.py [--batch 4096] [--rounds 35]
One line per round: accepted share, sweep us, search us."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=35)
    args = ap.parse_args()
    from pddp_amd import _native
    device = torch.device("cuda:0")
    s, z0, U, _ = bench.make_cartpole_solver(args.batch, 100, torch.float32,
                                             device, 0, 0)
    lib = _native.lib()
    s.set_nominal(z0, U)
    ps, pl = bench.EventPool(lib), bench.EventPool(lib)
    acc = []
    hist = torch.zeros(16, dtype=torch.int64, device=device)
    pr = bench.EventPool(lib)  # the rollouts alone (the separate line search)
    for _ in range(args.rounds):
        if s._nominal_sweep is not False and s.sweep_nominal(events=ps.pair()):
            lib.pddp_attach_events(*pr.pair())
            s.line_search(active=s.active)
            s.search_accept(5e-6, 1e10, 1 << 30, events=pl.pair(),
                            records=False)
        else:
            s.round(5e-6, 1e10, 1 << 30, backward_events=ps.pair(),
                    search_events=pl.pair())
        ok = (s.state == 1) | (s.state == 5)
        acc.append(ok.sum())
        hist += torch.bincount(s.Jc.argmin(1)[ok], minlength=16)[:16]
    torch.cuda.synchronize(device)
    print("winning step size of the accepted attempts (index: count):",
          {i: int(c) for i, c in enumerate(hist.tolist()) if c})
    ds, dl = np.array(ps.durations()) * 1e6, np.array(pl.durations()) * 1e6
    dr = np.array(pr.durations()) * 1e6 if pr.pairs else np.zeros(len(dl))
    acc = [int(a.item()) / args.batch for a in acc]
    for i, (a, x, y, r) in enumerate(zip(acc, ds, dl, dr)):
        print("round %2d  accepted %.3f  sweep %6.1f us  search %6.1f us  "
              "(rollouts alone %6.1f us)" % (i, a, x, y, r))
    # least squares: search = c0 + c1 * share
    A = np.stack([np.ones(len(acc) - 5), np.array(acc[5:])], 1)
    c = np.linalg.lstsq(A, dl[5:], rcond=None)[0]
    print("fit (rounds 5..): search = %.1f + %.1f * share us" % (c[0], c[1]))


if __name__ == "__main__":
    main()
