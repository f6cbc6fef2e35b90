"""Error of the fp32 sweep kernels against the fp64 kernel ON THE SAME RECORDS
(the bench workload's records after a few fit rounds, cast up): per-trajectory
relative error of the gains, distribution per variant.
    python tools/sweep_accuracy.py --variants 7,17 --rounds 4"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="7,17")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--rounds", default="0,4,12")
    a = ap.parse_args()
    B, N = a.batch, 100
    s, z0, U, _ = bench.make_cartpole_solver(B, N, torch.float32, "cuda", 0, 0)
    d, _, _, _ = bench.make_cartpole_solver(B, N, torch.float64, "cuda", 0, 0)
    s.set_nominal(z0, U)
    s.derivs(mask=s.fresh)
    done = 0
    for target in [int(x) for x in a.rounds.split(",")]:
        while done < target:
            s.round(5e-6, 1e10, 1 << 30)
            done += 1
        d.rec.copy_(s.rec.double())
        d.mu.copy_(s.mu)
        reg = s.mu.clone()
        d.backward(active=None, reg=reg, variant=16)
        ref = d.gains.clone()
        ok64 = d.bwd_status == 0
        scale = ref.abs().amax(dim=(1, 2)).clamp_min(1e-30)
        for v in [int(x) for x in a.variants.split(",")]:
            s.gains.zero_()
            s.backward(active=None, reg=reg, variant=v)
            ok = ok64 & (s.bwd_status == 0)
            e = ((s.gains.double() - ref).abs().amax(dim=(1, 2)) / scale)[ok]
            e = e.cpu().numpy()
            if e.size == 0:
                print("round %d variant %d: nothing to compare" % (done, v))
                continue
            print("round %2d variant %2d: n %d  median %.2e  p99 %.2e  max %.2e  "
                  "> 1e-2: %d  status flips %d" %
                  (done, v, e.size, np.median(e), np.percentile(e, 99), e.max(),
                   int((e > 1e-2).sum()),
                   int(((s.bwd_status == 0) != ok64).sum())), flush=True)


if __name__ == "__main__":
    main()
