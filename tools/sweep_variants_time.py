"""Times the backward sweep alone (events on the dispatch) for several kernel
variants on the bench workload's records, after a few fit rounds:
    python tools/sweep_variants_time.py --variants 7,17 --batch 4096
Prints microseconds (median / min) and the HBM-roofline fraction."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="7,17")
    ap.add_argument("--batch", default="4096")
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    lib = _native.lib()
    for B in [int(x) for x in a.batch.split(",")]:
        s, z0, U, _ = bench.make_cartpole_solver(B, a.horizon, dt, "cuda", 0, 0)
        s.set_nominal(z0, U)
        for _ in range(4):
            s.round(5e-6, 1e10, 1 << 30)
        nbytes = B * bench.algorithmic_bytes_per_trajectory(
            a.horizon, 4, 1, s.rec.element_size(), True)
        ref = None
        for v in [int(x) for x in a.variants.split(",")]:
            pool = bench.EventPool(lib)
            for _ in range(3):
                s.backward(active=s.active, variant=v)
            for _ in range(a.reps):
                s.backward(active=s.active, variant=v, events=pool.pair())
            torch.cuda.synchronize()
            ts = sorted(1e6 * x for x in pool.durations())
            med, mn = ts[len(ts) // 2], ts[0]
            g = s.gains.clone()
            st = int((s.bwd_status != 0).sum())
            if ref is None:
                ref = g
                dmax = 0.0
            else:
                dmax = float(((g - ref).abs().amax(dim=(1, 2)) /
                              ref.abs().amax(dim=(1, 2)).clamp_min(1e-30)).max())
            frac = (nbytes / (med * 1e-6) / 8e12) if nbytes else float("nan")
            print("B %6d variant %2d: median %7.2f us  min %7.2f us  frac %.3f  "
                  "failed %d  max rel diff vs first %.2e" %
                  (B, v, med, mn, frac, st, dmax), flush=True)


if __name__ == "__main__":
    main()
