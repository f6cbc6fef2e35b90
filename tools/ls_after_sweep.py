"""Times the line-search kernel right after different sweep variants (does the
sweep leave the gains where the line search finds them?)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.variant_ab import make  # noqa: E402


def main():
    s = make(4096, 100, torch.float32)
    s.derivs()
    reg = torch.full((4096,), 1.0, dtype=torch.float64, device="cuda")
    x = torch.zeros(1024 * 64, device="cuda")
    y = torch.zeros(64, device="cuda")
    for seq in ((7,), (9,), (9, "small"), (9, "wide"), (7, "small"), (7, "wide"),
                (9, "sync")):
        tot = 0.0
        for it in range(25):
            for v in seq:
                if v == "small":
                    y.add_(1.0)
                elif v == "wide":
                    x.add_(1.0)
                elif v == "sync":
                    torch.cuda.synchronize()
                else:
                    s.backward(reg=reg, variant=v)
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            s.line_search()
            e1.record()
            torch.cuda.synchronize()
            if it >= 5:
                tot += e0.elapsed_time(e1)
        print("sweep variants %s -> line search %.1f us" % (seq, tot / 20 * 1e3))


if __name__ == "__main__":
    main()
