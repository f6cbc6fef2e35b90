"""Are the line search's candidate stores free at B = 4096?  (VERDICT round 4,
weak 7: re-measure on the current kernel form.)  The search + accept launch
with its candidates kept (93 MB written per launch) and dropped
(pddp_search_candidates(2): the stores stay in the instruction stream with a
stride of zero - one scratch row that lives in L2), timed by events on the
dispatch, in rounds that reject everything (no second rollout of a winner) and
in rounds that accept."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
from pddp_amd import _native  # noqa: E402

lib = _native.lib()
s, z0, U, _ = bench.make_cartpole_solver(4096, 100, torch.float32, "cuda", 0, 0)
s._one_launch = False
for mode, name in ((1, "kept"), (2, "dropped"), (1, "kept"), (2, "dropped")):
    lib.pddp_search_candidates(mode)
    s.set_nominal(z0, U)
    pool = bench.EventPool(lib)
    acc = []
    for r in range(30):
        s.round(5e-6, 1e10, 1 << 30, search_events=pool.pair())
        acc.append(((s.state == 1) | (s.state == 5)).sum())
    torch.cuda.synchronize()
    d = np.array(pool.durations()) * 1e6
    a = np.array([int(v) for v in acc])
    rej, ok = a == 0, a > 2000
    print("%-8s rounds that reject all: %5.1f us (%d)   rounds that accept "
          "> half: %5.1f us (%d)   all: %5.1f" % (
              name, d[rej].mean() if rej.any() else float("nan"), rej.sum(),
              d[ok].mean() if ok.any() else float("nan"), ok.sum(), d.mean()))
lib.pddp_search_candidates(0)
