"""How often the closed-form BoxQP of the n=4 sweep falls back to the loop.
Needs a library built with the counters:

    make -C pddp_amd/csrc HIPFLAGS="... -DPDDP_QP_STATS"   (see Makefile for the flags)
    python tools/qp_stats.py

Round 1, bench workload (B=4096, N=100, 40 iterations): 700 of 4 096 000
wave-steps (0.017 %).  The counters cost ~1 ms per sweep: never ship this build.
"""
import ctypes, json, subprocess, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "40", "--warmup", "0", "--no-cpu-baseline"]
from pddp_amd import _native
lib = _native.lib()
out = (ctypes.c_ulonglong * 4)()
import bench
lib.pddp_debug_qp_stats(out, 1)
bench.main()
lib.pddp_debug_qp_stats(out, 1)
print("wave-steps slow/total", out[0], out[1], out[0] / max(out[1], 1),
      "traj-steps slow/total", out[2], out[3], out[2] / max(out[3], 1))
