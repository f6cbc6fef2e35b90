import ctypes, json, subprocess, sys, torch
sys.path.insert(0, ".")
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
