"""The headline round at B = 8192 / 12288 / 16384 under the four combinations
of (nominal sweep kernel: inline 3 / overlapped 4) x (candidates kept 1 /
dropped 2): which should auto pick between 8192 and 16384?"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import json, sys
sys.path.insert(0, %r)
import torch
from pddp_amd import _native
import bench
_native.lib().pddp_sweep_nominal_kernel(int(sys.argv[2]))
_native.lib().pddp_search_candidates(int(sys.argv[3]))
sys.argv = ["bench.py", "--batch", sys.argv[1], "--no-cpu-baseline", "--no-points", "--no-secondary", "--repeats", "3"]
bench.main()
''' % ROOT
for B in (8192, 12288, 16384):
    for sweep in (3, 4):
        for cand in (1, 2):
            out = subprocess.run([sys.executable, "-c", code, str(B), str(sweep), str(cand)],
                                 capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(B, sweep, cand, "FAILED", out.stderr[-400:]); continue
            d = json.loads(out.stdout.strip().splitlines()[-1])
            r = d["roofline"]
            print("B %5d sweep %d cand %d: %.4f ms  %.1f M/s  sweep %.1f us  search %.1f us" % (
                B, sweep, cand, d["ms_per_step"], d["value"] / 1e6, r["avg_launch_us"],
                r["other_kernels"][0]["avg_launch_us"]), flush=True)
