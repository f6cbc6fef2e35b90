"""bench.py's headline round under the paired (1) and the dense (2) form of
pddp_search_accept_f32 over a range of batches."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import json, sys
sys.path.insert(0, %r)
from pddp_amd import _native
import bench
_native.lib().pddp_search_form(int(sys.argv[2]))
sys.argv = ["bench.py", "--batch", sys.argv[1], "--no-cpu-baseline", "--no-points", "--no-secondary", "--repeats", "3"]
bench.main()
''' % ROOT
for B in [int(v) for v in sys.argv[1:]] or (4096, 8192, 12288, 16384, 32768, 65536):
    for form in (1, 2):
        out = subprocess.run([sys.executable, "-c", code, str(B), str(form)], capture_output=True, text=True, timeout=600)
        if out.returncode != 0:
            print(B, form, "FAILED", out.stderr[-400:]); continue
        d = json.loads(out.stdout.strip().splitlines()[-1])
        r = d["roofline"]
        print("B %5d form %d: %.4f ms  %.1f M/s  sweep %.1f us  search %.1f us" % (
            B, form, d["ms_per_step"], d["value"] / 1e6, r["avg_launch_us"], r["other_kernels"][0]["avg_launch_us"]), flush=True)
