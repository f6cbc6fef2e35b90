"""bench.py's headline workload over a range of batches: trajectory-iterations
per second, the sweep's and the search launch's durations (the cliff above
B = 8192 of round 3):  python tools/batch_scan.py [B ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for B in [int(v) for v in sys.argv[1:]] or [4096, 8192, 12288, 16384, 32768,
                                            65536]:
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--batch", str(B),
         "--no-cpu-baseline", "--no-points", "--no-secondary", "--repeats",
         "3"], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(B, "FAILED", out.stderr[-500:])
        continue
    d = json.loads(out.stdout.strip().splitlines()[-1])
    r = d["roofline"]
    row = dict(B=B, value=d["value"], ms_per_step=d["ms_per_step"],
               sweep_us=r["avg_launch_us"], sweep_frac=r["frac"],
               search_us=r["other_kernels"][0]["avg_launch_us"],
               search_frac=r["other_kernels"][0]["frac"])
    rows.append(row)
    print(json.dumps(row), flush=True)
