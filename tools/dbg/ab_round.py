"""A/B of two builds of the library on the same box: per-round time of the
one-launch round at 10 rounds per launch, alternating processes.
python tools/dbg/ab_round.py <libA> <libB> [reps]"""
import os
import subprocess
import sys

code = r"""
import os, sys, time, torch
sys.path.insert(0, %r)
import bench
s, z0, U, _ = bench.make_cartpole_solver(4096, 100, torch.float32, "cuda", 0, 0)
best = []
for rep in range(4):
    s.set_nominal(z0, U)
    for _ in range(5):
        s.round(5e-6, 1e10, 1 << 30)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        s.rounds(10, 5e-6, 1e10, 1 << 30)
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t) / 200 * 1e6)
print("%%.2f %%.2f" %% (min(best), sorted(best)[len(best) // 2]))
""" % os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
libs = sys.argv[1:3]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for r in range(reps):
    for lib in libs:
        env = dict(os.environ, PDDP_HIP_LIB=lib)
        out = subprocess.run([sys.executable, "-c", code], env=env,
                             capture_output=True, text=True)
        print(os.path.basename(os.path.dirname(lib)), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
