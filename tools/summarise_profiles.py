#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>_* (tools/collect_profiles.sh) into profiles/."""
import collections
import csv
import glob
import json
import os
import sys

def newest(pattern):
    """gpurun merges new output next to older runs' files: take the latest."""
    return max(glob.glob(pattern), key=os.path.getmtime)


tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
out = "profiles"
os.makedirs(out, exist_ok=True)

def stats_and_traffic(suffix, workload, extra_args):
    """kernel-trace stats + the FETCH / WRITE passes of one bench command."""
    try:
        src = newest("gpurun_out/%s_stats%s/*/*kernel_stats.csv" % (tag, suffix))
    except ValueError:
        return
    name = "%s%s" % (tag, suffix)
    with open(os.path.join(out, "%s_kernel_stats.csv" % name), "w") as f:
        w = csv.writer(f)
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage",
                "MinNs", "MaxNs", "StdDev"]
        w.writerow(cols)
        for r in csv.DictReader(open(src)):
            w.writerow([r["Name"][:110]] + [r[c] for c in cols[1:]])
    traffic = {}
    for kind, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = newest("gpurun_out/%s_pmc_%s%s/*/*counter_collection.csv"
                   % (tag, kind, suffix))
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "pddp" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(
                    float(r["Counter_Value"]))
        for k, v in acc.items():
            traffic.setdefault(k, {})[ctr + "_KB_avg"] = sum(v) / len(v)
            traffic[k][ctr + "_launches"] = len(v)
    for k, v in traffic.items():
        fe = v.get("FETCH_SIZE_KB_avg", 0.0) * 1024
        wr = v.get("WRITE_SIZE_KB_avg", 0.0) * 1024
        # gfx950: FETCH_SIZE counts 1/2 of a wide (16 B/lane) streaming read
        v["hbm_bytes_per_launch"] = 2 * fe + wr
    json.dump({"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | "
                          "WRITE_SIZE (separate passes) -- python3 bench.py "
                          "--steps 10 --warmup 2 --no-cpu-baseline" + extra_args,
               "formula": "2*FETCH_SIZE + WRITE_SIZE, KB = 1024 B "
                          "(MI355X_MICROARCH.md, HBM section)",
               "workload": workload, "kernels": traffic},
              open(os.path.join(out, "%s_pmc_traffic.json" % name), "w"),
              indent=1)


stats_and_traffic("", "cartpole n=4 m=1 N=100 B=4096 fp32, bounds +-10", "")
stats_and_traffic("_B16384", "cartpole n=4 m=1 N=100 B=16384 fp32, bounds +-10",
                  " --batch 16384")
stats_and_traffic("_dcbnn", "double cartpole BNN (configs[3] shard): n=27 m=1 "
                  "N=150 B=1024 fp32, [200,200] x 100 particles",
                  " --workload double_cartpole_bnn --steps 2 --warmup 1")

stats_and_traffic("_dcgp", "double cartpole GP (configs[3] as stated): n=27 m=1 "
                  "N=150 B=1024 fp32, 60 training points",
                  " --workload double_cartpole_gp")

stats_and_traffic("_cpbnn", "cartpole BNN (configs[2]): n=14 m=1 N=100 B=4096 "
                  "fp32, [200,200] x 100 particles",
                  " --workload cartpole_bnn --steps 1 --warmup 1")

stats_and_traffic("_mpc", "MPC loop (configs[4]): cartpole BNN, 256 restarts, "
                  "horizon 50, 20 control steps",
                  " --workload mpc_bnn --steps 20")

stats_and_traffic("_cpbnn_f64", "cartpole BNN (configs[2]) in float64: n=14 m=1 "
                  "N=100 B=4096, [200,200] x 100 particles",
                  " --workload cartpole_bnn --dtype f64 --steps 1 --warmup 1")

f = newest("gpurun_out/%s_pmc_sq/*/*counter_collection.csv" % tag)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "pddp" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][
            r["Counter_Name"]].append(float(r["Counter_Value"]))
sq = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
json.dump({"note": "averages per launch; SQ_WAVE_CYCLES / SQ_WAIT_* / "
                   "SQ_ACTIVE_* count quad-cycles", "kernels": sq},
          open(os.path.join(out, "%s_pmc_sq.json" % tag), "w"), indent=1)

points = {}
for p in sorted(glob.glob("gpurun_out/%s_bench*.json" % tag)):
    try:
        line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
        points[os.path.basename(p)[len(tag) + 1:-5]] = json.loads(line)
    except (IndexError, ValueError):
        pass
json.dump(points, open(os.path.join(out, "%s_bench_points.json" % tag), "w"),
          indent=1)
for k, d in points.items():
    if not d.get("roofline"):
        continue
    print(k, round(d["ms_per_step"], 4), "ms/step; sweep",
          round(d["roofline"]["avg_launch_us"], 1), "us, frac",
          round(d["roofline"]["frac"], 4), "value", int(d["value"]))
