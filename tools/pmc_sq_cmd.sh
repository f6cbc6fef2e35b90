#!/bin/bash
# SQ instruction / stall counters of any python command:
#   gpurun -- 'bash tools/pmc_sq_cmd.sh <tag> tools/sweep_ablation.py --variants 13'
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
SCRIPT=$R/$1; shift
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -- python3 $SCRIPT "$@" > /dev/null 2>&1
cd $R
python3 - "$TAG" <<'PY'
import collections, csv, glob, json, os, sys
tag = sys.argv[1]
f = max(glob.glob("gpurun_out/%s_pmc_sq/*/*counter_collection.csv" % tag), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "pddp" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = max(m.get("SQ_WAVES", 1), 1)
    print(k[:70], "launches %d waves %d" % (len(d["SQ_WAVES"]), w), "per wave: VALU %.0f SALU %.0f LDS %.0f | wave-cycles(x4) %.0f active %.0f%% wait_any %.0f%% wait_inst %.0f%%" % (
        m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_LDS", 0) / w,
        4 * m.get("SQ_WAVE_CYCLES", 0) / w, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1),
        100 * m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), 100 * m.get("SQ_WAIT_INST_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
