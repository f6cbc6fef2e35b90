#!/bin/bash
# SQ counters of the float64 network kernel (csrc/bnn_mlp_f64.hip):
#   gpurun -- 'bash tools/pmc_mlp64.sh <tag>'
set -u
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
rocprofv3 -L > $R/gpurun_out/${TAG}_counters_list.txt 2>&1
for PASS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  N=$(echo $PASS | md5sum | cut -c1-6)
  timeout 300 rocprofv3 --kernel-trace --pmc $PASS --output-format csv \
      -d $R/gpurun_out/${TAG}_pmc_$N -- python3 $R/tools/bnn_mlp_bench.py --dtype f64 --reps 3 > $R/gpurun_out/${TAG}_pmc_$N.log 2>&1
done
cd $R
python3 - "$TAG" <<'PY'
import collections, csv, glob, os, sys
tag = sys.argv[1]
for f in glob.glob("gpurun_out/%s_pmc_*/*/*counter_collection.csv" % tag):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "mlp_f64" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k[:60], {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
