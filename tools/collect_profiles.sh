#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# Writes raw output under gpurun_out/<tag>_* ; tools/summarise_profiles.py then
# condenses it into profiles/ (tracked).
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
# per-kernel time (kernel trace + stats only)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- \
    python3 $R/bench.py --steps 30 --warmup 5 > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_stats.err
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots)
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- \
    python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- \
    python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
# instruction mix / stall counters of the sweep
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -- \
    python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
# un-profiled runs: the headline line and the scaling points
python3 bench.py --steps 30 --warmup 5 > gpurun_out/${TAG}_bench.json 2>/dev/null
for B in 1024 8192 16384; do
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch $B 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_B$B.json
done
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --dtype f64 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_f64.json
for v in 1 2 3 6 7 9; do
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel-variant $v 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_variant$v.json
done
tail -c 600 gpurun_out/${TAG}_bench.json
