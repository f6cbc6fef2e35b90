#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun):
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# Writes raw output under gpurun_out/<tag>_* ; tools/summarise_profiles.py then
# condenses it into profiles/ (tracked).
set -u
TAG=${1:-r05}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "${2:-}" != "bnn" ]; then
cd /tmp
# per-kernel time (kernel trace + stats only)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- \
    python3 $R/bench.py --steps 30 --warmup 5 --repeats 1 --no-points --no-secondary > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> $R/gpurun_out/${TAG}_stats.err
# HBM traffic: FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots)
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -- \
    python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-points --no-secondary --no-cpu-baseline > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -- \
    python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-points --no-secondary --no-cpu-baseline > /dev/null 2>&1
# instruction mix / stall counters of the sweep
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq -- \
    python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-points --no-secondary --no-cpu-baseline > /dev/null 2>&1
cd $R
# un-profiled runs: the headline line and the scaling points
python3 bench.py --steps 30 --warmup 5 > gpurun_out/${TAG}_bench.json 2>/dev/null
for B in 1024 8192 12288 16384 65536; do
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-points --no-secondary --batch $B 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_B$B.json
done
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-points --no-secondary --dtype f64 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_f64.json
for v in 1 7 16 17; do
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-points --no-secondary --kernel-variant $v 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_variant$v.json
done
# the B = 16384 sweep (quad kernel) under the counters
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_B16384 -- \
    python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-points --no-secondary --no-cpu-baseline --batch 16384 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_B16384 -- \
    python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-points --no-secondary --no-cpu-baseline --batch 16384 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_B16384 -- \
    python3 $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-points --no-secondary --no-cpu-baseline --batch 16384 > /dev/null 2>&1
cd $R
# where a round goes, workgroup by workgroup (a -DPDDP_WG_TIMELINE build:
# tools/build_variant.sh tl -DPDDP_WG_TIMELINE)
if [ -f pddp_amd/lib_tl/libpddp_hip.so ]; then
  PDDP_HIP_LIB=pddp_amd/lib_tl/libpddp_hip.so python3 tools/wg_timeline.py 4096 --two 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_wg_timeline_two_launches.txt
  PDDP_HIP_LIB=pddp_amd/lib_tl/libpddp_hip.so python3 tools/wg_timeline.py 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_wg_timeline_one_launch.txt
  PDDP_HIP_LIB=pddp_amd/lib_tl/libpddp_hip.so python3 tools/wg_timeline.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_wg_timeline_B64.txt
  PDDP_HIP_LIB=pddp_amd/lib_tl/libpddp_hip.so python3 tools/wg_timeline.py 512 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_wg_timeline_B512.txt
fi
python3 tools/wg_timeline.py 4096 2>&1 | grep "mean of" > gpurun_out/${TAG}_rounds_per_launch.txt
tail -c 600 gpurun_out/${TAG}_bench.json
fi
cd $R
# configs[2] on the float64 kernels under the profiler
if [ "${2:-}" = "bnn64" ]; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_cpbnn_f64 -- \
      python3 $R/bench.py --workload cartpole_bnn --dtype f64 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_cpbnn_f64 -- \
      python3 $R/bench.py --workload cartpole_bnn --dtype f64 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_cpbnn_f64 -- \
      python3 $R/bench.py --workload cartpole_bnn --dtype f64 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R
fi
# the BNN workloads (configs[2], configs[3]'s shard, configs[4]); bench lines only
if [ "${2:-}" = "bnn" ]; then
  python3 bench.py --workload cartpole_bnn --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_cartpole_bnn.json
  python3 bench.py --workload double_cartpole_bnn --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_double_cartpole_bnn.json
  python3 bench.py --workload mpc_bnn --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_mpc_bnn.json
  # configs[3] as stated (GP plugin on pddp_gp_step): the line, then per-kernel time
  python3 bench.py --workload double_cartpole_gp 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_double_cartpole_gp.json
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_dcgp -- \
      python3 $R/bench.py --workload double_cartpole_gp --no-cpu-baseline --no-graph-replay > /dev/null 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_dcgp -- \
      python3 $R/bench.py --workload double_cartpole_gp --no-cpu-baseline --no-graph-replay > /dev/null 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_dcgp -- \
      python3 $R/bench.py --workload double_cartpole_gp --no-cpu-baseline --no-graph-replay > /dev/null 2>&1 )
  # configs[4]'s own launches under the counters (20 control steps)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_mpc -- \
      python3 $R/bench.py --workload mpc_bnn --steps 20 --no-cpu-baseline > /dev/null 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_mpc -- \
      python3 $R/bench.py --workload mpc_bnn --steps 20 --no-cpu-baseline > /dev/null 2>&1 )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_mpc -- \
      python3 $R/bench.py --workload mpc_bnn --steps 20 --no-cpu-baseline > /dev/null 2>&1 )
  tail -c 400 gpurun_out/${TAG}_bench_mpc_bnn.json
  # configs[3]'s shard under the profiler: per-kernel time and HBM traffic of
  # the n = 27 sweep (riccati_mfma32_kernel) and the BNN kernels
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_dcbnn -- \
      python3 $R/bench.py --workload double_cartpole_bnn --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_dcbnn -- \
      python3 $R/bench.py --workload double_cartpole_bnn --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_dcbnn -- \
      python3 $R/bench.py --workload double_cartpole_bnn --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  # configs[2] under the counters: the n = 14 sweep and the network kernel
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_cpbnn -- \
      python3 $R/bench.py --workload cartpole_bnn --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch_cpbnn -- \
      python3 $R/bench.py --workload cartpole_bnn --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write_cpbnn -- \
      python3 $R/bench.py --workload cartpole_bnn --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  cd $R
fi
