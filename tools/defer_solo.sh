#!/bin/bash
# Timing experiment for riccati_n4_defer.hpp: each working role ALONE (the
# other two return at once, every progress wait passes), i.e. what a phase of
# that role costs without any synchronisation.  Results of the sweep are
# garbage in these builds - never ship them.  Run on the GPU box:
#   bash tools/defer_solo.sh
set -e
cd "$(dirname "$0")/.."
F="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1"
for role in ${ROLES:-0 1 2}; do
  touch pddp_amd/csrc/riccati_defer.hip
  make -s -C pddp_amd/csrc FLAGS_riccati_defer="$F -DPDDP_DEFER_SOLO=$role" 2>/dev/null
  echo "== role $role alone (0 M, 1 Q, 2 Y)"
  python tools/sweep_variants_time.py --variants 25 --batch 4096 2>&1 | grep variant
done
touch pddp_amd/csrc/riccati_defer.hip
make -s -C pddp_amd/csrc 2>/dev/null
