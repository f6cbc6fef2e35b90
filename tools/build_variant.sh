#!/bin/bash
# An instrumented / experimental build of the library next to the product's:
#   tools/build_variant.sh <name> <extra hipcc flags...>
# -> pddp_amd/lib_<name>/libpddp_hip.so (git-ignored, travels with gpurun);
# select it with PDDP_HIP_LIB=pddp_amd/lib_<name>/libpddp_hip.so
set -e
name=$1; shift
here=$(cd "$(dirname "$0")/.." && pwd)
base="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -fno-fast-math -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical"
make -s -j6 -C "$here/pddp_amd/csrc" OUT=../lib_$name OBJ=../lib_$name/obj HIPFLAGS="$base $*"
ls -la "$here/pddp_amd/lib_$name/libpddp_hip.so"
