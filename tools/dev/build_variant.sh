#!/bin/bash
# dev tool: build a variant of the library in which only ONE translation unit (MYR_VARIANT_SYS, default SysCARTPOLE) is recompiled (from the source
# tree given as $1, default the repo's csrc) with extra compiler flags ($3...), linked with the other objects of the
# last regular build (build/obj).  Output: variants/$2 (git-ignored, travels with gpurun snapshots).
#   tools/dev/build_variant.sh /tmp/p3/x/y/csrc libvar.so -DMYR_PHASE_TIMING
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=${1:-$ROOT/myriad_amd/csrc}; OUT=${2:-libvar.so}; shift 2 || true
mkdir -p $ROOT/variants /tmp/variant_obj
TU=${MYR_VARIANT_SYS:-SysCARTPOLE}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -c -DMYR_TU_SYSTEM=$TU "$@" $SRC/myriad_hip.hip -o /tmp/variant_obj/$OUT.$TU.o
OBJS=$(ls $ROOT/build/obj/*.o | grep -v $TU.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $OBJS /tmp/variant_obj/$OUT.$TU.o -o $ROOT/variants/$OUT
echo built $ROOT/variants/$OUT
