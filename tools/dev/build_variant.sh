#!/bin/bash
# dev tool: build a variant of the library in which only the translation units named in MYR_VARIANT_SYS (comma list, default SysCARTPOLE)
# are recompiled (from the source tree given as $1, default the repo's csrc) with extra compiler flags ($3...), linked with the other
# objects of the last regular build (build/obj).  Output: variants/$2 (git-ignored, travels with gpurun snapshots).
#   MYR_VARIANT_SYS=SysCARTPOLE,SysVANDERPOL tools/dev/build_variant.sh "" libvar.so -DMYR_PHASE_TIMING
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=${1:-$ROOT/myriad_amd/csrc}; OUT=${2:-libvar.so}; shift 2 || true
mkdir -p $ROOT/variants /tmp/variant_obj
TUS=${MYR_VARIANT_SYS:-SysCARTPOLE}
OBJS=$(ls $ROOT/build/obj/*.o)
NEW=""
pids=""
for TU in ${TUS//,/ }; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form -c -DMYR_TU_SYSTEM=$TU "$@" $SRC/myriad_hip.hip -o /tmp/variant_obj/$OUT.$TU.o &
  pids="$pids $!"
  while [ $(jobs -r | wc -l) -ge ${MYR_VARIANT_JOBS:-8} ]; do wait -n; done
  OBJS=$(echo "$OBJS" | grep -v "/$TU.o")
  NEW="$NEW /tmp/variant_obj/$OUT.$TU.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $OBJS $NEW -o $ROOT/variants/$OUT 2>/dev/null
rm -f $ROOT/variants/$OUT.*.hipv4-* $ROOT/variants/$OUT.*.host-*
echo built $ROOT/variants/$OUT
