#!/bin/bash
# dev tool: recompile only the named translation units of myriad_hip.hip (main, SysCARTPOLE, SysNODE_CARTPOLE, ...) into build/obj and relink
# myriad_amd/libmyriad_hip.so from the objects of the last regular build.  No listing scan -- the regular build (__graft_entry__.build) stays the gate.
#   tools/dev/rebuild_tu.sh main SysNODE_CARTPOLE [-- extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TUS=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; TUS+=("$1"); shift; done
pids=()
for TU in "${TUS[@]}"; do
  # (a system built in parts is named Sys....p1 / .p3 / .p4 / .p5 / .p6 / .p2, as __graft_entry__.build names its objects)
  if [ "$TU" = main ]; then DEF=-DMYR_TU_MAIN; elif [[ "$TU" == *.p[1-6] ]]; then DEF="-DMYR_TU_SYSTEM=${TU%.p?} -DMYR_TU_PART=${TU##*.p}"; else DEF=-DMYR_TU_SYSTEM=$TU; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form $DEF "${EXTRA[@]}" -c $ROOT/myriad_amd/csrc/myriad_hip.hip -o $ROOT/build/obj/$TU.o &
  pids+=($!)
  while [ $(jobs -r | wc -l) -ge ${MYR_JOBS:-8} ]; do wait -n; done
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $ROOT/build/obj/*.o -o $ROOT/myriad_amd/libmyriad_hip.so
echo "relinked $ROOT/myriad_amd/libmyriad_hip.so (${TUS[*]})"
