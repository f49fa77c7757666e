#!/bin/bash
# Reproducer of the miscompiled lane kernel (round 3: "refused lane instantiation"): builds the SysROCKETLANDING translation unit of the
# library several ways and compares the lane kernel's first iterates with the wavefront kernel's (tools/dev/wprobe.py).
#   step 1 (this container, no GPU):   tools/dev/repro/lane_rocket_misched/run.sh build
#   step 2 (GPU box, via gpurun):      tools/dev/repro/lane_rocket_misched/run.sh probe
# `build` needs the objects of a regular build (build/obj, python -c "import __graft_entry__ as g; g.build()").
ROOT=$(cd "$(dirname "$0")/../../../.." && pwd); cd $ROOT
VARS="bad:-DMYR_LANE_ONE_SCHED_REGION
bad_O2:-DMYR_LANE_ONE_SCHED_REGION -O2
ok_region_split:
ok_O1:-DMYR_LANE_ONE_SCHED_REGION -O1
ok_no_unroll:-DMYR_LANE_ONE_SCHED_REGION -fno-unroll-loops
ok_no_misched:-DMYR_LANE_ONE_SCHED_REGION -mllvm -enable-misched=0
bad_no_post_misched:-DMYR_LANE_ONE_SCHED_REGION -mllvm -enable-post-misched=0
ok_no_macro_fusion:-DMYR_LANE_ONE_SCHED_REGION -mllvm -misched-fusion=false
ok_max_memory_clause:-DMYR_LANE_ONE_SCHED_REGION -mllvm -amdgpu-sched-strategy=max-memory-clause
bad_max_ilp:-DMYR_LANE_ONE_SCHED_REGION -mllvm -amdgpu-sched-strategy=max-ilp
ok_no_coalescing:-DMYR_LANE_ONE_SCHED_REGION -mllvm -join-liveintervals=false
bad_no_agpr_spills:-DMYR_LANE_ONE_SCHED_REGION -mllvm -amdgpu-spill-vgpr-to-agpr=0"
if [ "$1" = build ]; then
  echo "$VARS" | while IFS=: read name flags; do
    ( MYR_VARIANT_SYS=SysROCKETLANDING tools/dev/build_variant.sh "" librk_$name.so $flags > /tmp/librk_$name.log 2>&1; tail -n 1 /tmp/librk_$name.log ) &
    [ $(jobs -r | wc -l) -ge 6 ] && wait -n
  done; wait
elif [ "$1" = probe ]; then
  echo "$VARS" | while IFS=: read name flags; do
    printf "%-22s %-70s " "$name" "[$flags]"
    MYRIAD_HIP_LIB=$ROOT/variants/librk_$name.so WPROBE_MAX_ITER=0 WPROBE_VERBOSE=1 python tools/dev/wprobe.py ROCKETLANDING:HS:20:1 \
      "MYRIAD_SOLVE_MODE=wave,MYRIAD_SOLVE_MODE=lane" 2>/dev/null | grep "MODE=lane" | sed -E 's/.*(cost \[[0-9.]+\]).*/lane \1 (wavefront kernel, host twin: 2.637376)/'
  done
else
  echo "usage: $0 build|probe"
fi
