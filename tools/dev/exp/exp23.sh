#!/bin/bash
# exp23: the ROCKETLANDING Hermite-Simpson lane kernel under different compiler settings (which pass makes it wrong?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp23
for l in "" ${RK_LIBS:-O1 O2 nounroll nopromote noslp nosched}; do
  echo "=== lib ${l:-default}"
  if [ -n "$l" ]; then export MYRIAD_HIP_LIB=$GRAFT_REPO_ROOT/variants/librk_$l.so; else unset MYRIAD_HIP_LIB; fi
  for it in ${RK_ITS:-0 1 3}; do
    WPROBE_MAX_ITER=$it WPROBE_VERBOSE=1 timeout 300 python tools/dev/wprobe.py ROCKETLANDING:HS:20:1 "MYRIAD_SOLVE_MODE=wave,MYRIAD_SOLVE_MODE=lane+MYRIAD_LANE_UNVERIFIED=1" 2>&1 | grep -v "^compared" | sed "s/^/it$it /"
  done
done 2>&1 | tee gpurun_out/exp23/out.txt
