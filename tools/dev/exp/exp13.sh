#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp13; O=gpurun_out/exp13
export PYTHONUNBUFFERED=1 MYRIAD_LANE_UNVERIFIED=1
for lib in variants/lib_ws.so myriad_amd/libmyriad_hip.so; do
  echo "== $lib" >> $O/rocket.log
  for lim in 0 1 2 8 40; do
    WPROBE_VERBOSE=1 WPROBE_MAX_ITER=$lim MYRIAD_HIP_LIB=$PWD/$lib timeout 300 python tools/dev/wprobe.py ROCKETLANDING:HS:20:1 "MYRIAD_SOLVE_MODE=wave,MYRIAD_SOLVE_MODE=lane" 2>&1 | grep -v amdgpu | cut -c1-200 >> $O/rocket.log
  done
done
cat $O/rocket.log
