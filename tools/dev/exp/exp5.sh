#!/bin/bash
# experiment 5: the round-3 binary with a bit pattern left in LDS / freed device memory before every solve -- does its W = 2 result follow the pattern?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp5; O=gpurun_out/exp5
export PYTHONUNBUFFERED=1
FAIL="MOULDFUNGICIDE:HS:6:1,MOULDFUNGICIDE:HS:100:1,CANCERTREATMENT:HS:100:3,MOULDFUNGICIDE:HS:50:3"
for f in zero nan big one; do
  echo "=== fill $f, lib_r3" >> $O/fill.log
  WPROBE_FILL=$f MYRIAD_HIP_LIB=$PWD/variants/lib_r3.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_SOLVE_MODE=wave1 >> $O/fill.log 2>&1
  echo "=== fill $f, round-4 build" >> $O/fill4.log
  WPROBE_FILL=$f WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_SOLVE_MODE=wave1 >> $O/fill4.log 2>&1
done
grep -v amdgpu $O/fill.log | cut -c1-200
