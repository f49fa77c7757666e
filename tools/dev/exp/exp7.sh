#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp7; O=gpurun_out/exp7
export PYTHONUNBUFFERED=1
E="MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=nan,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=big,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=random,MYRIAD_FUSED_WAVES=1+MYRIAD_POISON=random"
timeout 600 python tools/dev/fresh_stats.py TIMBERHARVEST TRAP 6 3 12 "$E" > $O/default.log 2>&1
timeout 600 python tools/dev/fresh_stats.py TIMBERHARVEST TRAP 6 1 12 "$E" >> $O/default.log 2>&1
for v in strong callw; do
  MYRIAD_HIP_LIB=$PWD/variants/lib_$v.so timeout 600 python tools/dev/fresh_stats.py TIMBERHARVEST TRAP 6 3 12 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_FUSED_WAVES=2+MYRIAD_POISON=random" > $O/$v.log 2>&1
done
for i in 1 2 3 4; do MYRIAD_FUSED_WAVES=2 MYRIAD_HIP_LIB=$PWD/variants/lib_trace.so WPROBE_MAX_ITER=300 timeout 300 python tools/dev/fresh_stats.py TIMBERHARVEST TRAP 6 3 1 "MYRIAD_FUSED_WAVES=2" > $O/trace$i.log 2>&1; done
cat $O/default.log $O/strong.log $O/callw.log | grep -v amdgpu
