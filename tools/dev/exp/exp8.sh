#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp8; O=gpurun_out/exp8
export PYTHONUNBUFFERED=1
for v in v1 v2 v3; do
  MYRIAD_HIP_LIB=$PWD/variants/lib_$v.so timeout 600 python tools/dev/fresh_stats.py TIMBERHARVEST TRAP 6 3 16 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" > $O/$v.log 2>&1
  MYRIAD_HIP_LIB=$PWD/variants/lib_$v.so timeout 600 python tools/dev/fresh_stats.py TIMBERHARVEST HS 6 3 16 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" >> $O/$v.log 2>&1
done
timeout 600 python tools/dev/fresh_stats.py TIMBERHARVEST HS 6 3 16 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" > $O/default_hs.log 2>&1
grep -h "distinct" $O/*.log
