#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp9; O=gpurun_out/exp9
export PYTHONUNBUFFERED=1
for v in a b c d; do
  echo "=== variant $v" > $O/$v.log
  for case in "TIMBERHARVEST TRAP 6 3" "CANCERTREATMENT HS 50 1" "MOULDFUNGICIDE HS 100 3" "BIOREACTOR HS 20 3" "CANCERTREATMENT HS 100 3" "MOULDFUNGICIDE HS 6 1"; do
    MYRIAD_HIP_LIB=$PWD/variants/lib_$v.so timeout 300 python tools/dev/fresh_stats.py $case 12 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" >> $O/$v.log 2>&1
  done
  grep -h "variant\|distinct\|fault" $O/$v.log
done
