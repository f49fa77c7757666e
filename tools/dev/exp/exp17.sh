#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp17; O=gpurun_out/exp17
export PYTHONUNBUFFERED=1
MYRIAD_HIP_LIB=$PWD/variants/lib_both.so timeout 300 python tools/dev/fresh_stats.py CANCERTREATMENT TRAP 6 1 12 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" > $O/both.log 2>&1
grep -h "distinct\|   x" $O/both.log | cut -c1-200
for i in 1 2 3 4 5 6; do MYRIAD_HIP_LIB=$PWD/variants/lib_bt.so WPROBE_MAX_ITER=40 timeout 120 python tools/dev/fresh_stats.py CANCERTREATMENT TRAP 6 1 1 "MYRIAD_FUSED_WAVES=2" > $O/trace$i.log 2>&1; grep "x1:" $O/trace$i.log | cut -c1-160; done
MYRIAD_HIP_LIB=$PWD/variants/lib_bt.so WPROBE_MAX_ITER=40 timeout 120 python tools/dev/fresh_stats.py CANCERTREATMENT TRAP 6 1 1 "MYRIAD_FUSED_WAVES=1" > $O/trace_w1.log 2>&1; grep "x1:" $O/trace_w1.log | cut -c1-160
