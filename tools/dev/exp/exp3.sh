#!/bin/bash
# round 4, experiment 3: the ROUND-3 build (variants/lib_r3.so, sources of commit c77f06b): do its W = 2 failures reproduce; with the sweep called (v1) / inlined for W = 1 too (v2)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp3; O=gpurun_out/exp3
export PYTHONUNBUFFERED=1
FAIL="MOULDFUNGICIDE:HS:6:1,MOULDFUNGICIDE:HS:100:1,CANCERTREATMENT:HS:100:3,TIMBERHARVEST:HS:6:1,MOULDFUNGICIDE:HS:20:1,MOULDFUNGICIDE:HS:50:3"
for i in 1 2; do MYRIAD_HIP_LIB=$PWD/variants/lib_r3.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2,MYRIAD_SOLVE_MODE=wave1 >> $O/r3.log 2>&1; done
MYRIAD_HIP_LIB=$PWD/variants/lib_r3_v1.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2 >> $O/r3_callw.log 2>&1
MYRIAD_HIP_LIB=$PWD/variants/lib_r3_v2.so WPROBE_VERBOSE=1 timeout 600 python tools/dev/wprobe.py $FAIL MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2 >> $O/r3_inline.log 2>&1
MYRIAD_HIP_LIB=$PWD/variants/lib_r3.so timeout 1200 python tools/dev/wprobe.py all MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2 > $O/r3_all.log 2>&1
grep -c DIFF $O/*.log
