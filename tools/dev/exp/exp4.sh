#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp4; O=gpurun_out/exp4
export PYTHONUNBUFFERED=1
for W in 2 1; do
  MYRIAD_FUSED_WAVES=$W MYRIAD_HIP_LIB=$PWD/variants/lib_r3.so timeout 600 python tools/dev/race_stats.py MOULDFUNGICIDE HS 6 1 40 0,1,2,3,4,6,8,300 > $O/r3_mould6_w$W.log 2>&1
  MYRIAD_FUSED_WAVES=$W MYRIAD_HIP_LIB=$PWD/variants/lib_r3.so timeout 600 python tools/dev/race_stats.py MOULDFUNGICIDE HS 100 1 40 0,1,2,3,4,6,8,300 > $O/r3_mould100_w$W.log 2>&1
  MYRIAD_FUSED_WAVES=$W timeout 600 python tools/dev/race_stats.py MOULDFUNGICIDE HS 6 1 40 0,1,2,3,300 > $O/r4_mould6_w$W.log 2>&1
  MYRIAD_FUSED_WAVES=$W timeout 600 python tools/dev/race_stats.py MOULDFUNGICIDE HS 100 3 40 0,1,2,3,300 > $O/r4_mould100_w$W.log 2>&1
  MYRIAD_FUSED_WAVES=$W timeout 600 python tools/dev/race_stats.py CARTPOLE HS 100 64 20 1,3,300 > $O/r4_cart100_w$W.log 2>&1
done
cat $O/*.log | grep -v amdgpu
