#!/bin/bash
# the two forms round 3 gave up, on the round-4 tree (uniform wavefront index, scalar branches in the W > 1 sweeps): speculative second rung; sweep called by wavefront 0
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp14; O=gpurun_out/exp14
export PYTHONUNBUFFERED=1
for v in spec callw; do
  for c in "TIMBERHARVEST HS 6 3" "TIMBERHARVEST TRAP 6 3" "MOULDFUNGICIDE HS 100 3" "BIOREACTOR HS 20 3" "BIOREACTOR HS 100 3" "CANCERTREATMENT HS 50 1" "CANCERTREATMENT HS 100 3" "CARTPOLE HS 100 8"; do
    MYRIAD_HIP_LIB=$PWD/variants/lib_$v.so timeout 300 python tools/dev/fresh_stats.py $c 8 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" >> $O/$v.log 2>&1
  done
  echo "== $v"; grep -h "distinct\|fault" $O/$v.log
done
