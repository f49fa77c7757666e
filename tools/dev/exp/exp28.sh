#!/bin/bash
# exp28: the speculative rung + the called sweep together (fails the fresh-handle gate) under the compiler settings that cure the lane miscompile
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp28; export PYTHONUNBUFFERED=1
for l in $SC_LIBS; do
  echo "=== $l"
  MYRIAD_HIP_LIB=$PWD/variants/libsc_$l.so timeout 300 python tools/dev/fresh_stats.py ${SC_CASE:-CANCERTREATMENT TRAP 6 1} 12 "MYRIAD_FUSED_WAVES=1,MYRIAD_FUSED_WAVES=2" 2>&1 | grep "distinct\|   x" | cut -c1-170
done | tee gpurun_out/exp28/out.txt
