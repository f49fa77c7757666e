#!/bin/bash
# exp57 (round 5): the speculative rung + the called sweep (5-7 results on 12 fresh handles, round 4) -- inherited stack contents or a race?
#   MYRIAD_STACK_FILL leaves one pattern in the queue's private-segment memory before every solver launch: results that follow the pattern = a stack slot
#   read before it is written; the same scatter under every pattern = timing.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp57; export PYTHONUNBUFFERED=1
for l in both both_strong; do
for fill in none zero nan big random; do
  echo "=== lib $l, stack fill $fill"
  if [ $fill = none ]; then unset MYRIAD_STACK_FILL; else export MYRIAD_STACK_FILL=$fill; fi
  MYRIAD_HIP_LIB=$PWD/variants/libsc_$l.so timeout 300 python tools/dev/fresh_stats.py CANCERTREATMENT TRAP 6 1 12 "MYRIAD_FUSED_WAVES=2" 2>&1 | grep "distinct\|   x" | cut -c1-170
done
done | tee gpurun_out/exp57/out.txt
