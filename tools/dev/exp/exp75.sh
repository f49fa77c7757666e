#!/bin/bash
# exp75: per-pass cycles of the headline kernel, FIRST twelve iterations against whole solves (variants/libtiming.so: SysCARTPOLE.p1 with -DMYR_PHASE_TIMING)
for pi in -1 0; do    # -1: the two-phase launch (the first line per trajectory is phase 1: twelve iterations); 0: whole solves
  echo "MYRIAD_PARK_ITER=$pi"
  MYRIAD_PARK_ITER=$pi MYRIAD_VARIANT_LIB=variants/libtiming.so python tools/dev/phase_timing.py 4096 2>&1 | grep -E "traj [0-9]+ it|status" | awk '!s[$1,$2]++' | head -12
done
