#!/bin/bash
# exp71: the two-wavefront form beyond two trajectories per CU (B = 768 .. 2048): one wavefront per trajectory (default there) against MYRIAD_FUSED_WAVES=2
for B in 768 1024 1536 2048; do
  for w in 1 2; do
    MYRIAD_FUSED_WAVES=$w python bench.py --batch $B --steps 20 --warmup 3 --cpu-budget 0 --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('B=$B waves=$w', round(d['value']), 'solves/s', 'solver kernel', round(d['solver_kernel']['avg_ms'], 3), 'ms', d['converged_fraction'])"
  done
done
# Result (one MI355X, solver kernel ms, waves 1 / 2): B=768 3.65 / 4.81, B=1024 4.79 / 5.81, B=1536 6.26 / 7.81, B=2048 6.82 / 9.98 -- the switch at two
# trajectories per CU stays.  (B=768 and B=1024 are ONE round of the one-wavefront form: their time is the longest solve of the draw -- 34 iterations at B=1024.)
