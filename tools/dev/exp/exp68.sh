#!/bin/bash
# exp68: the one-wavefront sweep with a scalar stage count (readfirstlane of the call's argument) and with scalar branches on the pivot test,
# headline workload; variant libraries built by hand into myriad_amd/_var/ (A: as before, B: scalar N, C: scalar N + scalar pivot branches)
for v in A B C A B C; do
  MYRIAD_HIP_LIB=$PWD/myriad_amd/_var/lib_$v.so python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$v', round(d['value']), 'solves/s', 'solver kernel', round(d['solver_kernel']['avg_ms'], 3), 'ms', d['converged_fraction'])"
done
# Result (one MI355X, solver kernel ms, two rounds): A 12.645 / 12.611, B 12.585 / 12.568, C 12.542 / 12.577 -- within 0.5 %: not adopted.
