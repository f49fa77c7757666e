#!/bin/bash
# exp78: prefetch depth of the sweep's input ring (MYR_RICCATI_PF, default 4) on the headline workload; variant libraries in myriad_amd/_var/ (built by hand)
for v in pf2 pf3 default pf6 pf8 default; do
  if [ $v = default ]; then unset MYRIAD_HIP_LIB; else export MYRIAD_HIP_LIB=$PWD/myriad_amd/_var/lib_$v.so; fi
  python bench.py --steps 20 --warmup 3 --cpu-budget 0 --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$v', round(d['value']), 'solves/s', 'solver kernel', round(d['solver_kernel']['avg_ms'], 3), 'ms', d['converged_fraction'])"
done
# Result (solver kernel ms): PF=2 13.04, 3 12.68, 4 (default) 12.64 / 12.61, 6 12.63, 8 12.65: flat from 3 on -- the default stays.
