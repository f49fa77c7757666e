#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp16; O=gpurun_out/exp16
for v in default f1 f2 f3 f4 f5 f6 default; do
  if [ $v = default ]; then L=$PWD/myriad_amd/libmyriad_hip.so; else L=$PWD/variants/lib_$v.so; fi
  [ -f $L ] || continue
  MYRIAD_HIP_LIB=$L python bench.py --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value']), round(d['ms_per_step'],3), round(d['solver_kernel']['avg_ms'],3), d['converged_fraction'], d['iterations'])" >> $O/perf.txt
done
cat $O/perf.txt
