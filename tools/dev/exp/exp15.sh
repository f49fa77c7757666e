#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp15; O=gpurun_out/exp15
export PYTHONUNBUFFERED=1
for v in default spec callw both; do
  for B in 256 512; do
    if [ $v = default ]; then L=$PWD/myriad_amd/libmyriad_hip.so; else L=$PWD/variants/lib_$v.so; fi
    MYRIAD_HIP_LIB=$L python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['config']['global_batch'], round(d['value']), round(d['solver_kernel']['avg_ms'],3), d['converged_fraction'])" >> $O/perf.txt
  done
done
cat $O/perf.txt
for v in both; do
  MYRIAD_HIP_LIB=$PWD/variants/lib_$v.so timeout 900 python -m pytest tests/test_gpu_poison.py -q -m gpu -k "CARTPOLE or TIMBERHARVEST or MOULDFUNGICIDE or BIOREACTOR or CANCERTREATMENT" > $O/poison_$v.log 2>&1; tail -2 $O/poison_$v.log
  for c in "CANCERTREATMENT HS 100 3" "BIOREACTOR HS 20 3" "CARTPOLE HS 100 8"; do MYRIAD_HIP_LIB=$PWD/variants/lib_$v.so timeout 300 python tools/dev/fresh_stats.py $c 12 "MYRIAD_FUSED_WAVES=2" 2>&1 | grep "distinct\|fault"; done
done
