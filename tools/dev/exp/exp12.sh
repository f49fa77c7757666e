#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp12; O=gpurun_out/exp12
export PYTHONUNBUFFERED=1
for i in 1 2; do
  MYRIAD_HIP_LIB=$PWD/variants/lib_uni.so python bench.py --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/uni.jsonl
  python bench.py --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/default.jsonl
done
for B in 256 512; do MYRIAD_HIP_LIB=$PWD/variants/lib_uni.so python bench.py --batch $B --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 >> $O/uni.jsonl; done
MYRIAD_HIP_LIB=$PWD/variants/lib_uni.so timeout 900 python -m pytest tests/test_gpu_poison.py -q -m gpu -k "CARTPOLE or TIMBERHARVEST or MOULDFUNGICIDE or BIOREACTOR or CANCERTREATMENT or VANDERPOL" > $O/poison.log 2>&1; tail -2 $O/poison.log
MYRIAD_HIP_LIB=$PWD/variants/lib_uni.so timeout 900 python -m pytest tests/test_gpu_solve.py -q -m gpu > $O/solve.log 2>&1; tail -2 $O/solve.log
MYRIAD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --scaling weak --batch 2048 --cpu-budget 0 --no-other-configs > $O/rehearsal_gloo_weak.json 2> $O/rehearsal.err; echo "rehearsal rc $?"
python -c "
import json
for f in ('uni','default'):
  for l in open('$O/'+f+'.jsonl'):
    d=json.loads(l); print(f, d['config']['global_batch'], round(d['value']), round(d['ms_per_step'],3), round(d['solver_kernel']['avg_ms'],3))
d=json.loads(open('$O/rehearsal_gloo_weak.json').read().strip().split('\n')[-1]); print('rehearsal', d['n_gpus'], round(d['value']), d['ms_per_step'])
"
