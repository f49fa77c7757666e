#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/exp10; O=gpurun_out/exp10
python bench.py --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 > $O/bench.json
MYRIAD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --scaling strong --cpu-budget 0 --no-other-configs > $O/rehearsal_gloo_2ranks_strong.json 2> $O/rehearsal_gloo_strong.err; echo "rc $?" >> $O/rehearsal_gloo_strong.err
MYRIAD_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --scaling weak --batch 2048 --cpu-budget 0 --no-other-configs > $O/rehearsal_gloo_2ranks_weak.json 2> $O/rehearsal_gloo_weak.err; echo "rc $?" >> $O/rehearsal_gloo_weak.err
timeout 900 python -m pytest tests/test_gpu_elastic.py tests/test_integration_stub.py tests/test_gpu_solve.py tests/test_gpu_api.py -q -m gpu > $O/tests.log 2>&1; tail -4 $O/tests.log
python -c "
import json
for f in ('bench','rehearsal_gloo_2ranks_strong','rehearsal_gloo_2ranks_weak'):
  try:
    d=json.loads(open('$O/'+f+'.json').read().strip().split('\n')[-1]); print(f, round(d['value']), d['ms_per_step'], d['n_gpus'], d['config']['per_gpu_batch'], d['download']['value_without_download'])
  except Exception as e: print(f, 'failed', e)
"
tail -3 $O/rehearsal_gloo_strong.err
