#!/bin/bash
cd $GRAFT_REPO_ROOT
for cs in 2 2 2 1; do echo "== copy streams $cs"; MYRIAD_BENCH_COPY_STREAMS=$cs MYRIAD_PARK_ITER=12 MYRIAD_BENCH_TRACE=1 python bench.py --cpu-budget 0 --no-other-configs 2>gpurun_out/exp38_err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value', round(d['value']), round(d['ms_per_step'],3), round(d['download']['value_without_download']))"; grep "slow\|per-step" gpurun_out/exp38_err.txt | cut -c1-160; done
