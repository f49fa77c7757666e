#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8; do MYRIAD_BENCH_TRACE=1 python bench.py --cpu-budget 0 --no-other-configs 2>gpurun_out/exp41_err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'no-download', round(d['download']['value_without_download']))"; grep "slow" gpurun_out/exp41_err.txt | cut -c1-100; done
