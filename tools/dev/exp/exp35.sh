#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 60 --cpu-budget 0 --no-other-configs 2>gpurun_out/exp35_err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'kernel', round(d['solver_kernel']['avg_ms'],3), 'no-download', round(d['download']['value_without_download']))"; grep "per-step" gpurun_out/exp35_err.txt | cut -c1-400; }
run MYRIAD_PARK_ITER=12 MYRIAD_BENCH_TRACE=1
run MYRIAD_PARK_ITER=0 MYRIAD_BENCH_TRACE=1
run MYRIAD_PARK_ITER=12 MYRIAD_BENCH_TRACE=1 GPU_MAX_HW_QUEUES=8
