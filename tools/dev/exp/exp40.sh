#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; for i in 1 2 3; do env "$@" MYRIAD_BENCH_TRACE=1 python bench.py --cpu-budget 0 --no-other-configs 2>gpurun_out/exp40_err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'no-download', round(d['download']['value_without_download']))"; grep "slow" gpurun_out/exp40_err.txt | cut -c1-100; done; }
run ROC_SIGNAL_POOL_SIZE=512
run HIP_LAUNCH_BLOCKING=0 GPU_MAX_HW_QUEUES=2
run ROC_USE_FGS_KERNARG=0
run A=1
