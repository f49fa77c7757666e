#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --cpu-budget 0 --no-other-configs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'kernel', round(d['solver_kernel']['avg_ms'],3), 'eval', round(d['roofline']['avg_ms'],4), round(d['roofline']['frac'],3), 'no-download', round(d['download']['value_without_download']))"; }
run A=1; run MYRIAD_PARK_ITER=0; run A=1; run MYRIAD_PARK_ITER=0; run A=1
python -m pytest tests/test_gpu_poison.py -q -x -k "two_phase or headline" 2>&1 | grep -E "passed|failed"
